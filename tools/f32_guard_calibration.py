"""fp32-bulk mean coefficients against the fp64 ones, beside the library's condition estimate of K_m + jitter
(gprhip_condition) and numpy's: calibrates the GPRHIP_F32_COEFF_TOL guard.   usage (GPU box): python3 tools/f32_guard_calibration.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpr_amd  # noqa: E402
from bench import synth  # noqa: E402

print("%3s %5s %6s %10s %10s %10s %10s %10s" % ("d", "m", "n", "cond_est", "cond_np", "bound", "coeff_err", "l_err"))
for d in (1, 3, 8, 16):
    for m in (50, 128, 200, 350, 512):
        n = 6000
        X, y, Z = synth(100 + d * 7 + m, n, m, d)
        kw = dict(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
        res = {}
        for prec in (gpr_amd.F64, gpr_amd.F32_BULK):
            p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m, precision=prec)
            p.set_inputs(X)
            p.set_targets(y)
            ev = p.eval(**kw)
            cond, bound = p.condition()
            if prec == gpr_amd.F64:
                km = p.debug_fetch_matrix("km")
                km = km + np.triu(km, 1).T + 1e-6 * np.eye(m)
                w = np.linalg.eigvalsh(km)
                cond_np = w[-1] / w[0]
            res[prec] = (ev, cond, bound)
            p.close()
        e64, e32 = res[gpr_amd.F64][0], res[gpr_amd.F32_BULK][0]
        cerr = np.max(np.abs(e32.coeffs - e64.coeffs)) / np.max(np.abs(e64.coeffs))
        print("%3d %5d %6d %10.3g %10.3g %10.3g %10.3g %10.3g" % (d, m, n, res[gpr_amd.F32_BULK][1], cond_np,
                                                              res[gpr_amd.F32_BULK][2], cerr, abs(e32.l - e64.l) / abs(e64.l)), flush=True)
