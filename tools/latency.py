"""Whole-evaluation wall time at small and medium problem sizes (where launch latency, not the matrix cores, sets the
time): gradient and evidence-only evaluations of cov_se_iso, median of `REPS` calls after 3 warm-ups.
    usage (GPU box, repo root): python3 tools/latency.py [n,m,d ...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpr_amd  # noqa: E402
from bench import synth  # noqa: E402

REPS = int(os.environ.get("REPS", 30))
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (2000, 50, 3), (1000, 10, 1), (100000, 50, 3), (1000000, 50, 8), (2000, 128, 3), (10000, 256, 8), (50000, 512, 8),
    (100000, 1024, 8), (16384, 2048, 8)]
for n, m, d in shapes:
    X, y, Z = synth(1, n, m, d)
    p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m)
    p.set_inputs(X)
    p.set_targets(y)
    kw = dict(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
    out = []
    for want_grad in (True, False):
        ts = []
        for i in range(REPS + 3):
            t0 = time.perf_counter()
            ev = p.eval(want_grad=want_grad, **kw)
            ts.append(time.perf_counter() - t0)
        out.append(1e3 * float(np.median(ts[3:])))
    p.set_timing(2)
    p.eval(**kw)
    p.eval(**kw)
    st = {k: round(v, 3) for k, v in p.last_timings().items()}
    print("n=%d m=%d d=%d: gradient eval %.3f ms, evidence only %.3f ms; stages %s" % (n, m, d, out[0], out[1], st),
          flush=True)
    p.close()
