"""Stage times of the headline shape without the finiteness asserts of bench.py (for timing-only lab switches)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpr_amd
from bench import synth
n, m, d = 1_000_000, 2048, 8
X, y, Z = synth(2, n, m, d)
p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m)
p.set_inputs(X); p.set_targets(y); p.set_timing(2)
kw = dict(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
acc = {}
for i in range(5):
    try:
        p.eval(**kw)
    except gpr_amd.GprHipError as e:
        pass  # wrong numbers may fail the second factorisation; the stage timers of the stages that ran remain
    if i >= 2:
        for k, v in p.last_timings().items():
            acc.setdefault(k, []).append(v)
print({k: round(float(np.mean(v)), 2) for k, v in acc.items() if np.mean(v) > 50})
