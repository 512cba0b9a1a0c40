"""BASELINE.json config 4's per-GPU shard at full size: cov_se_iso, n=1M of 8M rows, m=4096, d=16, fp64; prints the
whole-evaluation time and the per-stage times (separate timing pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpr_amd
from bench import synth

n, m, d = int(os.environ.get("N", 1_000_000)), 4096, 16
X, y, Z = synth(4, n, m, d)
p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m)
p.set_inputs(X); p.set_targets(y)
kw = dict(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
p.set_timing(0)
for it in range(3):
    t0 = time.time()
    ev = p.eval(**kw)
    dt = time.time() - t0
F = n * (6.0 * m * m + 4.0 * m * d) + 2.0 * m ** 3
print("C4 shard: %.3f s/eval  %.3f Mpts/s  l=%.6f  algorithmic %.1f TFLOP/s" % (dt, n / dt / 1e6, ev.l, F / dt * 1e-12))
p.set_timing(2)
p.eval(**kw)
print("   ", {k: round(v, 2) for k, v in sorted(p.last_timings().items(), key=lambda kv: -kv[1])})
p.close()
