"""BASELINE.json config 3 at full size: cov_se_fat (ARD special case), n=1M, m=4096, d=32, fp32 bulk and fp64;
prints whole-evaluation time and the per-stage times (separate timing pass).  PREC=f32|f64 restricts to one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpr_amd

n, m, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("M", 4096)), 32
rng = np.random.default_rng(3)
X = np.asfortranarray(rng.normal(size=(d, n)))
y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
ell = rng.uniform(-0.5, 0.5, size=d)
P = np.asfortranarray(np.diag(np.exp(-ell)) / np.sqrt(d))
Z = np.asfortranarray((P.T @ X[:, rng.permutation(n)[:m]]) + 0.01 * rng.normal(size=(d, m)))
which = os.environ.get("PREC", "")
for prec, name in ((gpr_amd.F32_BULK, "f32"), (gpr_amd.F64, "f64")):
    if which and which != name:
        continue
    p = gpr_amd.Problem(gpr_amd.COV_SE_FAT, n, d, d, m, precision=prec)
    p.set_inputs(X); p.set_targets(y)
    p.set_timing(int(os.environ.get("TIMING", 0)))
    for it in range(3):
        t0 = time.time()
        ev = p.eval(log_sf2=0.0, sigma2=0.1, inducing=Z, tproj=P)
        dt = time.time() - t0
    F = n * (6.0 * m * m + 4.0 * m * d) + 2.0 * m ** 3
    print("%s: %.3f s/eval  %.3f Mpts/s  l=%.6f  |grad|=%.4e  algorithmic %.1f TFLOP/s" % (
        name, dt, n / dt / 1e6, ev.l, np.linalg.norm(ev.grad), F / dt * 1e-12))
    p.set_timing(2)
    p.eval(log_sf2=0.0, sigma2=0.1, inducing=Z, tproj=P)
    tm = p.last_timings()
    print("   ", {k: round(v, 2) for k, v in sorted(tm.items(), key=lambda kv: -kv[1])})
    p.close()
