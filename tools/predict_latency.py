"""Wall time of gprhip_predict (means only / means + variances) and of the training statistics for few-inducing-point
models: few and many test points, from a small and from a large training set.
    usage (GPU box, repo root): python3 tools/predict_latency.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gpr_amd
from bench import synth
for n, m, d, nt in ((2000, 50, 3, 1000), (2000, 50, 3, 100000), (100000, 50, 3, 100000)):
    X, y, Z = synth(1, n, m, d)
    Xt = np.asfortranarray(np.random.default_rng(0).normal(size=(d, nt)))
    p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m)
    p.set_inputs(X); p.set_targets(y)
    p.eval(log_ell=0.5*np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z)
    for what in ("means", "both"):
        ts = []
        for _ in range(13):
            t0 = time.perf_counter()
            if what == "means": p.predict(Xt, want_variances=False)
            else: p.predict(Xt, predictive=True)
            ts.append(time.perf_counter() - t0)
        print(n, m, nt, what, "%.3f ms" % (1e3*np.median(ts[3:])))
    t = []
    for _ in range(13):
        t0 = time.perf_counter(); p.train_stats(want_means=False); t.append(time.perf_counter()-t0)
    print(n, m, "train_stats %.3f ms" % (1e3*np.median(t[3:])))
    p.close()
