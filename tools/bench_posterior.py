"""Timing of the posterior paths at the headline shape: n=1M, m=2048, d=8, fp64.

    python tools/bench_posterior.py [--points N] [--inducing M] [--test NT] [--cov NC]

Prints one JSON line: prediction throughput (test points/s, host buffers in and out, so PCIe-inclusive),
training-set statistics time, covariance-matrix and sampler times.  Algorithmic flops: prediction
variances 2*nt*m^2 (two triangular products, half-dense), covariances 2*nt*m^2 + 4*nc^2*m (fp64, full squares).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpr_amd  # noqa: E402
from bench import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--inducing", type=int, default=2048)
    ap.add_argument("--dims", type=int, default=8)
    ap.add_argument("--test", type=int, default=262_144)
    ap.add_argument("--cov", type=int, default=8192)
    a = ap.parse_args()
    n, m, d = a.points, a.inducing, a.dims
    X, y, Z = synth(2, n, m, d)
    p = gpr_amd.Problem(gpr_amd.COV_SE_ISO, n, d, d, m)
    p.set_inputs(X)
    p.set_targets(y)
    ev = p.eval(log_ell=0.5 * np.log(d), log_sf2=0.0, sigma2=0.1, inducing=Z, want_grad=False)
    rng = np.random.default_rng(0)
    Xt = np.asfortranarray(rng.normal(size=(d, a.test)))

    def timed(f, reps=3):
        f()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = f()
        return (time.perf_counter() - t0) / reps, out

    t_mean, _ = timed(lambda: p.predict(Xt, want_variances=False))
    t_var, (mu, var) = timed(lambda: p.predict(Xt, predictive=False))
    t_stats, (sums, _) = timed(lambda: p.train_stats())
    Xc = np.asfortranarray(Xt[:, :a.cov])
    t_cov, cov = timed(lambda: p.covariances(Xc, kind="FITC", predictive=False))
    z = rng.normal(size=(a.cov, 64))
    t_smp, S = timed(lambda: p.cov_samples(cov, mu[:a.cov], z, add_diag=0.1))
    assert np.all(np.isfinite(S)) and np.all(var > -1e-9)
    print(json.dumps({
        "shape": {"n": n, "m": m, "d": d, "test_points": a.test, "cov_points": a.cov},
        "l": ev.l,
        "predict_means_pts_per_s": a.test / t_mean, "predict_means_ms": t_mean * 1e3,
        "predict_means_variances_pts_per_s": a.test / t_var, "predict_means_variances_ms": t_var * 1e3,
        "predict_variances_tflops": 2.0 * a.test * m * m / t_var * 1e-12,
        "train_stats_ms": t_stats * 1e3, "train_stats_pts_per_s": n / t_stats,
        "rmse": float(np.sqrt(sums[0] / n)),
        "covariances_ms": t_cov * 1e3,
        "covariances_tflops": (2.0 * a.cov * m * m + 4.0 * a.cov * a.cov * m) / t_cov * 1e-12,
        "cov_sampler_ms": t_smp * 1e3, "cov_sampler_potrf_tflops": (a.cov ** 3 / 3.0) / t_smp * 1e-12,
    }))
    p.close()


if __name__ == "__main__":
    main()
