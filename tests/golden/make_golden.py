"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/fitc_oracle.py).

The reference (OCaml) cannot be run in the build image and ships no golden vectors, so these
fixtures are outputs of the oracle's reference-sequence evaluation (`evaluate`: per-hyper traces,
QR-based model) on seeded inputs.  They pin (a) the oracle against accidental change and (b) the
HIP path on the GPU box, where the oracle's own runtime deps (numpy/scipy) are also present.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fitc_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def synth(seed, n, m, d):
    """BASELINE.md section 2 generator (numpy PCG64)."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(d, n))
    y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
    Z = X[:, rng.permutation(n)[:m]] + 0.01 * rng.normal(size=(d, m))
    return np.asfortranarray(X), y, np.asfortranarray(Z)


def gen_data_1d(seed, n, m):
    """test/gen_data.ml:23-44 recipe: f(x) = sin(3x)/x + |x-3|/(x^2+1), noise sigma 0.7, x ~ U(-5,5)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-5.0, 5.0, size=n)
    y = np.sin(3 * x) / x + np.abs(x - 3) / (x * x + 1) + 0.7 * rng.normal(size=n)
    Z = x[rng.permutation(n)[:m]][None, :].copy()
    return np.asfortranarray(x[None, :]), y, np.asfortranarray(Z)


def save(name, k, X, y, Z, sigma2, variational, extra):
    out = O.evaluate(k, Z, X, y, sigma2, variational=variational, keep=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), X=X, y=y, Z=Z, sigma2=sigma2, variational=variational,
        l1=out["l1"], l2=out["l2"], l=out["l"], dl_dsigma2=out["dl_dsigma2"], grad=out["grad"],
        coeffs=out["coeffs"], model_dl_dsigma2=out["model_dl_dsigma2"], model_grad=out["model_grad"],
        r_vec=out["model"]["r_vec"], is_vec=out["model"]["is_vec"], v_vec=out["trained"]["v_vec"],
        w_vec=out["trained"]["w_vec"], **extra)
    print(name, "l=%.12g" % out["l"], "n_hypers=%d" % len(out["grad"]))


def mp_truth(k, X, y, Z, sigma2, dps=40):
    """l1, l2 and the mean coefficients of FITC in `dps`-digit arithmetic from the dense definition
    (doc/manual/gpr_manual.tex:694-701), starting from the double-precision covariance entries."""
    import mpmath as mp
    mp.mp.dps = dps
    n, m = X.shape[1], Z.shape[1]
    km, _ = O.spec_calc_shared_upper(k, Z)
    km = np.triu(km) + np.triu(km, 1).T
    knm, _ = O.spec_calc_shared_cross(k, X, Z)
    Km = mp.matrix(km.tolist()) + mp.mpf(O.CHOLESKY_JITTER) * mp.eye(m)
    Knm = mp.matrix(knm.tolist())
    A1 = Knm * (Km ** -1)
    s = [mp.mpf(k.sf2) - sum(A1[i, j] * Knm[i, j] for j in range(m)) + mp.mpf(sigma2) for i in range(n)]
    B = Km.copy()
    for i in range(n):
        for a in range(m):
            for b in range(a, m):
                B[a, b] += Knm[i, a] * Knm[i, b] / s[i]
    for a in range(m):
        for b in range(a):
            B[a, b] = B[b, a]
    yv = [mp.mpf(float(v)) for v in y]
    c = mp.matrix([sum(Knm[i, a] * yv[i] / s[i] for i in range(n)) for a in range(m)])
    t = mp.lu_solve(B, c)
    l1 = -mp.mpf(0.5) * (mp.log(mp.det(B)) - mp.log(mp.det(Km)) + sum(mp.log(si) for si in s) + n * mp.log(2 * mp.pi))
    l2 = -mp.mpf(0.5) * (sum(yv[i] ** 2 / s[i] for i in range(n)) - sum(c[a] * t[a] for a in range(m)))
    return float(l1), float(l2), np.array([float(v) for v in t])


def save_posterior(name, k, X, y, Z, sigma2, Xt, z, extra):
    """Posterior quantities either side of the evidence path: Means/Variances,
    FITC_/FIC_covariances, Cov_sampler with the given draws z, Stats."""
    out = O.evaluate(k, Z, X, y, sigma2, want_grad=False, keep=True)
    knm, _ = O.spec_calc_shared_cross(k, X, Z)
    train_means = knm @ out["coeffs"]
    st = O.stats_calc(y, train_means, out["l"])
    means = O.predict_means(k, Z, out["coeffs"], Xt)
    fitc = O.fitc_covariances(k, Z, out["model"], Xt)
    fic = O.fic_covariances(k, Z, out["model"], Xt)
    smp = O.cov_sampler_calc(means, fitc, sigma2, predictive=True)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), X=X, y=y, Z=Z, sigma2=sigma2, Xt=Xt, z=z, l=out["l"],
        coeffs=out["coeffs"], train_means=train_means, means=means,
        variances=O.predict_variances(k, Z, out["model"], Xt, predictive=False),
        fitc_cov=fitc, fic_cov=fic, samples=O.cov_sampler_samples(smp, z),
        stats=np.array([st[key] for key in STAT_KEYS], dtype=np.float64), **extra)
    print(name, "l=%.12g" % out["l"], "rmse=%.6g" % st["rmse"])


STAT_KEYS = ("n_samples", "target_variance", "sse", "mse", "rmse", "smse", "msll", "mad", "maxad")


def main():
    # C1 shape of BASELINE.json (n=2000 m=50 d=3), seed 1, standard + variational
    X, y, Z = synth(1, 2000, 50, 3)
    le = 0.5 * np.log(3)
    for var in (False, True):
        save("iso_c1" + ("_var" if var else ""), O.SeIsoKernel(le, 0.0), X, y, Z, 0.1, var,
             dict(kind="iso", log_ell=le, log_sf2=0.0))
    # test_derivatives.ml shape: n=10, m=5, D=3
    X, y, Z = synth(7, 10, 5, 3)
    save("iso_tiny", O.SeIsoKernel(0.1, -0.2), X, y, Z, 1.0, False, dict(kind="iso", log_ell=0.1, log_sf2=-0.2))
    # ragged sizes (not multiples of any tile), d=8
    X, y, Z = synth(11, 777, 131, 8)
    le = 0.5 * np.log(8)
    save("iso_ragged", O.SeIsoKernel(le, 0.3), X, y, Z, 0.05, False, dict(kind="iso", log_ell=le, log_sf2=0.3))
    # save_data.ml recipe (1-D function), n=1000, m=10
    X, y, Z = gen_data_1d(3, 1000, 10)
    save("iso_gen_data", O.SeIsoKernel(0.0, 0.0), X, y, Z, 0.49, False, dict(kind="iso", log_ell=0.0, log_sf2=0.0))
    # Cov_se_fat with a general projection (D=5 -> d=3) and the ARD special case (diagonal tproj)
    rng = np.random.default_rng(21)
    Xb = np.asfortranarray(rng.normal(size=(5, 400)))
    yb = np.sin(Xb.sum(0)) + 0.1 * rng.normal(size=400)
    P = np.asfortranarray(0.5 * rng.normal(size=(5, 3)))
    kf = O.SeFatKernel(3, 0.2, P)
    Zf = np.asfortranarray(O.se_fat_project(kf, Xb[:, rng.permutation(400)[:20]]) + 0.01 * rng.normal(size=(3, 20)))
    for var in (False, True):
        save("fat_proj" + ("_var" if var else ""), kf, Xb, yb, Zf, 0.1, var,
             dict(kind="fat", d=3, log_sf2=0.2, tproj=P))
    ell = rng.uniform(-0.5, 0.5, size=4)
    Pd = np.asfortranarray(np.diag(np.exp(-ell)))
    Xa = np.asfortranarray(rng.normal(size=(4, 300)))
    ya = np.sin(Xa.sum(0)) + 0.1 * rng.normal(size=300)
    ka = O.SeFatKernel(4, 0.0, Pd)
    Za = np.asfortranarray(O.se_fat_project(ka, Xa[:, :16]) + 0.01 * rng.normal(size=(4, 16)))
    save("fat_ard", ka, Xa, ya, Za, 0.1, False, dict(kind="fat", d=4, log_sf2=0.0, tproj=Pd))
    # heteroskedastic noise on diag(K_m) together with a projection (lib/cov_se_fat.ml:136-142)
    lh = rng.normal(size=20) - 3.0
    kh = O.SeFatKernel(3, 0.2, P, lh)
    save("fat_hetero", kh, Xb, yb, Zf, 0.1, False, dict(kind="fat", d=3, log_sf2=0.2, tproj=P, log_hetero=lh))
    # everything on, as Cov_se_fat.Eval.Inputs.create_default_kernel_params does (lib/cov_se_fat.ml:191-213)
    lms = 0.3 * rng.normal(size=(3, 20))
    kall = O.SeFatKernel(3, 0.2, P, lh, lms)
    save("fat_all", kall, Xb, yb, Zf, 0.1, False,
         dict(kind="fat", d=3, log_sf2=0.2, tproj=P, log_hetero=lh, log_multiscales=lms))
    # multiscales without projection
    lms4 = 0.2 * rng.normal(size=(4, 16))
    kms = O.SeFatKernel(4, 0.0, None, None, lms4)
    save("fat_multiscale", kms, Xa, ya, np.asfortranarray(Xa[:, :16] + 0.01), 0.2, True,
         dict(kind="fat", d=4, log_sf2=0.0, log_multiscales=lms4))
    # Cov_se_fat without projection
    kn = O.SeFatKernel(4, -0.1, None)
    save("fat_noproj", kn, Xa, ya, np.asfortranarray(Xa[:, :16] + 0.01), 0.2, False,
         dict(kind="fat", d=4, log_sf2=-0.1))
    # Jitter-dominated K_m (ell = e, cond(K_m + jitter) ~ 1e7 .. 4e7): the regime where forming B = K_m + K_mn S^-1 K_nm
    # and factoring it loses 1e-5 (lib/fitc_gp.ml:170-182 is why the reference uses QR).  C1 shape with the
    # oracle's values, and a small case that also carries a 40-digit evaluation.
    X, y, Z = synth(1, 2000, 50, 3)
    for tag, s2 in (("lo", 1e-4), ("hi", 1.0)):
        save("illcond_c1_" + tag, O.SeIsoKernel(1.0, 0.0), X, y, Z, s2, False, dict(kind="iso", log_ell=1.0, log_sf2=0.0))
    X, y, Z = synth(5, 200, 30, 3)
    for tag, s2 in (("lo", 1e-4), ("hi", 1.0)):
        l1, l2, t = mp_truth(O.SeIsoKernel(1.0, 0.0), X, y, Z, s2)
        save("illcond_small_" + tag, O.SeIsoKernel(1.0, 0.0), X, y, Z, s2, False,
             dict(kind="iso", log_ell=1.0, log_sf2=0.0, mp_l1=l1, mp_l2=l2, mp_coeffs=t))
    # posterior fixtures: iso at a ragged size, and Cov_se_fat with projection + hetero + multiscales
    # (K_tt of the covariances is the plain kernel of the projected test points, lib/cov_se_fat.ml:221)
    prng = np.random.default_rng(99)
    X, y, Z = synth(41, 500, 20, 2)
    save_posterior("posterior_iso", O.SeIsoKernel(0.2, 0.1), X, y, Z, 0.15,
                   np.asfortranarray(prng.normal(size=(2, 37))), np.asfortranarray(prng.normal(size=(37, 4))),
                   dict(kind="iso", log_ell=0.2, log_sf2=0.1))
    save_posterior("posterior_fat_all", kall, Xb, yb, Zf, 0.1,
                   np.asfortranarray(prng.normal(size=(5, 29))), np.asfortranarray(prng.normal(size=(29, 3))),
                   dict(kind="fat", d=3, log_sf2=0.2, tproj=P, log_hetero=lh, log_multiscales=lms))


if __name__ == "__main__":
    main()
