"""Pins the CPU oracle (oracle/fitc_oracle.py).  The reference ships no golden vectors and cannot
be run here (no OCaml toolchain), so the oracle is pinned by independent known answers:
dense textbook FITC, finite differences, a 40-digit mpmath evaluation, the formulas of the
reference's Octave cross-check (test/oct.m), and the reference's own gradient self-test recipe."""
import numpy as np
import pytest

from oracle import fitc_oracle as O
from tests.util import (STAT_KEYS, golden_names, illcond_golden_names, load_golden, oracle_kernel, posterior_golden_names, relinf,
                        synth)


@pytest.mark.parametrize("variational", [False, True])
def test_qr_path_equals_dense_textbook_fitc(variational):
    X, y, Z = synth(0, 200, 12, 3)
    k = O.SeIsoKernel(0.5 * np.log(3), 0.0)
    out = O.evaluate(k, Z, X, y, 0.1, variational=variational, want_grad=False)
    dense = O.dense_fitc_log_evidence(k, Z, X, y, 0.1, variational=variational)
    assert abs(out["l"] - dense) <= 1e-11 * abs(dense)


def _central_fd(f, x0, eps=1e-5):
    return (f(x0 + eps) - f(x0 - eps)) / (2 * eps)


@pytest.mark.parametrize("variational", [False, True])
def test_iso_gradient_against_central_differences(variational):
    X, y, Z = synth(3, 150, 8, 3)
    le, ls, s2 = 0.3, -0.1, 0.2
    k = O.SeIsoKernel(le, ls)
    out = O.evaluate(k, Z, X, y, s2, variational=variational)
    L = lambda le_, ls_, Z_, s2_: O.dense_fitc_log_evidence(O.SeIsoKernel(le_, ls_), Z_, X, y, s2_, variational)
    assert abs(_central_fd(lambda v: L(v, ls, Z, s2), le) - out["grad"][0]) < 1e-5 * max(1, abs(out["grad"][0]))
    assert abs(_central_fd(lambda v: L(le, v, Z, s2), ls) - out["grad"][1]) < 1e-5 * max(1, abs(out["grad"][1]))
    assert abs(_central_fd(lambda v: L(le, ls, Z, v), s2) - out["dl_dsigma2"]) < 1e-5 * max(1, abs(out["dl_dsigma2"]))
    for ind, dim in ((1, 1), (4, 2), (8, 3)):
        def f(v):
            Zp = Z.copy()
            Zp[dim - 1, ind - 1] = v
            return L(le, ls, Zp, s2)
        g = out["grad"][2 + (ind - 1) * 3 + dim - 1]
        assert abs(_central_fd(f, Z[dim - 1, ind - 1]) - g) < 1e-5 * max(1, abs(g))


def test_fat_gradient_against_central_differences():
    rng = np.random.default_rng(5)
    D, d, n, m = 5, 3, 120, 7
    X = np.asfortranarray(rng.normal(size=(D, n)))
    y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
    P = 0.5 * rng.normal(size=(D, d))
    k = O.SeFatKernel(d, 0.2, P)
    Z = np.asfortranarray(O.se_fat_project(k, X[:, :m]) + 0.01)
    out = O.evaluate(k, Z, X, y, 0.1)
    hypers = out["hypers"]
    L = lambda ls_, P_, Z_: O.dense_fitc_log_evidence(O.SeFatKernel(d, ls_, P_), Z_, X, y, 0.1)
    assert abs(_central_fd(lambda v: L(v, P, Z), 0.2) - out["grad"][0]) < 1e-5 * max(1, abs(out["grad"][0]))
    for big, small in ((1, 1), (3, 2), (5, 3)):
        def f(v):
            Pp = P.copy()
            Pp[big - 1, small - 1] = v
            return L(0.2, Pp, Z)
        g = out["grad"][hypers.index(("proj", big, small))]
        assert abs(_central_fd(f, P[big - 1, small - 1]) - g) < 1e-5 * max(1, abs(g))
    def fz(v):
        Zp = Z.copy()
        Zp[1, 2] = v
        return L(0.2, P, Zp)
    g = out["grad"][hypers.index(("inducing", 3, 2))]
    assert abs(_central_fd(fz, Z[1, 2]) - g) < 1e-5 * max(1, abs(g))


def test_model_only_gradient_is_l1_derivative():
    X, y, Z = synth(9, 100, 6, 2)
    k = O.SeIsoKernel(0.2, 0.1)
    out = O.evaluate(k, Z, X, y, 0.3)
    l1 = lambda le: O.evaluate(O.SeIsoKernel(le, 0.1), Z, X, y, 0.3, want_grad=False)["l1"]
    assert abs(_central_fd(l1, 0.2) - out["model_grad"][0]) < 1e-5 * max(1, abs(out["model_grad"][0]))
    l1s = lambda s2: O.evaluate(k, Z, X, y, s2, want_grad=False)["l1"]
    assert abs(_central_fd(l1s, 0.3) - out["model_dl_dsigma2"]) < 1e-5 * max(1, abs(out["model_dl_dsigma2"]))


def test_against_mpmath_40_digits():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    X, y, Z = synth(2, 40, 6, 2)
    k = O.SeIsoKernel(0.4, 0.0)
    s2 = 0.05
    n, m = 40, 6
    km, _ = O.spec_calc_shared_upper(k, Z)
    km = np.triu(km) + np.triu(km, 1).T
    knm, _ = O.spec_calc_shared_cross(k, X, Z)
    Km = mp.matrix(km.tolist()) + mp.mpf(O.CHOLESKY_JITTER) * mp.eye(m)
    Knm = mp.matrix(knm.tolist())
    A1 = Knm * (Km ** -1)
    s = [mp.mpf(k.sf2) - sum(A1[i, j] * Knm[i, j] for j in range(m)) + mp.mpf(s2) for i in range(n)]
    B = Km.copy()
    for i in range(n):
        for a in range(m):
            for b in range(m):
                B[a, b] += Knm[i, a] * Knm[i, b] / s[i]
    c = mp.matrix([sum(Knm[i, a] * mp.mpf(float(y[i])) / s[i] for i in range(n)) for a in range(m)])
    t = mp.lu_solve(B, c)
    l1 = -mp.mpf(0.5) * (mp.log(mp.det(B)) - mp.log(mp.det(Km)) + sum(mp.log(si) for si in s) + n * mp.log(2 * mp.pi))
    l2 = -mp.mpf(0.5) * (sum(mp.mpf(float(y[i])) ** 2 / s[i] for i in range(n)) - sum(c[a] * t[a] for a in range(m)))
    out = O.evaluate(k, Z, X, y, s2, want_grad=False)
    assert abs(out["l1"] - float(l1)) < 1e-10 * abs(float(l1))
    assert abs(out["l2"] - float(l2)) < 1e-10 * abs(float(l2))
    assert np.max(np.abs(out["coeffs"] - np.array([float(v) for v in t]))) < 1e-8 * np.max(np.abs(out["coeffs"]))


def test_oct_m_identities():
    """test/oct.m:88-180 -- l = l1 + l2; dl = dl1 + dl2 with W1/X1 (model part) and W2/X2 (target part);
    dls = dls1 + dls2; variational vl1 = l1 - 0.5 is'r."""
    X, y, Z = synth(4, 90, 7, 2)
    k = O.SeIsoKernel(0.1, 0.2)
    out = O.evaluate(k, Z, X, y, 0.4, keep=True)
    model, cm, tr = out["model"], out["cm"], out["trained"]
    n = model["n"]
    Q = model["q_mat"][:n]
    y_ = np.sqrt(model["is_vec"]) * y
    u = y_ - Q @ (Q.T @ y_)
    assert abs(out["l2"] - (-0.5 * float(u @ y_))) < 1e-12 * abs(out["l2"])       # oct.m:119-122
    U, S = O.calc_us_mat(model)
    Tm = O._upper_to_full(cm["t_mat"])
    v1 = model["is_vec"] * (1 - np.sum(Q * Q, axis=1))                              # oct.m:133
    w = np.sqrt(model["is_vec"]) * u                                                # oct.m:140
    v2 = w * w
    t = S.T @ y                                                                     # oct.m:118
    assert np.allclose(t, out["coeffs"], rtol=1e-9, atol=1e-12)
    W1 = Tm - (U * v1[:, None]).T @ U
    W2 = np.outer(t, t) - (U * v2[:, None]).T @ U
    X1 = S - U * v1[:, None]
    X2 = np.outer(w, t) - U * v2[:, None]
    ht = out["hyper_t"]
    assert np.allclose(np.triu(W1 - W2), np.triu(ht["w_mat"]), rtol=1e-9, atol=1e-10)  # W = W1 - W2
    assert np.allclose(X1 - X2, ht["x_mat"], rtol=1e-9, atol=1e-10)                    # X = X1 - X2
    assert abs(out["dl_dsigma2"] - (-0.5 * v1.sum() + 0.5 * v2.sum())) < 1e-10       # oct.m:149-151
    assert abs(out["model_dl_dsigma2"] - (-0.5 * v1.sum())) < 1e-10
    var = O.evaluate(k, Z, X, y, 0.4, variational=True, want_grad=False)
    assert abs(var["l1"] - (out["l1"] - 0.5 * float(model["is_vec"] @ model["r_vec"]))) < 1e-10  # oct.m:159


def test_reference_self_test_recipe():
    """lib/fitc_gp.ml:1398-1462 at the reference's eps=1e-8 / tol=1e-2, on test_derivatives.ml's shape."""
    rng = np.random.default_rng(1)
    X = np.asfortranarray(rng.uniform(size=(3, 10)))
    y = rng.uniform(size=10)
    Z = np.asfortranarray(X[:, :5] + 0.0)
    k = O.SeIsoKernel(0.0, 0.0)
    out = O.evaluate(k, Z, X, y, 1.0)
    eps, tol = 1e-8, 1e-2
    base = out["l"]
    l_s2 = O.evaluate(k, Z, X, y, 1.0 + eps, want_grad=False)["l"]
    assert abs((l_s2 - base) / eps - out["dl_dsigma2"]) <= tol
    for i, h in enumerate(out["hypers"]):
        if h[0] == "log_ell":
            k2, Z2 = O.SeIsoKernel(eps, 0.0), Z
        elif h[0] == "log_sf2":
            k2, Z2 = O.SeIsoKernel(0.0, eps), Z
        else:
            k2, Z2 = k, Z.copy()
            Z2[h[2] - 1, h[1] - 1] += eps
        l2 = O.evaluate(k2, Z2, X, y, 1.0, want_grad=False)["l"]
        assert abs((l2 - base) / eps - out["grad"][i]) <= tol, h


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixtures_reproduce(name):
    g = load_golden(name)
    k = oracle_kernel(g)
    out = O.evaluate(k, g["Z"], g["X"], g["y"], float(g["sigma2"]), variational=bool(g["variational"]))
    for key in ("l1", "l2", "l", "dl_dsigma2"):
        assert abs(out[key] - float(g[key])) <= 1e-9 * max(1.0, abs(float(g[key]))), key
    assert relinf(out["grad"], g["grad"]) < 1e-8
    assert relinf(out["coeffs"], g["coeffs"]) < 1e-8
    fast = O.evaluate_fast(k, g["Z"], g["X"], g["y"], float(g["sigma2"]), variational=bool(g["variational"]))
    assert relinf(fast["grad"], g["grad"]) < 1e-9


def test_prediction_against_dense_fitc_posterior():
    """Means/Variances (lib/fitc_gp.ml:418-425, :498-518) against the textbook FITC predictive equations
    mean = k*^T B^-1 K_mn S^-1 y,  var = k** - k*^T (K_m^-1 - B^-1) k*."""
    X, y, Z = synth(13, 150, 9, 2)
    k = O.SeIsoKernel(0.3, 0.1)
    out = O.evaluate(k, Z, X, y, 0.2, want_grad=False, keep=True)
    rng = np.random.default_rng(0)
    Xt = np.asfortranarray(rng.normal(size=(2, 17)))
    mean = O.predict_means(k, Z, out["coeffs"], Xt)
    var = O.predict_variances(k, Z, out["model"], Xt, predictive=False)
    km, _ = O.spec_calc_shared_upper(k, Z)
    km = np.triu(km) + np.triu(km, 1).T + O.CHOLESKY_JITTER * np.eye(9)
    knm, _ = O.spec_calc_shared_cross(k, X, Z)
    ktm, _ = O.spec_calc_shared_cross(k, Xt, Z)
    s = k.sf2 - np.einsum("ij,ji->i", knm, np.linalg.solve(km, knm.T)) + 0.2
    B = km + (knm / s[:, None]).T @ knm
    mean_ref = ktm @ np.linalg.solve(B, knm.T @ (y / s))
    var_ref = k.sf2 - np.einsum("ij,ji->i", ktm, (np.linalg.inv(km) - np.linalg.inv(B)) @ ktm.T)
    assert np.max(np.abs(mean - mean_ref)) < 1e-9 * np.max(np.abs(mean_ref))
    assert np.max(np.abs(var - var_ref)) < 1e-7
    assert np.allclose(O.predict_variances(k, Z, out["model"], Xt), var + 0.2)


def test_hyper_order():
    assert O.se_iso_hypers(2, 2) == [("log_ell",), ("log_sf2",), ("inducing", 1, 1), ("inducing", 1, 2),
                                     ("inducing", 2, 1), ("inducing", 2, 2)]      # lib/cov_se_iso.ml:188-202
    k = O.SeFatKernel(2, 0.0, np.ones((3, 2)))
    hs = O.se_fat_hypers(k, 1)
    assert hs[:3] == [("log_sf2",), ("inducing", 1, 1), ("inducing", 1, 2)]
    assert hs[3:] == [("proj", b, s) for b in (1, 2, 3) for s in (1, 2)]           # lib/cov_se_fat.ml:318-326


def test_error_behaviour():
    X, y, Z = synth(0, 20, 3, 2)
    k = O.SeIsoKernel(0.0, 0.0)
    with pytest.raises(ValueError, match="sigma2 < 0"):      # lib/fitc_gp.ml:148-149
        O.evaluate(k, Z, X, y, -1.0)
    with pytest.raises(ValueError, match="targets"):         # lib/fitc_gp.ml:283-284
        O.evaluate(k, Z, X, y[:-1], 0.1)
    Zdup = np.asfortranarray(np.repeat(Z[:, :1], 3, axis=1))
    # duplicate inducing points are only saved by the jitter (lib/utils.ml:35): still factorises
    O.evaluate(k, Zdup, X, y, 0.1, want_grad=False)


def test_covariances_sampler_and_stats_against_dense_posterior():
    """FITC_/FIC_covariances, Cov_sampler and Stats (lib/fitc_gp.ml:304-374, :533-697) against the textbook
    FITC posterior  cov = K** - K*m (K_m^-1 - B^-1) Km*  and direct definitions."""
    X, y, Z = synth(17, 160, 8, 2)
    k = O.SeIsoKernel(0.2, -0.1)
    out = O.evaluate(k, Z, X, y, 0.3, want_grad=False, keep=True)
    rng = np.random.default_rng(3)
    Xt = np.asfortranarray(rng.normal(size=(2, 11)))
    km, _ = O.spec_calc_shared_upper(k, Z)
    km = np.triu(km) + np.triu(km, 1).T + O.CHOLESKY_JITTER * np.eye(8)
    knm, _ = O.spec_calc_shared_cross(k, X, Z)
    ktm, _ = O.spec_calc_shared_cross(k, Xt, Z)
    s = k.sf2 - np.einsum("ij,ji->i", knm, np.linalg.solve(km, knm.T)) + 0.3
    B = km + (knm / s[:, None]).T @ knm
    ktt = O.spec_inputs_calc_upper(k, Xt)
    ktt = np.triu(ktt) + np.triu(ktt, 1).T
    ref = ktt - ktm @ (np.linalg.inv(km) - np.linalg.inv(B)) @ ktm.T
    cov = O.fitc_covariances(k, Z, out["model"], Xt)
    assert np.max(np.abs(cov - np.triu(ref))) < 1e-7
    # its diagonal is Variances.calc (Common_covariances.get_variances, lib/fitc_gp.ml:564-565)
    assert np.allclose(np.diag(cov), O.predict_variances(k, Z, out["model"], Xt, predictive=False), atol=1e-10)
    # FIC: Q_t Q_t^T + diag(k** - rowsum(K_tm^2)) exactly as lib/fitc_gp.ml:603-627 writes it
    fic = O.fic_covariances(k, Z, out["model"], Xt)
    fic_ref = ktm @ np.linalg.inv(B) @ ktm.T + np.diag(k.sf2 - np.sum(ktm * ktm, axis=1))
    assert np.max(np.abs(fic - np.triu(fic_ref))) < 1e-7
    assert np.allclose(np.diag(O.covariances_get(cov, 0.3)), np.diag(cov) + 0.3)
    # sampler: chol^T chol == cov + sigma2 + jitter; with z = I the samples are means + rows of chol
    means = O.predict_means(k, Z, out["coeffs"], Xt)
    smp = O.cov_sampler_calc(means, cov, 0.3, predictive=True)
    u = np.triu(smp["cov_chol"])
    full = np.triu(cov) + np.triu(cov, 1).T + (0.3 + O.CHOLESKY_JITTER) * np.eye(11)
    assert np.max(np.abs(u.T @ u - full)) < 1e-12
    S = O.cov_sampler_samples(smp, np.eye(11))
    assert np.max(np.abs(S - (u.T + means[:, None]))) < 1e-14
    # stats
    tm = knm @ out["coeffs"]
    st = O.stats_calc(y, tm, out["l"])
    assert st["n_samples"] == 160 and abs(st["mse"] - np.mean((y - tm) ** 2)) < 1e-14
    assert abs(st["smse"] - st["mse"] / np.mean(y * y)) < 1e-14
    assert abs(st["msll"] - (-0.5 * np.log(2 * np.pi * np.mean(y * y)) - 0.5 - out["l"] / 160)) < 1e-14
    assert abs(st["mad"] - np.mean(np.abs(y - tm))) < 1e-14 and st["maxad"] == np.max(np.abs(y - tm))


@pytest.mark.parametrize("name", posterior_golden_names())
def test_posterior_golden_fixtures_reproduce(name):
    g = load_golden(name)
    k = oracle_kernel(g)
    s2 = float(g["sigma2"])
    out = O.evaluate(k, g["Z"], g["X"], g["y"], s2, want_grad=False, keep=True)
    assert abs(out["l"] - g["l"]) < 1e-12 * abs(g["l"])
    means = O.predict_means(k, g["Z"], out["coeffs"], g["Xt"])
    assert relinf(means, g["means"]) < 1e-12
    fitc = O.fitc_covariances(k, g["Z"], out["model"], g["Xt"])
    assert relinf(fitc, g["fitc_cov"]) < 1e-12
    assert relinf(O.fic_covariances(k, g["Z"], out["model"], g["Xt"]), g["fic_cov"]) < 1e-12
    smp = O.cov_sampler_calc(means, fitc, s2, predictive=True)
    assert relinf(O.cov_sampler_samples(smp, g["z"]), g["samples"]) < 1e-11
    knm, _ = O.spec_calc_shared_cross(k, g["X"], g["Z"])
    st = O.stats_calc(g["y"], knm @ out["coeffs"], out["l"])
    assert relinf([st[key] for key in STAT_KEYS], g["stats"]) < 1e-12


@pytest.mark.parametrize("shape", [(200, 8, 1), (150, 7, 3)])
def test_against_snelson_spgp_lik(shape):
    """test/oct.m:183-191 compares the reference with Edward Snelson's SPGP routine (test/spgp_lik.m), which the
    reference ships in its test directory.  The same comparison here, for the oracle: evidence, d/dlog_ell,
    d/dlog_sf2, d/dsigma2 (oct.m's eds_* mappings) and, beyond oct.m, every pseudo-input coordinate."""
    from oracle.snelson_spgp import spgp_lik
    n, m, d = shape
    X, y, Z = synth(23, n, m, d)
    log_ell, log_sf2, sigma2 = 0.3, -0.2, 0.4
    out = O.evaluate(O.SeIsoKernel(log_ell, log_sf2), Z, X, y, sigma2)
    hyp = np.concatenate([np.full(d, -2.0 * log_ell), [log_sf2, np.log(sigma2)]])     # oct.m:185 (log_inv_ell2)
    ew = np.concatenate([Z.T.ravel(order="F"), hyp])                                  # oct.m:186
    fw, dfw = spgp_lik(ew, y, np.ascontiguousarray(X.T), m)
    assert abs(out["l"] - (-fw)) < 1e-10 * abs(fw)                                    # eds_evidence
    dfxb = dfw[:m * d].reshape(m, d, order="F")
    dfb, dfc, dfsig = dfw[m * d:m * d + d], dfw[-2], dfw[-1]
    g = out["grad"]
    scale = np.max(np.abs(g))
    assert abs(g[0] - 2.0 * np.sum(dfb)) < 1e-8 * scale                               # eds_dlog_ell = 2 dfw(end-2)
    assert abs(g[1] - (-dfc)) < 1e-8 * scale                                          # eds_dlog_sf2
    assert abs(out["dl_dsigma2"] - (-dfsig / sigma2)) < 1e-8 * abs(out["dl_dsigma2"])  # eds_dsigma2
    assert np.max(np.abs(g[2:].reshape(m, d) - (-dfxb))) < 1e-8 * scale               # pseudo-inputs


def test_c_restatement_agrees_with_numpy_oracle():
    """oracle/fitc_ref.c (the reference's LAPACK sequence in C, also the timed CPU baseline) against
    oracle/fitc_oracle.py on the C1 shape and on a ragged one: two independent restatements of lib/fitc_gp.ml."""
    from oracle import fitc_ref as R
    for seed, n, m, d, le, lsf, s2 in ((1, 2000, 50, 3, 0.5 * np.log(3), 0.0, 0.1), (7, 777, 33, 5, 0.9, -0.3, 0.02)):
        rng = np.random.default_rng(seed)
        X = np.asfortranarray(rng.normal(size=(d, n)))
        y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
        Z = np.asfortranarray(X[:, rng.permutation(n)[:m]] + 0.01 * rng.normal(size=(d, m)))
        ref = O.evaluate(O.SeIsoKernel(le, lsf), Z, X, y, s2)
        for threads in (1, 4):
            got = R.iso_eval(X, y, Z, le, lsf, s2, threads=threads)
            assert abs(got["l"] - ref["l"]) <= 1e-10 * abs(ref["l"])
            assert abs(got["l1"] - ref["l1"]) <= 1e-10 * abs(ref["l1"])
            assert abs(got["dl_dsigma2"] - ref["dl_dsigma2"]) <= 1e-9 * abs(ref["dl_dsigma2"])
            assert np.max(np.abs(got["grad"] - ref["grad"])) <= 1e-9 * np.max(np.abs(ref["grad"]))
            assert np.max(np.abs(got["coeffs"] - ref["coeffs"])) <= 1e-9 * np.max(np.abs(ref["coeffs"]))


@pytest.mark.parametrize("name", illcond_golden_names())
def test_ill_conditioned_fixtures(name):
    """Jitter-dominated K_m (ell = e, cond ~ 1e7): the fixtures reproduce, and where a 40-digit evaluation is stored
    the oracle's QR path (lib/fitc_gp.ml:170-182) stays within 1e-10 of it on l1 and l2 -- the accuracy the device path's
    whitened SYRK + potrf has to match (tests/test_gpu_parity.py::test_ill_conditioned_regime)."""
    g = load_golden(name)
    out = O.evaluate(oracle_kernel(g), g["Z"], g["X"], g["y"], float(g["sigma2"]))
    assert abs(out["l"] - float(g["l"])) <= 1e-12 * abs(float(g["l"]))
    assert relinf(out["grad"], g["grad"]) <= 1e-10
    if "mp_l1" in g:
        assert abs(out["l1"] - float(g["mp_l1"])) <= 1e-10 * abs(float(g["mp_l1"]))
        assert abs(out["l2"] - float(g["mp_l2"])) <= 1e-10 * abs(float(g["mp_l2"]))
        assert relinf(out["coeffs"], g["mp_coeffs"]) <= 1e-9


def test_calc_model_inputs_restatements_against_dense_formulas():
    """Variances.calc_model_inputs / FITC_ / FIC_covariances.calc_model_inputs (lib/fitc_gp.ml:487-496, :569-579,
    :609-614) from dense textbook expressions: variances equal Variances.calc at the training inputs; the covariance
    forms carry the model's Q factor, Q_n = diag(sqrt(1/s)) K_nm R^-1 with R^T R = B = K_m + K_mn S^-1 K_nm."""
    rng = np.random.default_rng(12)
    n, m, d = 60, 9, 2
    X = np.asfortranarray(rng.normal(size=(d, n)))
    Z = np.asfortranarray(X[:, :m] + 0.05 * rng.normal(size=(d, m)))
    y = rng.normal(size=n)
    k = O.SeIsoKernel(0.2, 0.3)
    s2 = 0.3
    ref = O.evaluate(k, Z, X, y, s2, want_grad=False, keep=True)
    mod = ref["model"]
    var = O.variances_model_inputs(mod)
    assert np.allclose(var, O.predict_variances(k, Z, mod, X, predictive=False), rtol=1e-10, atol=1e-12)
    knm, _ = O.spec_calc_shared_cross(k, X, Z)
    km = np.triu(mod["inducing"]["km"]) + np.triu(mod["inducing"]["km"], 1).T + 1e-6 * np.eye(m)
    qn = knm @ np.linalg.inv(km)
    r = k.sf2 - np.einsum("ij,ij->i", qn, knm)
    s = r + s2
    B = km + knm.T @ (knm / s[:, None])
    dq = (knm / np.sqrt(s)[:, None])
    qq = dq @ np.linalg.solve(B, dq.T)
    kn = np.exp(k.log_sf2 - 0.5 * np.exp(-2 * k.log_ell) * ((X.T[:, None, :] - X.T[None, :, :]) ** 2).sum(-1))
    fitc = kn - qn @ knm.T + qq
    got = O.fitc_covariances_model_inputs(k, mod, X)
    assert np.max(np.abs(got - np.triu(fitc))) <= 1e-9
    fic = qq + np.diag(r)
    assert np.max(np.abs(O.fic_covariances_model_inputs(mod) - np.triu(fic))) <= 1e-9


@pytest.mark.parametrize("n,m,d,log_ell", [(900, 60, 1, -0.05), (1200, 150, 3, 0.5), (700, 90, 8, 1.1)])
def test_oracle_against_an_80_bit_evaluation(n, m, d, log_ell):
    """The oracle's evidence and mean coefficients against textbook formulas evaluated in x87 long double
    (tests/util.py::longdouble_fitc -- no LAPACK, none of the reference's operation sequence): a pin that does not share
    the oracle's arithmetic.  d = 1 with a length scale near 1 is the jitter-dominated regime (cond(K_m + jitter) ~ 1e9),
    where the coefficients carry cond * eps."""
    from tests.util import longdouble_fitc, synth
    X, y, Z = synth(900 + n, n, m, d)
    k = O.SeIsoKernel(log_ell, 0.2)
    ref = O.evaluate(k, Z, X, y, 0.07, want_grad=False)
    Xt = np.asfortranarray(np.random.default_rng(5).normal(size=(d, 64)))
    ref = O.evaluate(k, Z, X, y, 0.07, want_grad=False, keep=True)
    l, t, mean, var = longdouble_fitc(X, y, Z, log_ell, 0.2, 0.07, Xt=Xt)
    assert abs(ref["l"] - l) <= 1e-11 * abs(l)
    # posterior at new inputs (Means.calc / Variances.calc): measured 1e-10 .. 1e-14 / 1e-9 .. 1e-15
    assert np.max(np.abs(O.predict_means(k, Z, ref["coeffs"], Xt) - mean)) <= 1e-9 * max(np.max(np.abs(mean)), 1e-3)
    assert np.max(np.abs(O.predict_variances(k, Z, ref["model"], Xt, predictive=False) - var)) <= 1e-8 * k.sf2
    # measured: l 4e-14 / 2e-13 / 1e-15, coefficients 1.0e-9 / 1.1e-9 / 1.8e-13
    assert np.max(np.abs(ref["coeffs"] - t)) <= (1e-8 if d <= 3 else 1e-10) * np.max(np.abs(t))


def test_oracle_gradient_against_80_bit_central_differences():
    """Selected entries of the oracle's gradient (length scale, amplitude, noise, inducing coordinates: the reference's
    per-hyper trace formulas, lib/fitc_gp.ml:943-1021) against central differences of the 80-bit evidence: with
    h = 1e-6 in long double the difference quotient is good to ~1e-11 of the gradient's scale, five orders below what
    an fp64 difference quotient resolves."""
    from tests.util import longdouble_fitc, synth
    n, m, d = 500, 40, 3
    X, y, Z = synth(77, n, m, d)
    le, lsf, s2 = 0.45, -0.1, 0.08
    ref = O.evaluate(O.SeIsoKernel(le, lsf), Z, X, y, s2)
    LD = np.longdouble
    h = LD(1e-6)

    def l_of(dle=0, dlsf=0, ds2=0, dz=None):
        Zp = np.asarray(Z, LD).copy()
        if dz is not None:
            Zp[dz[0], dz[1]] += dz[2]
        # (longdouble_fitc converts its inputs itself; pass longdouble scalars through)
        return _ld_l(X, y, Zp, LD(le) + dle, LD(lsf) + dlsf, LD(s2) + ds2)

    def _ld_l(X, y, Zp, a, b, c):
        import tests.util as U
        Xl, yl = np.asarray(X, LD), np.asarray(y, LD)
        ie, sf2 = np.exp(LD(-2) * a), np.exp(b)

        def cov(A, B):
            return sf2 * np.exp(LD(-0.5) * ie * ((A.T[:, None, :] - B.T[None, :, :]) ** 2).sum(-1))
        Um = U._ld_chol_upper(cov(Zp, Zp) + LD(1e-6) * np.eye(m, dtype=LD))
        K = cov(Xl, Zp)
        V = np.zeros_like(K)
        for j in range(m):
            V[:, j] = (K[:, j] - V[:, :j] @ Um[:j, j]) / Um[j, j]
        s = sf2 - (V * V).sum(1) + c
        R = U._ld_chol_upper(np.eye(m, dtype=LD) + V.T @ (V / s[:, None]))
        cc = V.T @ (yl / s)
        bb = np.zeros_like(cc)
        for i in range(m):
            bb[i] = (cc[i] - np.dot(R[:i, i], bb[:i])) / R[i, i]
        return (LD(-0.5) * (2 * np.sum(np.log(np.diag(R))) + np.sum(np.log(s)) + n * np.log(2 * LD(np.pi)))
                - LD(0.5) * (np.dot(yl, yl / s) - np.dot(bb, bb)))

    l0, _ = longdouble_fitc(X, y, Z, le, lsf, s2)
    assert abs(float(l_of()) - l0) <= 1e-15 * abs(l0)
    scale = np.max(np.abs(ref["grad"]))
    g_le = float((l_of(dle=h) - l_of(dle=-h)) / (2 * h))
    g_sf = float((l_of(dlsf=h) - l_of(dlsf=-h)) / (2 * h))
    g_s2 = float((l_of(ds2=h) - l_of(ds2=-h)) / (2 * h))
    assert abs(ref["grad"][0] - g_le) <= 1e-9 * scale          # Hyper.get_all order: Log_ell, Log_sf2, inducing ...
    assert abs(ref["grad"][1] - g_sf) <= 1e-9 * scale
    assert abs(ref["dl_dsigma2"] - g_s2) <= 1e-9 * max(abs(g_s2), 1.0)
    for ind, dim in ((0, 0), (7, 2), (39, 1)):                 # Inducing_hyper {ind; dim}: index 2 + ind * d + dim
        g = float((l_of(dz=(dim, ind, h)) - l_of(dz=(dim, ind, -h))) / (2 * h))
        assert abs(ref["grad"][2 + ind * d + dim] - g) <= 1e-9 * scale, (ind, dim)


def test_oracle_fat_gradient_against_80_bit_central_differences():
    """Cov_se_fat with projection + heteroskedastic noise + multiscales: one entry of every hyper family of the oracle's
    gradient (Log_sf2, Inducing_hyper, Proj, Log_hetero_skedasticity, Log_multiscale_m05: the derivative formulas of
    lib/cov_se_fat.ml:418-641 through lib/fitc_gp.ml:943-1021) against central differences of the covariance *values*
    evaluated in 80-bit arithmetic (tests/util.py::longdouble_fat_evidence) -- the derivative code is not involved in
    the reference value, and h = 1e-6 in long double resolves ~1e-10 of the gradient's scale."""
    from tests.util import longdouble_fat_evidence
    LD = np.longdouble
    rng = np.random.default_rng(21)
    n, m, D, d = 400, 30, 4, 2
    X = np.asfortranarray(rng.normal(size=(D, n)))
    y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
    P = np.asfortranarray(rng.normal(size=(D, d)) / np.sqrt(D * d))
    Z = np.asfortranarray((P.T @ X)[:, :m] + 0.05 * rng.normal(size=(d, m)))
    het = rng.uniform(-5, -3, size=m)
    lms = np.asfortranarray(rng.uniform(-0.5, 0.5, size=(d, m)))
    lsf, s2 = 0.15, 0.06
    k = O.SeFatKernel(d, lsf, P, het, lms)
    ref = O.evaluate(k, Z, X, y, s2)
    hyp = O.se_fat_hypers(k, m)
    assert len(hyp) == ref["grad"].shape[0]
    l0 = longdouble_fat_evidence(X, y, Z, lsf, s2, P, het, lms)
    assert abs(float(l0) - ref["l"]) <= 1e-11 * abs(ref["l"])
    h = LD(1e-6)
    scale = np.max(np.abs(ref["grad"]))

    def fd(**kw):
        def ev(sign):
            a = dict(Z=np.asarray(Z, LD).copy(), lsf=LD(lsf), s2=LD(s2), P=np.asarray(P, LD).copy(),
                     het=np.asarray(het, LD).copy(), lms=np.asarray(lms, LD).copy())
            for key, idx in kw.items():
                if idx is None:
                    a[key] = a[key] + sign * h
                else:
                    a[key][idx] += sign * h
            return longdouble_fat_evidence(X, y, a["Z"], a["lsf"], a["s2"], a["P"], a["het"], a["lms"])
        return float((ev(1) - ev(-1)) / (2 * h))

    checks = [(("log_sf2",), fd(lsf=None)),
              (("inducing", 5, 2), fd(Z=(1, 4))),            # hyper indices are 1-based, Fortran (dim, ind) storage
              (("proj", 3, 1), fd(P=(2, 0))),
              (("log_hetero", 11), fd(het=(10,))),
              (("log_multiscale", 7, 2), fd(lms=(1, 6)))]
    for name, g in checks:
        got = ref["grad"][hyp.index(name)]
        assert abs(got - g) <= 1e-8 * scale, (name, got, g)
    assert abs(ref["dl_dsigma2"] - fd(s2=None)) <= 1e-8 * max(1.0, abs(ref["dl_dsigma2"]))
