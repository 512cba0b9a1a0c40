import glob
import os

import numpy as np

from oracle import fitc_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_names():
    """Evidence + gradient fixtures."""
    return [n for n in _names() if not n.startswith("posterior_") and not n.startswith("illcond_")]


def illcond_golden_names():
    """Jitter-dominated K_m (ell = e): the regime SURVEY.md 7 singles out; own stated tolerances."""
    return [n for n in _names() if n.startswith("illcond_")]


def posterior_golden_names():
    """Prediction / covariance / sampler / stats fixtures (make_golden.save_posterior)."""
    return [n for n in _names() if n.startswith("posterior_")]


STAT_KEYS = ("n_samples", "target_variance", "sse", "mse", "rmse", "smse", "msll", "mad", "maxad")


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    g["kind"] = str(g["kind"])
    return g


def oracle_kernel(g):
    if g["kind"] == "iso":
        return O.SeIsoKernel(float(g["log_ell"]), float(g["log_sf2"]))
    return O.SeFatKernel(int(g["d"]), float(g["log_sf2"]), g.get("tproj"), g.get("log_hetero"), g.get("log_multiscales"))


def synth(seed, n, m, d):
    """BASELINE.md section 2 synthetic generator."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(d, n))
    y = np.sin(X.sum(0)) + 0.1 * rng.normal(size=n)
    Z = X[:, rng.permutation(n)[:m]] + 0.01 * rng.normal(size=(d, m))
    return np.asfortranarray(X), y, np.asfortranarray(Z)


def relinf(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


# ---- an 80-bit (x87 long double) evaluation of the FITC mean coefficients and evidence from the textbook formulas:
# dense Cholesky / substitution loops in numpy longdouble (eps 1.1e-19), independent of LAPACK and of the oracle's
# operation sequence.  Cov_se_iso only; a few thousand points, a few hundred inducing points.
def _ld_chol_upper(A):
    m = A.shape[0]
    U = np.zeros_like(A)
    for j in range(m):
        U[j, j] = np.sqrt(A[j, j] - np.dot(U[:j, j], U[:j, j]))
        U[j, j + 1:] = (A[j, j + 1:] - U[:j, j] @ U[:j, j + 1:]) / U[j, j]
    return U


def longdouble_fitc(X, y, Z, log_ell, log_sf2, sigma2, jitter=1e-6, Xt=None):
    """Returns (l, t) in float64 -- with test inputs Xt also the posterior means K_tm t and variances
    sf2 - |K_tm U^-1|^2 + |K_tm R^-1|^2, R = R~ U (lib/fitc_gp.ml:418-425, :498-518) -- from an evaluation in numpy longdouble: U = chol(K_m + jitter I), V = K_nm U^-1,
    s = sf2 - |V_i|^2 + sigma2, B~ = I + V^T S^-1 V = R~^T R~, t = U^-1 R~^-1 R~^-T V^T (y / s),
    l = -1/2 (log|B~| + sum log s + n log 2 pi) - 1/2 (y^T S^-1 y - |R~^-T V^T (y/s)|^2)."""
    LD = np.longdouble
    assert np.finfo(LD).eps < 1e-18, "numpy longdouble is not an extended type here"
    Xl, Zl, yl = np.asarray(X, LD), np.asarray(Z, LD), np.asarray(y, LD)
    n, m = Xl.shape[1], Zl.shape[1]
    ie, sf2 = np.exp(LD(-2) * LD(log_ell)), np.exp(LD(log_sf2))

    def cov(A, B):
        return sf2 * np.exp(LD(-0.5) * ie * ((A.T[:, None, :] - B.T[None, :, :]) ** 2).sum(-1))

    U = _ld_chol_upper(cov(Zl, Zl) + LD(jitter) * np.eye(m, dtype=LD))
    K = cov(Xl, Zl)
    V = np.zeros_like(K)
    for j in range(m):
        V[:, j] = (K[:, j] - V[:, :j] @ U[:j, j]) / U[j, j]
    s = sf2 - (V * V).sum(1) + LD(sigma2)
    R = _ld_chol_upper(np.eye(m, dtype=LD) + V.T @ (V / s[:, None]))
    c = V.T @ (yl / s)
    b = np.zeros_like(c)
    for i in range(m):
        b[i] = (c[i] - np.dot(R[:i, i], b[:i])) / R[i, i]
    tt = np.zeros_like(c)
    for i in range(m - 1, -1, -1):
        tt[i] = (b[i] - np.dot(R[i, i + 1:], tt[i + 1:])) / R[i, i]
    t = np.zeros_like(c)
    for i in range(m - 1, -1, -1):
        t[i] = (tt[i] - np.dot(U[i, i + 1:], t[i + 1:])) / U[i, i]
    l = (LD(-0.5) * (2 * np.sum(np.log(np.diag(R))) + np.sum(np.log(s)) + n * np.log(2 * LD(np.pi)))
         - LD(0.5) * (np.dot(yl, yl / s) - np.dot(b, b)))
    if Xt is None:
        return float(l), t.astype(np.float64)
    Kt = cov(np.asarray(Xt, LD), Zl)
    Vt = np.zeros_like(Kt)
    for j in range(m):
        Vt[:, j] = (Kt[:, j] - Vt[:, :j] @ U[:j, j]) / U[j, j]
    Qt = np.zeros_like(Kt)
    for j in range(m):   # Q_t = V_t R~^-1 = K_tm R^-1
        Qt[:, j] = (Vt[:, j] - Qt[:, :j] @ R[:j, j]) / R[j, j]
    var = sf2 - (Vt * Vt).sum(1) + (Qt * Qt).sum(1)
    return float(l), t.astype(np.float64), (Kt @ t).astype(np.float64), var.astype(np.float64)


def longdouble_fat_evidence(X, y, Z, log_sf2, sigma2, tproj=None, log_hetero=None, log_multiscales_m05=None, jitter=1e-6):
    """FITC log evidence of Cov_se_fat in numpy longdouble (all inputs may be longdouble arrays): values only --
    k(p, z_c) = sf2 exp(-1/2 sum_k [(p_k - z_kc)^2 / ms_kc + log ms_kc]) with p = tproj^T x, ms = exp(log_ms) + 1/2
    (1 without multiscales); K_m off-diagonal with scale ms_kr + ms_kc - 1, diagonal sf2 exp(-1/2 sum_k log(2 ms_kc - 1))
    + exp(log_hetero_c) + jitter  (lib/cov_se_fat.ml:62-75, :85-142, :224-252).  Used to difference numerically."""
    LD = np.longdouble
    Xl, Zl, yl = np.asarray(X, LD), np.asarray(Z, LD), np.asarray(y, LD)
    d, m = Zl.shape
    n = Xl.shape[1]
    lsf = LD(log_sf2)
    P = Xl if tproj is None else np.asarray(tproj, LD).T @ Xl
    ms = None if log_multiscales_m05 is None else np.exp(np.asarray(log_multiscales_m05, LD)) + LD(0.5)
    acc = np.zeros((n, m), LD)
    for k in range(d):
        diff = P[k, :][:, None] - Zl[k, :][None, :]
        acc += diff * diff if ms is None else diff * (diff / ms[k, :][None, :]) + np.log(ms[k, :][None, :])
    K = np.exp(lsf - LD(0.5) * acc)
    accm = np.zeros((m, m), LD)
    for k in range(d):
        diff = Zl[k, :][:, None] - Zl[k, :][None, :]
        if ms is None:
            accm += diff * diff
        else:
            sc = ms[k, :][:, None] + ms[k, :][None, :] - LD(1)
            accm += diff * (diff / sc) + np.log(sc)
    Km = np.exp(lsf - LD(0.5) * accm)
    dg = np.full(m, np.exp(lsf)) if ms is None else np.exp(lsf - LD(0.5) * np.log(2 * ms - 1).sum(0))
    if log_hetero is not None:
        dg = dg + np.exp(np.asarray(log_hetero, LD))
    Km[np.diag_indices(m)] = dg + LD(jitter)
    U = _ld_chol_upper(Km)
    V = np.zeros_like(K)
    for j in range(m):
        V[:, j] = (K[:, j] - V[:, :j] @ U[:j, j]) / U[j, j]
    s = np.exp(lsf) - (V * V).sum(1) + LD(sigma2)          # calc_diag = sf2 (lib/cov_se_fat.ml:222)
    R = _ld_chol_upper(np.eye(m, dtype=LD) + V.T @ (V / s[:, None]))
    c = V.T @ (yl / s)
    b = np.zeros_like(c)
    for i in range(m):
        b[i] = (c[i] - np.dot(R[:i, i], b[:i])) / R[i, i]
    return (LD(-0.5) * (2 * np.sum(np.log(np.diag(R))) + np.sum(np.log(s)) + n * np.log(2 * LD(np.pi)))
            - LD(0.5) * (np.dot(yl, yl / s) - np.dot(b, b)))
