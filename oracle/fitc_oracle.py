"""CPU oracle for the FITC nLML + hyper-gradient path of mmottl/gpr.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path (gpr_amd/) never does.

This is a numpy / scipy-LAPACK restatement of the reference's *own* operation
sequence (direct-difference covariance loops -> dpotrf -> dtrsm -> stacked
dgeqrf/dorgqr -> dpotri x2 -> dtrsm x2 -> dsyr/dsyrk x2 -> per-hyper traces).
Every function cites the reference file:line it follows (paths relative to the
reference root).  Matrices are Fortran-ordered like the reference's Bigarrays:
inputs d x n (one point per column), inducing d x m, knm n x m.

PARITY UNPINNED BY REFERENCE FIXTURES: the reference ships no golden vectors
(its test/ executables print values, `dune runtest` runs nothing) and no OCaml
toolchain exists in the build image, so the reference itself cannot be run.
The oracle is pinned instead by (see tests/test_oracle.py):
  * Edward Snelson's SPGP routine test/spgp_lik.m -- the third-party code the
    reference keeps in its test directory and cross-checks itself against in
    test/oct.m:183-191 -- restated in oracle/snelson_spgp.py: evidence and the
    full gradient (length scale, amplitude, noise, every pseudo-input),
  * the algebraic identity  l1+l2 == dense textbook FITC log-likelihood,
  * central finite differences of that dense likelihood for every hyper,
  * an mpmath 50-digit evaluation at tiny n,
  * 80-bit (x87 long double) evaluations of the textbook formulas at n ~ 1000 (tests/util.py::longdouble_fitc,
    longdouble_fat_evidence): evidence, mean coefficients, and -- by central differences of the 80-bit evidence --
    gradient entries of every hyper family of both kernels (tests/test_oracle.py::test_oracle_*80_bit*),
  * the formulas of the reference's Octave cross-check test/oct.m:88-180,
  * the reference's own gradient self-test recipe (lib/fitc_gp.ml:1223-1462).

Lacaml semantics assumed (Lacaml is not vendored in the reference tree):
  Mat.syrk_diag ~alpha a ~beta ~y : y <- alpha*diag(a a^T) + beta*y
  Mat.symm2_trace a b             : tr(a b), a and b symmetric, upper stored
  Mat.gemm_trace ~transa:`T a b   : tr(a^T b) = sum(a .* b)
  potri (on a potrf'd upper factor U) : upper triangle of (U^T U)^-1
    -- lib/block_diag.mli:39-41 documents exactly this use ("using its already
    precomputed Cholesky factor"), and doc/manual/gpr_manual.tex:710 requires
    T = K_m^-1 - B^-1.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
from scipy.linalg import blas, lapack

CHOLESKY_JITTER = 1e-6  # lib/utils.ml:35
LOG_2PI = math.log(2.0 * (4.0 * math.atan(1.0)))  # lib/utils.ml:39-40


class NotPositiveDefinite(RuntimeError):
    """Mirrors Lacaml's Failure on potrf info > 0."""


def _F(a):
    return np.asfortranarray(a, dtype=np.float64)


# ---------------------------------------------------------------------------
# Cov_se_iso  (lib/cov_se_iso.ml)
# ---------------------------------------------------------------------------
@dataclass
class SeIsoKernel:
    """lib/cov_se_iso.ml:33-44 (Eval.Kernel.t / create)."""

    log_ell: float
    log_sf2: float
    inv_ell2: float = field(init=False)
    inv_ell2_05: float = field(init=False)
    sf2: float = field(init=False)

    def __post_init__(self):
        self.inv_ell2 = math.exp(-2.0 * self.log_ell)
        self.inv_ell2_05 = -0.5 * self.inv_ell2
        self.sf2 = math.exp(self.log_sf2)


def _sqr_diff_cross(inputs, inducing):
    """lib/cov_se_iso.ml:128-144: res[r,c] = sum_i (inputs[i,r]-inducing[i,c])^2,
    accumulated over i in increasing order (same rounding as the scalar loop)."""
    d, n = inputs.shape
    m = inducing.shape[1]
    res_t = np.zeros((m, n))  # C-order (m, n) == Fortran (n, m): contiguous along r like the reference
    for i in range(d):
        diff = inputs[i, :][None, :] - inducing[i, :][:, None]
        res_t += diff * diff
    return res_t.T


def se_iso_calc_sqr_diff_upper(inducing):
    """lib/cov_se_iso.ml:56-72.  Note the operand order: inducing[i,c]-inducing[i,r].
    The strict lower triangle is left uninitialised by the reference; here NaN so
    that any consumer reading it is caught."""
    d, m = inducing.shape
    res = np.zeros((m, m), order="F")
    for i in range(d):
        diff = inducing[i, :][None, :] - inducing[i, :][:, None]  # [r,c] = z_c - z_r
        res += diff * diff
    res[np.tril_indices(m, -1)] = np.nan
    return res


def se_iso_calc_upper_with_sqr_diff(k: SeIsoKernel, sqr_diff):
    """lib/cov_se_iso.ml:74-84; diagonal is exactly sf2."""
    res = np.exp(k.log_sf2 + k.inv_ell2_05 * sqr_diff)
    m = res.shape[0]
    res[np.diag_indices(m)] = k.sf2
    return _F(res)


def se_iso_calc_cross_with_sqr_diff(k: SeIsoKernel, sqr_diff):
    """lib/cov_se_iso.ml:146-156."""
    return _F(np.exp(k.log_sf2 + k.inv_ell2_05 * sqr_diff))


def se_iso_hypers(d, m):
    """lib/cov_se_iso.ml:188-202: [Log_ell; Log_sf2; (ind=1,dim=1..d); (ind=2,..); ...]."""
    hypers = [("log_ell",), ("log_sf2",)]
    for ind in range(1, m + 1):
        for dim in range(1, d + 1):
            hypers.append(("inducing", ind, dim))
    return hypers


# ---------------------------------------------------------------------------
# Cov_se_fat, projection-only sub-case  (lib/cov_se_fat.ml)
# ---------------------------------------------------------------------------
@dataclass
class SeFatKernel:
    """lib/cov_se_fat.ml:55-75: projection, heteroskedastic noise on diag(K_m) (:136-142) and
    multiscales (:115-134, :241-251) are all optional."""

    d: int
    log_sf2: float
    tproj: Optional[np.ndarray]  # big_dim x d, or None
    log_hetero_skedasticity: Optional[np.ndarray] = None  # m, or None
    log_multiscales_m05: Optional[np.ndarray] = None      # d x m, or None
    sf2: float = field(init=False)
    hetero_skedasticity: Optional[np.ndarray] = field(init=False, default=None)
    multiscales: Optional[np.ndarray] = field(init=False, default=None)

    def __post_init__(self):
        self.sf2 = math.exp(self.log_sf2)
        if self.log_hetero_skedasticity is not None:
            self.log_hetero_skedasticity = np.asarray(self.log_hetero_skedasticity, dtype=np.float64)
            self.hetero_skedasticity = np.exp(self.log_hetero_skedasticity)  # lib/cov_se_fat.ml:63-65
        if self.log_multiscales_m05 is not None:
            self.log_multiscales_m05 = _F(self.log_multiscales_m05)
            self.multiscales = _F(np.exp(self.log_multiscales_m05) + 0.5)    # lib/cov_se_fat.ml:66-69
        if self.tproj is not None:
            self.tproj = _F(self.tproj)
            if self.tproj.shape[1] != self.d:  # lib/cov_se_fat.ml:38-48
                raise ValueError("Cov_se_fat.Params.create: tproj projection (%d) disagrees "
                                 "with target dimension d (%d)" % (self.tproj.shape[1], self.d))


def se_fat_project(k: SeFatKernel, inputs):
    """lib/cov_se_fat.ml:215-218: gemm ~transa:`T tproj inputs."""
    if k.tproj is None:
        return _F(inputs)
    return _F(blas.dgemm(1.0, k.tproj, _F(inputs), trans_a=1))


def se_fat_calc_upper_vanilla(k: SeFatKernel, mat):
    """lib/cov_se_fat.ml:85-100 (diff = mat[i,r]-mat[i,c]; exp(log_sf2 - 0.5*x))."""
    d, n = mat.shape
    acc = np.zeros((n, n), order="F")
    for i in range(k.d):
        diff = mat[i, :][:, None] - mat[i, :][None, :]
        acc += diff * diff
    res = np.exp(k.log_sf2 - 0.5 * acc)
    res[np.diag_indices(n)] = k.sf2
    res[np.tril_indices(n, -1)] = np.nan
    return _F(res)


def se_fat_calc_upper_multiscale(k: SeFatKernel, inducing):
    """lib/cov_se_fat.ml:115-134: off-diagonal sum_i diff*(diff/scale) + log scale with
    scale = ms[i,r] + ms[i,c] - 1; diagonal sum_i log(2 ms[i,c] - 1)."""
    ms = k.multiscales
    m = inducing.shape[1]
    acc = np.zeros((m, m), order="F")
    for i in range(k.d):
        diff = inducing[i, :][:, None] - inducing[i, :][None, :]
        scale = ms[i, :][:, None] + ms[i, :][None, :] - 1.0
        acc = acc + diff * (diff / scale) + np.log(scale)
    res = np.exp(k.log_sf2 - 0.5 * acc)
    dacc = np.zeros(m)
    for i in range(k.d):
        dacc = dacc + np.log(ms[i, :] + ms[i, :] - 1.0)
    res[np.diag_indices(m)] = np.exp(k.log_sf2 - 0.5 * dacc)
    res[np.tril_indices(m, -1)] = np.nan
    return _F(res)


def se_fat_calc_cross_with_projections(k: SeFatKernel, projections, inducing):
    """lib/cov_se_fat.ml:224-252."""
    acc = np.zeros((projections.shape[1], inducing.shape[1]), order="F")
    for i in range(k.d):
        diff = projections[i, :][:, None] - inducing[i, :][None, :]
        if k.multiscales is None:
            acc += diff * diff
        else:  # update_tmp_sum, lib/cov_se_fat.ml:102-103
            scale = k.multiscales[i, :][None, :]
            acc = acc + diff * (diff / scale) + np.log(scale)
    return _F(np.exp(k.log_sf2 - 0.5 * acc))


def se_fat_hypers(k: SeFatKernel, m):
    """lib/cov_se_fat.ml:290-342: [Log_sf2; inducing (ind-major); Proj (big_dim-major); hetero; multiscale]."""
    hypers = [("log_sf2",)]
    for ind in range(1, m + 1):
        for dim in range(1, k.d + 1):
            hypers.append(("inducing", ind, dim))
    if k.tproj is not None:
        for big in range(1, k.tproj.shape[0] + 1):
            for small in range(1, k.d + 1):
                hypers.append(("proj", big, small))
    if k.hetero_skedasticity is not None:
        for i in range(1, len(k.hetero_skedasticity) + 1):
            hypers.append(("log_hetero", i))
    if k.multiscales is not None:
        for ind in range(1, k.multiscales.shape[1] + 1):
            for dim in range(1, k.d + 1):
                hypers.append(("log_multiscale", ind, dim))
    return hypers


# ---------------------------------------------------------------------------
# Spec dispatch (the two Specs.Deriv instances on the path)
# ---------------------------------------------------------------------------
def spec_calc_shared_upper(k, inducing):
    """Cov_se_iso.Deriv.Inducing.calc_shared_upper lib/cov_se_iso.ml:241-245 /
    Cov_se_fat.Deriv.Inducing.calc_shared_upper lib/cov_se_fat.ml:414-416."""
    inducing = _F(inducing)
    if isinstance(k, SeIsoKernel):
        sq = se_iso_calc_sqr_diff_upper(inducing)
        km = se_iso_calc_upper_with_sqr_diff(k, sq)
        return km, dict(kernel=k, inducing=inducing, sqr_diff_mat=sq, eval_mat=km)
    km = (se_fat_calc_upper_vanilla(k, inducing) if k.multiscales is None
          else se_fat_calc_upper_multiscale(k, inducing))
    if k.hetero_skedasticity is not None:  # lib/cov_se_fat.ml:136-142
        km[np.diag_indices(km.shape[0])] += k.hetero_skedasticity
    return km, dict(kernel=k, inducing=inducing, eval_mat=km)


def spec_calc_shared_cross(k, inputs, inducing):
    """lib/cov_se_iso.ml:290-295 / lib/cov_se_fat.ml:546-554."""
    inputs = _F(inputs)
    inducing = _F(inducing)
    if isinstance(k, SeIsoKernel):
        sq = _sqr_diff_cross(inputs, inducing)
        knm = se_iso_calc_cross_with_sqr_diff(k, sq)
        return knm, dict(kernel=k, inputs=inputs, inducing=inducing, sqr_diff_mat=sq, eval_mat=knm)
    proj = se_fat_project(k, inputs)
    knm = se_fat_calc_cross_with_projections(k, proj, inducing)
    return knm, dict(kernel=k, inputs=inputs, inducing=inducing, projections=proj, eval_mat=knm)


def spec_calc_diag(k, n):
    """lib/cov_se_iso.ml:126 / lib/cov_se_fat.ml:222: constant vector sf2."""
    return np.full(n, k.sf2)


def spec_hypers(k, d_inducing, m):
    if isinstance(k, SeIsoKernel):
        return se_iso_hypers(d_inducing, m)
    return se_fat_hypers(k, m)


def spec_calc_deriv_upper(shared_upper, hyper):
    """lib/cov_se_iso.ml:247-280 / lib/cov_se_fat.ml:418-516 (no hetero/multiscale)."""
    k = shared_upper["kernel"]
    eval_mat = shared_upper["eval_mat"]
    inducing = shared_upper["inducing"]
    m = eval_mat.shape[0]
    iso = isinstance(k, SeIsoKernel)
    kind = hyper[0]
    if kind == "log_sf2":
        het = None if iso else k.hetero_skedasticity
        if het is None:
            return ("factor", 1.0)
        res = eval_mat.copy(order="F")  # lib/cov_se_fat.ml:423-428
        res[np.diag_indices(m)] -= het
        return ("dense", res)
    if kind == "log_hetero":  # `Diag_vec, lib/cov_se_fat.ml:430-440
        if k.hetero_skedasticity is None:
            raise RuntimeError("Cov_se_fat.Deriv.Inducing.calc_deriv_upper: heteroskedastic modeling "
                               "disabled, cannot calculate derivative")
        deriv = np.zeros(m)
        deriv[hyper[1] - 1] = k.hetero_skedasticity[hyper[1] - 1]
        return ("diag_vec", deriv)
    if kind == "log_ell":
        res = eval_mat * shared_upper["sqr_diff_mat"] * k.inv_ell2
        res[np.diag_indices(m)] = 0.0
        return ("dense", _F(res))
    if kind == "proj":
        return ("const", 0.0)
    ms = None if iso else k.multiscales
    if kind == "log_multiscale":  # lib/cov_se_fat.ml:441-485
        _, ind, dim = hyper
        if ms is None:
            raise RuntimeError("Cov_se_fat.Deriv.Inducing.calc_deriv_upper: multiscale modeling disabled, "
                               "cannot calculate derivative")
        res = np.zeros(m)
        zc = inducing[dim - 1, ind - 1]
        multiscale = ms[dim - 1, ind - 1]
        multiscale_const = multiscale - 1.0
        multiscale_h = 0.5 - multiscale
        multiscale_factor = 0.5 * multiscale_h
        for i in range(1, m + 1):
            if i == ind:
                dval = eval_mat[ind - 1, ind - 1]
                if k.hetero_skedasticity is not None:
                    dval = dval - k.hetero_skedasticity[ind - 1]
                res[i - 1] = multiscale_h / (multiscale + multiscale_const) * dval
                continue
            kel = eval_mat[i - 1, ind - 1] if i < ind else eval_mat[ind - 1, i - 1]
            diff = inducing[dim - 1, i - 1] - zc
            iscale = 1.0 / (ms[dim - 1, i - 1] + multiscale_const)
            sdiff = diff * iscale
            res[i - 1] = (iscale - sdiff * sdiff) * multiscale_factor * kel
        return ("sparse_rows", res, ind)
    if kind == "inducing":
        _, ind, dim = hyper
        scale = k.inv_ell2 if iso else 1.0
        res = np.zeros(m)
        zc = inducing[dim - 1, ind - 1]
        for i in range(1, m + 1):
            if i == ind:
                continue
            kel = eval_mat[i - 1, ind - 1] if i < ind else eval_mat[ind - 1, i - 1]
            if iso:
                res[i - 1] = scale * (inducing[dim - 1, i - 1] - zc) * kel
            elif ms is None:
                res[i - 1] = (inducing[dim - 1, i - 1] - zc) * kel
            else:  # lib/cov_se_fat.ml:501-513
                sc = ms[dim - 1, i - 1] + (ms[dim - 1, ind - 1] - 1.0)
                res[i - 1] = (inducing[dim - 1, i - 1] - zc) / sc * kel
        return ("sparse_rows", res, ind)
    raise ValueError(hyper)


def spec_calc_deriv_diag(k, hyper):
    """lib/cov_se_iso.ml:297-299 / lib/cov_se_fat.ml:527-531."""
    return ("factor", 1.0) if hyper[0] == "log_sf2" else ("const", 0.0)


def spec_calc_deriv_cross(shared_cross, hyper):
    """lib/cov_se_iso.ml:301-327 / lib/cov_se_fat.ml:563-641 (no multiscale)."""
    k = shared_cross["kernel"]
    eval_mat = shared_cross["eval_mat"]
    inducing = shared_cross["inducing"]
    iso = isinstance(k, SeIsoKernel)
    kind = hyper[0]
    if kind == "log_sf2":
        return ("factor", 1.0)
    if kind == "log_hetero":  # lib/cov_se_fat.ml:597
        return ("const", 0.0)
    ms = None if iso else k.multiscales
    if kind == "log_multiscale":  # lib/cov_se_fat.ml:598-622
        _, ind, dim = hyper
        if ms is None:
            raise RuntimeError("Cov_se_fat.Deriv.Inputs.calc_deriv_cross: multiscale modeling disabled, "
                               "cannot calculate derivative")
        multiscale = ms[dim - 1, ind - 1]
        multiscale_factor = 0.5 * (0.5 - multiscale)
        diff = shared_cross["projections"][dim - 1, :] - inducing[dim - 1, ind - 1]
        iscale = 1.0 / multiscale
        sdiff = diff * iscale
        return ("sparse_cols", (iscale - sdiff * sdiff) * multiscale_factor * eval_mat[:, ind - 1], ind)
    if kind == "log_ell":
        return ("dense", _F(eval_mat * shared_cross["sqr_diff_mat"] * k.inv_ell2))
    if kind == "inducing":
        _, ind, dim = hyper
        zc = inducing[dim - 1, ind - 1]
        if iso:
            col = k.inv_ell2 * (shared_cross["inputs"][dim - 1, :] - zc) * eval_mat[:, ind - 1]
        elif ms is None:
            col = (shared_cross["projections"][dim - 1, :] - zc) * eval_mat[:, ind - 1]
        else:  # lib/cov_se_fat.ml:633-638
            col = (1.0 / ms[dim - 1, ind - 1]) * (shared_cross["projections"][dim - 1, :] - zc) * eval_mat[:, ind - 1]
        return ("sparse_cols", col, ind)
    if kind == "proj":
        _, big, small = hyper
        if k.tproj is None:
            raise RuntimeError("Cov_se_fat.Deriv.Inputs.calc_deriv_cross: tproj disabled, "
                               "cannot calculate derivative")
        alpha = shared_cross["inputs"][big - 1, :][:, None]
        proj = shared_cross["projections"][small - 1, :][:, None]
        ind_el = inducing[small - 1, :][None, :]
        if ms is None:
            return ("dense", _F(alpha * (ind_el - proj) * eval_mat))
        return ("dense", _F(alpha * ((ind_el - proj) / ms[small - 1, :][None, :]) * eval_mat))  # :585-595
    raise ValueError(hyper)


# ---------------------------------------------------------------------------
# Fitc_gp engine  (lib/fitc_gp.ml)
# ---------------------------------------------------------------------------
def log_det(chol):
    """lib/utils.ml:95-101: 2*sum(log diag), summed from i=n down to 1."""
    acc = 0.0
    for v in np.diag(chol)[::-1]:
        acc += math.log(v)
    return acc + acc


def potrf_upper(a):
    c, info = lapack.dpotrf(a, lower=0, clean=0, overwrite_a=0)
    if info != 0:
        raise NotPositiveDefinite("potrf: leading minor of order %d is not positive definite" % info)
    return _F(c)


def ichol(chol):
    """lib/utils.ml:110-113: upper triangle of (U^T U)^-1 via dpotri."""
    inv, info = lapack.dpotri(np.triu(chol), lower=0, overwrite_c=0)
    if info != 0:
        raise NotPositiveDefinite("potri info=%d" % info)
    return _F(inv)


def inducing_calc_internal(k, points, km):
    """lib/fitc_gp.ml:53-57."""
    chol_km = np.triu(km).copy(order="F")  # lacpy ~uplo:`U
    chol_km[np.diag_indices(chol_km.shape[0])] += CHOLESKY_JITTER
    chol_km = potrf_upper(chol_km)
    return dict(kernel=k, points=_F(points), km=km, chol_km=chol_km, log_det_km=log_det(chol_km))


def model_calc_internal(inducing, knm, sigma2, kn_diag, v_mat, r_vec):
    """Common_model.calc_internal lib/fitc_gp.ml:151-220."""
    if sigma2 < 0.0:
        raise ValueError("Model.check_sigma2: sigma2 < 0")
    n, m = v_mat.shape
    s_vec = r_vec + sigma2
    is_vec = 1.0 / s_vec
    log_det_s_vec = 0.0
    for i in range(n - 1, -1, -1):  # loop from n down to 1
        log_det_s_vec += math.log(s_vec[i])
    sqrt_is_vec = np.sqrt(is_vec)
    q_mat = np.zeros((n + m, m), order="F")
    q_mat[:n, :] = knm * sqrt_is_vec[:, None]  # lacpy + scal_rows
    q_mat[n:, :] = np.triu(inducing["chol_km"])
    lwork = int(lapack.dgeqrf_lwork(n + m, m)[0])  # blocked Householder QR (workspace query)
    qr, tau, _, info = lapack.dgeqrf(q_mat, lwork=lwork)
    assert info == 0
    r_mat = np.triu(qr[:m, :m]).copy(order="F")
    q_full, work, info = lapack.dorgqr(qr, tau, lwork=-1)  # workspace query
    q_full, _, info = lapack.dorgqr(qr, tau, lwork=int(work[0]))
    assert info == 0
    q_full = _F(q_full)
    log_det_r = 0.0
    for r in range(m - 1, -1, -1):  # lib/fitc_gp.ml:183-203 incl. sign fix
        el = r_mat[r, r]
        if not el > 0.0:
            r_mat[r, r:] = -r_mat[r, r:]
            q_full[:n, r] = -q_full[:n, r]
            el = -el
        log_det_r += math.log(el)
    log_det_r += log_det_r
    l1 = -0.5 * (log_det_r - inducing["log_det_km"] + log_det_s_vec + float(n) * LOG_2PI)
    return dict(inducing=inducing, knm=knm, sigma2=sigma2, kn_diag=kn_diag, v_mat=v_mat,
                r_vec=r_vec, is_vec=is_vec, sqrt_is_vec=sqrt_is_vec, q_mat=q_full,
                r_mat=r_mat, l1=l1, n=n, m=m)


def model_calc_with_kn_diag(inducing, knm, sigma2, kn_diag, variational=False):
    """lib/fitc_gp.ml:222-229 (+ Variational_model.from_common :262-263)."""
    v_mat = _F(blas.dtrsm(1.0, inducing["chol_km"], knm, side=1, lower=0, trans_a=0))
    r_vec = kn_diag - np.einsum("ij,ij->i", v_mat, v_mat)  # Mat.syrk_diag ~alpha:-1 ~beta:1
    model = model_calc_internal(inducing, knm, sigma2, kn_diag, v_mat, r_vec)
    model["model_kind"] = "variational" if variational else "standard"
    if variational:
        model["l1"] = model["l1"] + (-0.5 * float(np.dot(model["is_vec"], r_vec)))
    return model


def trained_prepare_internal(model, y):
    """lib/fitc_gp.ml:279-286."""
    n = model["n"]
    if y.shape[0] != n:
        raise ValueError("Trained.calc: Vec.dim targets (%d) <> n (%d)" % (y.shape[0], n))
    y_ = y * model["sqrt_is_vec"]
    qt_y_ = model["q_mat"][:n, :].T @ y_
    return y_, qt_y_


def trained_calc_eval(model, y):
    """Eval Trained.calc lib/fitc_gp.ml:288-292."""
    y_, qt_y_ = trained_prepare_internal(model, y)
    l2 = -0.5 * (float(y_ @ y_) - float(qt_y_ @ qt_y_))
    coeffs = blas.dtrsv(model["r_mat"], qt_y_, lower=0)
    return dict(model=model, y=y, coeffs=coeffs, l=model["l1"] + l2, l2=l2)


def cm_calc(model):
    """Deriv Common_model.calc_common/calc_internal lib/fitc_gp.ml:1037-1078."""
    n, m = model["n"], model["m"]
    inv_km = ichol(model["inducing"]["chol_km"])
    t_mat = np.triu(inv_km) - np.triu(ichol(model["r_mat"]))
    q_n = model["q_mat"][:n, :]
    q_diag = np.einsum("ij,ij->i", q_n, q_n)
    return dict(eval_model=model, inv_km=inv_km, t_mat=_F(t_mat), q_diag=q_diag)


def cm_calc_v1_vec(cm):
    """lib/fitc_gp.ml:1092-1108."""
    model = cm["eval_model"]
    is_vec = model["is_vec"]
    if model["model_kind"] == "standard":
        return is_vec * (1.0 - cm["q_diag"])
    return is_vec * (2.0 - is_vec * model["r_vec"] - cm["q_diag"])


def common_calc_log_evidence_sigma2(cm, v_vec):
    """lib/fitc_gp.ml:1112-1119."""
    s = float(np.sum(v_vec))
    if cm["eval_model"]["model_kind"] == "variational":
        s -= float(np.sum(cm["eval_model"]["is_vec"]))
    return -0.5 * s


def calc_us_mat(model):
    """Shared.calc_us_mat lib/fitc_gp.ml:931-939."""
    n = model["n"]
    u_mat = _F(blas.dtrsm(1.0, model["inducing"]["chol_km"], model["v_mat"], side=1, lower=0, trans_a=1))
    s_mat = _F(blas.dtrsm(1.0, model["r_mat"], _F(model["q_mat"][:n, :]), side=1, lower=0, trans_a=1))
    s_mat *= model["sqrt_is_vec"][:, None]
    return u_mat, s_mat


def deriv_trained_calc(cm, y):
    """Deriv Trained.calc lib/fitc_gp.ml:1158-1181."""
    model = cm["eval_model"]
    n = model["n"]
    y_, qt_y_ = trained_prepare_internal(model, y)
    u_vec = y_ - model["q_mat"][:n, :] @ qt_y_
    l2 = -0.5 * float(u_vec @ y_)
    coeffs = blas.dtrsv(model["r_mat"], qt_y_, lower=0)
    w_vec = u_vec * model["sqrt_is_vec"]
    v_vec = cm_calc_v1_vec(cm) - w_vec * w_vec
    return dict(common_model=cm, w_vec=w_vec, v_vec=v_vec, coeffs=coeffs, l2=l2,
                l=model["l1"] + l2, y=y)


def _upper_to_full(a):
    return np.triu(a) + np.triu(a, 1).T


def model_prepare_hyper(cm):
    """Cm.prepare_hyper lib/fitc_gp.ml:1126-1136 (model log-evidence only)."""
    model = cm["eval_model"]
    v_vec = cm_calc_v1_vec(cm)
    u_mat, x_mat = calc_us_mat(model)
    u1 = u_mat * np.sqrt(v_vec)[:, None]
    w_mat = np.triu(cm["t_mat"]) - np.triu(u1.T @ u1)
    x_mat = x_mat - u_mat * v_vec[:, None]
    return dict(cm=cm, v_vec=v_vec, w_mat=_F(w_mat), x_mat=_F(x_mat))


def trained_prepare_hyper(tr):
    """Trained.prepare_hyper lib/fitc_gp.ml:1192-1207."""
    cm = tr["common_model"]
    model = cm["eval_model"]
    u_mat, x_mat = calc_us_mat(model)
    t_vec = tr["coeffs"]
    w_mat = np.triu(cm["t_mat"]) - np.triu(np.outer(t_vec, t_vec))  # syr ~alpha:-1
    u1 = u_mat * np.sqrt(cm_calc_v1_vec(cm))[:, None]
    w_mat = w_mat - np.triu(u1.T @ u1)  # syrk ~trans:`T ~alpha:-1
    u2 = u_mat * tr["w_vec"][:, None]
    w_mat = w_mat + np.triu(u2.T @ u2)
    x_mat = x_mat - u_mat * tr["v_vec"][:, None] - np.outer(tr["w_vec"], t_vec)  # axpy, ger
    return dict(cm=cm, v_vec=tr["v_vec"], w_mat=_F(w_mat), x_mat=_F(x_mat))


def symm2_trace(a, b):
    """Mat.symm2_trace: tr(a b) for symmetric a, b referenced by their upper triangles."""
    ua, ub = np.triu(a, 1), np.triu(b, 1)
    return float(np.sum(np.diag(a) * np.diag(b)) + 2.0 * np.sum(ua * ub))


def sum_symm_mat(a):
    """lib/utils.ml:81-92."""
    rest = float(np.sum(np.triu(a, 1)))
    return rest + float(np.sum(np.diag(a))) + rest


def symm2_sparse_trace_single(w_mat, srow, ind):
    """lib/utils.ml:196-220 specialised to one sparse row (rows = [ind]):
    full = sum_{r != ind} W[min,max] * s_r ; half = W[ind,ind]*s_ind ; result 2*full+half."""
    m = w_mat.shape[0]
    c = ind
    full = 0.0
    half = 0.0
    for r in range(1, m + 1):
        mat_el = w_mat[c - 1, r - 1] if r > c else w_mat[r - 1, c - 1]
        if r == c:      # rows_ix = 1, rows_el = c: neither r < c nor c < c -> half branch
            half += mat_el * srow[c - 1]
        else:           # r < c: full; r > c: rows_ix was incremented past m -> full
            full += mat_el * srow[r - 1]
    return full + half + full


def shared_calc_log_evidence(hyper_t, shared, hyper):
    """Shared.calc_log_evidence lib/fitc_gp.ml:1005-1021 with the term functions
    calc_dkn_diag_term :943-954, calc_dkm_term :956-973, calc_dknm_term :975-1003."""
    v_vec, w_mat, x_mat = hyper_t["v_vec"], hyper_t["w_mat"], hyper_t["x_mat"]
    k = shared["kernel"]
    dd = spec_calc_deriv_diag(k, hyper)
    if dd[0] == "factor":
        dkn_diag_term = 0.0 if dd[1] == 0.0 else dd[1] * float(shared["kn_diag"] @ v_vec)
    else:
        dkn_diag_term = 0.0 if dd[1] == 0.0 else dd[1] * float(np.sum(v_vec))
    dk = spec_calc_deriv_upper(shared["shared_upper"], hyper)
    if dk[0] == "dense":
        dkm_term = symm2_trace(w_mat, dk[1])
    elif dk[0] == "sparse_rows":
        dkm_term = symm2_sparse_trace_single(w_mat, dk[1], dk[2])
    elif dk[0] == "diag_vec":  # lib/fitc_gp.ml:962-967
        dkm_term = float(np.sum(dk[1] * np.diag(w_mat)))
    elif dk[0] == "const":
        dkm_term = 0.0 if dk[1] == 0.0 else dk[1] * sum_symm_mat(w_mat)
    else:  # factor
        dkm_term = 0.0 if dk[1] == 0.0 else dk[1] * symm2_trace(w_mat, shared["km"])
    dx = spec_calc_deriv_cross(shared["shared_cross"], hyper)
    if dx[0] == "dense":
        dknm_term = float(np.sum(x_mat * dx[1]))
    elif dx[0] == "sparse_cols":
        dknm_term = float(x_mat[:, dx[2] - 1] @ dx[1])
    elif dx[0] == "const":
        dknm_term = 0.0 if dx[1] == 0.0 else dx[1] * float(np.sum(x_mat))
    else:
        dknm_term = 0.0 if dx[1] == 0.0 else dx[1] * float(np.sum(x_mat * shared["knm"]))
    return (-0.5 * (dkn_diag_term - dkm_term)) - dknm_term


# ---------------------------------------------------------------------------
# Prediction
# ---------------------------------------------------------------------------
def predict_means(k, inducing_points, coeffs, test_inputs):
    """Means.calc lib/fitc_gp.ml:418-425: gemv knm coeffs with knm = calc_cross at the test points."""
    ktm, _ = spec_calc_shared_cross(k, test_inputs, inducing_points)
    return ktm @ coeffs


def predict_variances(k, inducing_points, model, test_inputs, predictive=True):
    """Variances.calc lib/fitc_gp.ml:498-518 (+ get ?predictive :520-529): two dtrsm `R and two
    Mat.syrk_diag on the test cross-covariance."""
    ktm, _ = spec_calc_shared_cross(k, test_inputs, inducing_points)
    y = spec_calc_diag(k, ktm.shape[0])
    tmp = _F(blas.dtrsm(1.0, model["inducing"]["chol_km"], ktm, side=1, lower=0, trans_a=0))
    y = y - np.einsum("ij,ij->i", tmp, tmp)
    tmp = _F(blas.dtrsm(1.0, model["r_mat"], ktm, side=1, lower=0, trans_a=0))
    var = y + np.einsum("ij,ij->i", tmp, tmp)
    return var + model["sigma2"] if predictive else var


def spec_inputs_calc_upper(k, inputs):
    """Spec.Eval.Inputs.calc_upper: lib/cov_se_iso.ml:124 (= Inducing.calc_upper on the inputs) and
    lib/cov_se_fat.ml:221 (calc_upper_vanilla of the projected inputs: no multiscales, no hetero term).
    Strict lower triangle NaN, as the reference leaves it undefined."""
    inputs = _F(inputs)
    if isinstance(k, SeIsoKernel):
        return se_iso_calc_upper_with_sqr_diff(k, se_iso_calc_sqr_diff_upper(inputs))
    return se_fat_calc_upper_vanilla(k, se_fat_project(k, inputs))


def _syrk_upper(alpha, a, beta, c):
    """dsyrk `U ~trans:`N: the upper triangle of alpha*a*a^T + beta*c; lower triangle of c kept."""
    return _F(blas.dsyrk(alpha, a, beta=beta, c=c, lower=0, trans=0, overwrite_c=0))


def fitc_covariances(k, inducing_points, model, test_inputs):
    """FITC_covariances.calc lib/fitc_gp.ml:585-599: calc_upper - syrk(K_tm chol_km^-1) + syrk(K_tm r_mat^-1).
    Upper triangle defined."""
    ktm, _ = spec_calc_shared_cross(k, test_inputs, inducing_points)
    cov = spec_inputs_calc_upper(k, test_inputs)
    cov = np.triu(np.nan_to_num(cov, nan=0.0))  # dsyrk never reads the lower triangle; keep it finite
    tmp = _F(blas.dtrsm(1.0, model["inducing"]["chol_km"], ktm, side=1, lower=0, trans_a=0))
    cov = _syrk_upper(-1.0, tmp, 1.0, _F(cov))
    tmp = _F(blas.dtrsm(1.0, model["r_mat"], ktm, side=1, lower=0, trans_a=0))
    cov = _syrk_upper(1.0, tmp, 1.0, cov)
    return np.triu(cov)


def fic_covariances(k, inducing_points, model, test_inputs):
    """FIC_covariances.calc lib/fitc_gp.ml:617-627 -> calc_common :603-609.  Note :620: the diagonal
    correction is Mat.syrk_diag ~alpha:-1 ktm -- computed from K_tm itself."""
    ktm, _ = spec_calc_shared_cross(k, test_inputs, inducing_points)
    kt_diag = spec_calc_diag(k, ktm.shape[0])
    r_vec = kt_diag - np.einsum("ij,ij->i", ktm, ktm)
    q_mat = _F(blas.dtrsm(1.0, model["r_mat"], ktm, side=1, lower=0, trans_a=0))
    nt = ktm.shape[0]
    cov = _syrk_upper(1.0, q_mat, 0.0, np.zeros((nt, nt), order="F"))
    cov[np.diag_indices(nt)] += r_vec
    return np.triu(cov)


def variances_model_inputs(model):
    """Variances.calc_model_inputs lib/fitc_gp.ml:487-496: r_vec + rowsum((K_nm r_mat^-1)^2)."""
    tmp = _F(blas.dtrsm(1.0, model["r_mat"], model["knm"], side=1, lower=0, trans_a=0))
    return model["r_vec"] + np.einsum("ij,ij->i", tmp, tmp)


def fitc_covariances_model_inputs(k, model, inputs):
    """FITC_covariances.calc_model_inputs lib/fitc_gp.ml:569-579: calc_upper - syrk v_mat + syrk ~n q_mat, where
    q_mat is the model's Q factor, i.e. its first n rows are diag(sqrt is) K_nm R^-1 (:176-182) -- as written."""
    n = model["n"]
    cov = np.triu(np.nan_to_num(spec_inputs_calc_upper(k, inputs), nan=0.0))
    cov = _syrk_upper(-1.0, model["v_mat"], 1.0, _F(cov))
    cov = _syrk_upper(1.0, _F(model["q_mat"][:n, :]), 1.0, cov)
    return np.triu(cov)


def fic_covariances_model_inputs(model):
    """FIC_covariances.calc_model_inputs lib/fitc_gp.ml:609-614 -> calc_common :601-607 with the model's q_mat, r_vec."""
    n = model["n"]
    cov = _syrk_upper(1.0, _F(model["q_mat"][:n, :]), 0.0, np.zeros((n, n), order="F"))
    cov[np.diag_indices(n)] += model["r_vec"]
    return np.triu(cov)


def covariances_get(cov, sigma2, predictive=True):
    """Common_covariances.get_common lib/fitc_gp.ml:549-559."""
    if not predictive:
        return cov
    res = cov.copy()
    res[np.diag_indices(res.shape[0])] += sigma2
    return res


def cov_sampler_calc(means, cov, sigma2, predictive=True, jitter=CHOLESKY_JITTER):
    """Common_cov_sampler.calc lib/fitc_gp.ml:659-675: potrf of cov (+sigma2) + jitter."""
    a = np.triu(cov).copy()
    n = a.shape[0]
    if predictive:
        a[np.diag_indices(n)] += sigma2
    a[np.diag_indices(n)] += jitter
    return dict(means=np.asarray(means, dtype=np.float64), cov_chol=potrf_upper(_F(a)))


def cov_sampler_samples(sampler, z):
    """Common_cov_sampler.samples lib/fitc_gp.ml:685-697 with the standard normal draws `z` (n_means x n)
    given instead of drawn from GSL: trmm ~transa:`T cov_chol z, then + means per column."""
    z = _F(z)
    res = _F(blas.dtrmm(1.0, sampler["cov_chol"], z, side=0, lower=0, trans_a=1, diag=0))
    return res + sampler["means"][:, None]


def stats_calc(y, means, l):
    """Stats.calc lib/fitc_gp.ml:353-373 given the training means (Trained.calc_means :296-297)."""
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    target_variance = float(y @ y) / n                      # :318 (Vec.sqr_nrm2 y / n -- no centring)
    sse = float(np.sum((y - means) ** 2))
    mse = sse / n
    prior_l = -0.5 * math.log(2.0 * math.pi * target_variance) - 0.5   # :329-330
    ad = np.abs(y - means)
    return dict(n_samples=n, target_variance=target_variance, sse=sse, mse=mse, rmse=math.sqrt(mse),
                smse=mse / target_variance, msll=prior_l - l / n, mad=float(np.sum(ad)) / n,
                maxad=float(np.max(ad)))


# ---------------------------------------------------------------------------
# One full evaluation, the reference way (multim_dcommon lib/fitc_gp.ml:1612-1636)
# ---------------------------------------------------------------------------
def evaluate(k, inducing_points, inputs, targets, sigma2, variational=False,
             want_grad=True, hypers=None, keep=False):
    """Returns dict(l1, l2, l, coeffs, dl_dsigma2, grad (reference hyper order),
    model_dl_dsigma2, model_grad [l1 only]).  `hypers` restricts the gradient
    (the optimiser's ?hypers, lib/fitc_gp.ml:1535); default = Hyper.get_all."""
    inducing_points = _F(inducing_points)
    inputs = _F(inputs)
    targets = np.asarray(targets, dtype=np.float64)
    n = inputs.shape[1]
    km, shared_upper = spec_calc_shared_upper(k, inducing_points)       # Deriv.Inducing.calc :881-888
    inducing = inducing_calc_internal(k, inducing_points, km)
    knm, shared_cross = spec_calc_shared_cross(k, inputs, inducing_points)  # Deriv.Inputs.calc :902-911
    kn_diag = spec_calc_diag(k, n)
    model = model_calc_with_kn_diag(inducing, knm, sigma2, kn_diag, variational)
    out = dict(l1=model["l1"])
    if not want_grad:
        tr = trained_calc_eval(model, targets)
        out.update(l2=tr["l2"], l=tr["l"], coeffs=tr["coeffs"])
        if keep:
            out["model"] = model
        return out
    cm = cm_calc(model)
    tr = deriv_trained_calc(cm, targets)
    out.update(l2=tr["l2"], l=tr["l"], coeffs=tr["coeffs"])
    out["dl_dsigma2"] = common_calc_log_evidence_sigma2(cm, tr["v_vec"])
    out["model_dl_dsigma2"] = common_calc_log_evidence_sigma2(cm, cm_calc_v1_vec(cm))
    shared = dict(kernel=k, km=km, knm=knm, kn_diag=kn_diag, shared_upper=shared_upper,
                  shared_cross=shared_cross)
    if hypers is None:
        hypers = spec_hypers(k, inducing_points.shape[0], inducing_points.shape[1])
    ht = trained_prepare_hyper(tr)
    out["grad"] = np.array([shared_calc_log_evidence(ht, shared, h) for h in hypers])
    hm = model_prepare_hyper(cm)
    out["model_grad"] = np.array([shared_calc_log_evidence(hm, shared, h) for h in hypers])
    out["hypers"] = hypers
    if keep:
        out.update(model=model, cm=cm, trained=tr, hyper_t=ht, shared=shared)
    return out


# ---------------------------------------------------------------------------
# Vectorised gradient (same math, all hypers at once) -- used where looping over
# 2+d*m hypers in Python is too slow (cpu_baseline leg, mid-size parity tests).
# Validated against `evaluate` in tests/test_oracle.py.
# ---------------------------------------------------------------------------
def evaluate_fast(k, inducing_points, inputs, targets, sigma2, variational=False):
    inducing_points = _F(inducing_points)
    inputs = _F(inputs)
    targets = np.asarray(targets, dtype=np.float64)
    n = inputs.shape[1]
    d, m = inducing_points.shape
    km, shared_upper = spec_calc_shared_upper(k, inducing_points)
    inducing = inducing_calc_internal(k, inducing_points, km)
    knm, shared_cross = spec_calc_shared_cross(k, inputs, inducing_points)
    kn_diag = spec_calc_diag(k, n)
    model = model_calc_with_kn_diag(inducing, knm, sigma2, kn_diag, variational)
    cm = cm_calc(model)
    tr = deriv_trained_calc(cm, targets)
    ht = trained_prepare_hyper(tr)
    v_vec, w_mat, x_mat = ht["v_vec"], ht["w_mat"], ht["x_mat"]
    out = dict(l1=model["l1"], l2=tr["l2"], l=tr["l"], coeffs=tr["coeffs"],
               dl_dsigma2=common_calc_log_evidence_sigma2(cm, tr["v_vec"]))
    iso = isinstance(k, SeIsoKernel)
    scale = k.inv_ell2 if iso else 1.0
    wfull = _upper_to_full(w_mat)
    het = None if iso else k.hetero_skedasticity
    km_nohet = km if het is None else km - np.diag(het)
    kmfull = _upper_to_full(km_nohet)
    e_mat = x_mat * knm                       # X .* K_nm
    g_sf2 = -0.5 * (k.sf2 * float(np.sum(v_vec)) - symm2_trace(w_mat, km_nohet)) - float(np.sum(e_mat))
    wk = wfull * kmfull
    np.fill_diagonal(wk, 0.0)
    # inducing hyper (ind=c, dim=kk): 0.5*dkm_term - dknm_term
    pts = shared_cross["inputs"] if iso else shared_cross["projections"]
    ms = None if iso else k.multiscales
    if ms is None:
        dkm = 2.0 * scale * (inducing_points @ wk - inducing_points * np.sum(wk, axis=0)[None, :])
        dknm = scale * (pts @ e_mat - inducing_points * np.sum(e_mat, axis=0)[None, :])
    else:
        dkm = np.empty((d, m))
        for kk in range(d):
            zdiff = inducing_points[kk, :][:, None] - inducing_points[kk, :][None, :]      # z_r - z_c
            sc = ms[kk, :][:, None] + ms[kk, :][None, :] - 1.0
            dkm[kk] = 2.0 * np.sum(wk * zdiff / sc, axis=0)
        dknm = (pts @ e_mat - inducing_points * np.sum(e_mat, axis=0)[None, :]) / ms
    g_ind = (0.5 * dkm - dknm).T.reshape(-1)  # ind-major, dim fastest
    if iso:
        sq_u = np.nan_to_num(shared_upper["sqr_diff_mat"], nan=0.0)
        dkm_ell = symm2_trace(w_mat, np.triu(km * sq_u * k.inv_ell2, 1))
        dknm_ell = float(np.sum(e_mat * shared_cross["sqr_diff_mat"])) * k.inv_ell2
        g_ell = 0.5 * dkm_ell - dknm_ell
        out["grad"] = np.concatenate([[g_ell, g_sf2], g_ind])
    else:
        parts = [[g_sf2], g_ind]
        if k.tproj is not None:
            if ms is None:
                ez = e_mat @ inducing_points.T                      # n x d : sum_c E_rc z_small,c
                rs = np.sum(e_mat, axis=1)[:, None] * pts.T          # n x d : rowsum(E)_r p_small,r
            else:
                ez = e_mat @ (inducing_points / ms).T                # sum_c E_rc z_small,c / ms_small,c
                rs = (e_mat @ (1.0 / ms).T) * pts.T
            g_proj = -(shared_cross["inputs"] @ (ez - rs))      # big x small
            parts.append(g_proj.reshape(-1))
        if het is not None:
            parts.append(0.5 * het * np.diag(w_mat))            # `Diag_vec: 1/2 het_i W_ii
        if ms is not None:
            g_ms = np.empty((m, d))
            kdiag = np.diag(km_nohet)
            for kk in range(d):
                zdiff = inducing_points[kk, :][:, None] - inducing_points[kk, :][None, :]
                isc = 1.0 / (ms[kk, :][:, None] + ms[kk, :][None, :] - 1.0)
                inner = isc - (zdiff * isc) ** 2
                np.fill_diagonal(inner, 0.0)
                factor = 0.5 * (0.5 - ms[kk, :])
                dkm_ms = 2.0 * factor * np.sum(wk * inner, axis=0) + \
                    np.diag(w_mat) * (0.5 - ms[kk, :]) / (2.0 * ms[kk, :] - 1.0) * kdiag
                pd = pts[kk, :][:, None] - inducing_points[kk, :][None, :]
                dknm_ms = factor * np.sum(e_mat * (1.0 / ms[kk, :][None, :] - (pd / ms[kk, :][None, :]) ** 2), axis=0)
                g_ms[:, kk] = 0.5 * dkm_ms - dknm_ms
            parts.append(g_ms.reshape(-1))
        out["grad"] = np.concatenate(parts)
    return out


# ---------------------------------------------------------------------------
# Independent known-answer check: dense textbook FITC log marginal likelihood
# (doc/manual/gpr_manual.tex:684-701; Snelson & Ghahramani 2006), no QR, no
# inducing-space tricks.  O(n^3): small n only.
# ---------------------------------------------------------------------------
def dense_fitc_log_evidence(k, inducing_points, inputs, targets, sigma2, variational=False):
    inducing_points = _F(inducing_points)
    inputs = _F(inputs)
    n = inputs.shape[1]
    km, _ = spec_calc_shared_upper(k, inducing_points)
    km = _upper_to_full(np.nan_to_num(km, nan=0.0)) + CHOLESKY_JITTER * np.eye(km.shape[0])
    knm, _ = spec_calc_shared_cross(k, inputs, inducing_points)
    qnn = knm @ np.linalg.solve(km, knm.T)  # km already carries the heteroskedastic diagonal
    r = k.sf2 - np.diag(qnn)
    cov = qnn + np.diag(r + sigma2)
    sign, logdet = np.linalg.slogdet(cov)
    assert sign > 0
    alpha = np.linalg.solve(cov, targets)
    l = -0.5 * (logdet + float(targets @ alpha) + n * LOG_2PI)
    if variational:
        l += -0.5 * float(np.sum(r / (r + sigma2)))
    return l
