"""Device-resident FITC problem: thin object wrapper over the gprhip C ABI.

Holds the training inputs/targets of one shard in HBM and runs evaluations of the FITC log
evidence + gradient for changing hyper-parameters (what the reference's optimiser callback
multim_dcommon does per call, lib/fitc_gp.ml:1612-1636).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import COV_SE_FAT, COV_SE_ISO, F32_BULK, F64, Hypers, Result

CHOLESKY_JITTER = 1e-6  # Utils.cholesky_jitter, lib/utils.ml:35


@dataclass
class Evaluation:
    l1: float                 # Model.calc_log_evidence
    l2: float
    l: float                  # Trained.calc_log_evidence
    dl_dsigma2: Optional[float]
    grad: Optional[np.ndarray]    # reference Hyper.get_all order
    coeffs: np.ndarray        # Trained.calc_mean_coeffs


def _f64_ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Problem:
    def __init__(self, cov_kind, n, D, d, m, device=0, chunk_rows=0, precision=F64):
        """precision: F64 (reference parity) or F32_BULK (n x m contractions in fp32, m x m work in fp64)."""
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.cov_kind, self.n, self.D, self.d, self.m = cov_kind, int(n), int(D), int(d), int(m)
        self.device, self.precision = device, precision
        _lib.check(self._lib.gprhip_problem_create_ex(device, cov_kind, int(precision), self.n, self.D, self.d,
                                                      self.m, int(chunk_rows), C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.gprhip_problem_destroy(self._h)
            self._h = C.c_void_p()

    def is_open(self):
        return bool(self._h)

    def _handle(self):
        if not self._h:
            raise _lib.GprHipError(_lib.ESTATE, "gpr_amd.Problem: the problem has been closed")
        return self._h

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data
    def set_inputs(self, inputs):
        """inputs: D x n (reference layout, one point per column); numpy, any order."""
        x = np.asfortranarray(inputs, dtype=np.float64)
        if x.shape != (self.D, self.n):
            raise ValueError("set_inputs: expected shape (%d, %d), got %s" % (self.D, self.n, x.shape))
        _lib.check(self._lib.gprhip_set_inputs(self._handle(), _f64_ptr(x), self.D))

    def set_targets(self, targets):
        y = np.ascontiguousarray(targets, dtype=np.float64)
        if y.shape != (self.n,):
            # Trained.calc: Vec.dim targets <> n  (lib/fitc_gp.ml:283-284)
            raise ValueError("Trained.calc: Vec.dim targets (%d) <> n (%d)" % (y.shape[0], self.n))
        _lib.check(self._lib.gprhip_set_targets(self._handle(), _f64_ptr(y)))

    def set_inputs_device(self, ptr):
        """ptr: device address of a contiguous point-major [n][D] fp64 array (e.g. tensor.data_ptr())."""
        _lib.check(self._lib.gprhip_set_inputs_device(self._handle(), C.c_void_p(ptr)))

    def set_targets_device(self, ptr):
        _lib.check(self._lib.gprhip_set_targets_device(self._handle(), C.c_void_p(ptr)))

    # ---- evaluation
    def n_hypers(self, has_tproj=False, has_hetero=False, has_multiscale=False):
        flags = int(has_tproj) | (int(has_hetero) << 1) | (int(has_multiscale) << 2)
        return int(self._lib.gprhip_n_hypers(self._handle(), flags))

    def _hypers(self, log_ell, log_sf2, sigma2, inducing, tproj, variational, model_only, jitter,
                log_hetero_skedasticity=None, log_multiscales_m05=None, reuse_v=False):
        z = np.asfortranarray(inducing, dtype=np.float64)
        if z.shape != (self.d, self.m):
            raise ValueError("inducing: expected shape (%d, %d), got %s" % (self.d, self.m, z.shape))
        h = Hypers()
        h.log_ell, h.log_sf2, h.sigma2 = float(log_ell), float(log_sf2), float(sigma2)
        h.inducing = _f64_ptr(z)
        keep = [z]
        if tproj is not None:
            tp = np.asfortranarray(tproj, dtype=np.float64)
            if tp.shape != (self.D, self.d):
                raise ValueError("tproj: expected shape (%d, %d), got %s" % (self.D, self.d, tp.shape))
            h.tproj = _f64_ptr(tp)
            keep.append(tp)
        h.variational, h.model_only, h.jitter = int(variational), int(model_only), float(jitter)
        h.reuse_v = int(reuse_v)
        if log_hetero_skedasticity is not None:
            lh = np.ascontiguousarray(log_hetero_skedasticity, dtype=np.float64)
            if lh.shape != (self.m,):
                raise ValueError("log_hetero_skedasticity: expected %d entries" % self.m)
            h.log_hetero_skedasticity = _f64_ptr(lh)
            keep.append(lh)
        if log_multiscales_m05 is not None:
            lm = np.asfortranarray(log_multiscales_m05, dtype=np.float64)
            if lm.shape != (self.d, self.m):
                raise ValueError("log_multiscales_m05: expected shape (%d, %d)" % (self.d, self.m))
            h.log_multiscales_m05 = _f64_ptr(lm)
            keep.append(lm)
        return h, keep

    def eval(self, *, log_sf2, sigma2, inducing, log_ell=0.0, tproj=None, variational=False,
             model_only=False, want_grad=True, jitter=CHOLESKY_JITTER, log_hetero_skedasticity=None,
             log_multiscales_m05=None, reuse_v=False):
        """reuse_v=True: only sigma2/targets changed since the previous eval on this problem
        (Model.update_sigma2, lib/fitc_gp.ml:234-236): K_nm, V and r are reused."""
        h, keep = self._hypers(log_ell, log_sf2, sigma2, inducing, tproj, variational, model_only, jitter,
                               log_hetero_skedasticity, log_multiscales_m05, reuse_v)
        res = Result()
        nh = self.n_hypers(tproj is not None, log_hetero_skedasticity is not None, log_multiscales_m05 is not None)
        grad = np.empty(nh if want_grad else 1, dtype=np.float64)
        coeffs = np.empty(self.m, dtype=np.float64)
        _lib.check(self._lib.gprhip_eval(self._handle(), C.byref(h), int(want_grad), C.byref(res),
                                         _f64_ptr(grad), _f64_ptr(coeffs)))
        del keep
        return Evaluation(res.l1, res.l2, res.l, res.dl_dsigma2 if want_grad else None,
                          grad[:res.n_hypers] if want_grad else None, coeffs)

    # ---- prediction
    def predict(self, test_inputs, predictive=True, want_variances=True):
        """Means.calc / Variances.calc (lib/fitc_gp.ml:418-425, :498-518) at test points (D x nt), using
        the model state of the last evaluation.  Returns (means, variances or None)."""
        xt = np.asfortranarray(test_inputs, dtype=np.float64)
        if xt.ndim != 2 or xt.shape[0] != self.D:
            raise ValueError("predict: expected test inputs of shape (%d, nt)" % self.D)
        nt = xt.shape[1]
        means = np.empty(nt, dtype=np.float64)
        var = np.empty(nt, dtype=np.float64) if want_variances else None
        _lib.check(self._lib.gprhip_predict(self._handle(), _f64_ptr(xt), self.D, nt, int(predictive), _f64_ptr(means),
                                            _f64_ptr(var) if want_variances else None))
        return means, var

    def train_stats(self, want_means=False):
        """Residual sums of the training set under the last evaluation's mean coefficients
        (Trained.calc_means + the loops of Stats.calc, lib/fitc_gp.ml:296-297, :353-373).
        Returns (sums = [sse, sum|y-mean|, max|y-mean|, sum y^2], means or None)."""
        sums = np.empty(4, dtype=np.float64)
        means = np.empty(self.n, dtype=np.float64) if want_means else None
        _lib.check(self._lib.gprhip_train_stats(self._handle(), _f64_ptr(means) if want_means else None,
                                                _f64_ptr(sums)))
        return sums, means

    def covariances(self, test_inputs, kind="FITC", predictive=True):
        """FITC_covariances.calc / FIC_covariances.calc (lib/fitc_gp.ml:585-599, :617-627): nt x nt posterior
        covariance between the test points (full symmetric matrix)."""
        xt = np.asfortranarray(test_inputs, dtype=np.float64)
        if xt.ndim != 2 or xt.shape[0] != self.D:
            raise ValueError("covariances: expected test inputs of shape (%d, nt)" % self.D)
        nt = xt.shape[1]
        cov = np.empty((nt, nt), dtype=np.float64, order="F")
        _lib.check(self._lib.gprhip_covariances(self._handle(), _f64_ptr(xt), self.D, nt, {"FITC": 0, "FIC": 1}[kind],
                                                int(predictive), _f64_ptr(cov)))
        return cov

    def cov_samples(self, covariances, means, z, add_diag=0.0, jitter=CHOLESKY_JITTER):
        """Common_cov_sampler.calc + samples (lib/fitc_gp.ml:656-697): means + chol(cov + (add_diag+jitter) I)^T z
        for every column of z (nt x ns standard normal draws)."""
        cov = np.asfortranarray(covariances, dtype=np.float64)
        nt = cov.shape[0]
        means = np.ascontiguousarray(means, dtype=np.float64)
        z = np.asfortranarray(z, dtype=np.float64)
        if z.ndim == 1:
            z = np.asfortranarray(z.reshape(nt, 1))
        if cov.shape != (nt, nt) or means.shape != (nt,) or z.shape[0] != nt:
            raise ValueError("cov_samples: shapes of covariances/means/z disagree")
        ns = z.shape[1]
        out = np.empty((nt, ns), dtype=np.float64, order="F")
        _lib.check(self._lib.gprhip_cov_samples(self._handle(), _f64_ptr(cov), nt, nt, float(add_diag), float(jitter),
                                                _f64_ptr(means), _f64_ptr(z), ns, _f64_ptr(out)))
        return out

    # ---- model export / import (bin/ocaml_gpr.ml:207-232, :373-413)
    def co_variance_coeffs(self):
        """Model.calc_co_variance_coeffs (lib/fitc_gp.ml:240): (chol_km, r_mat) of the last evaluation, each an
        m x m upper-triangular Fortran matrix."""
        u = np.empty((self.m, self.m), dtype=np.float64, order="F")
        r = np.empty((self.m, self.m), dtype=np.float64, order="F")
        _lib.check(self._lib.gprhip_co_variance_coeffs(self._handle(), _f64_ptr(u), _f64_ptr(r)))
        return u, r

    def load_predictor(self, *, log_sf2, sigma2, inducing, coeffs=None, co_variance_coeffs=None, log_ell=0.0,
                       tproj=None, jitter=CHOLESKY_JITTER, log_hetero_skedasticity=None,
                       log_multiscales_m05=None):
        """Mean_predictor.calc / Co_variance_predictor.calc (lib/fitc_gp.ml:386-391, :446-447): install a saved
        model's predictor state; `predict` / `covariances` then work without any evaluation."""
        h, keep = self._hypers(log_ell, log_sf2, sigma2, inducing, tproj, False, False, jitter,
                               log_hetero_skedasticity, log_multiscales_m05)
        c = u = r = None
        if coeffs is not None:
            c = np.ascontiguousarray(coeffs, dtype=np.float64)
            if c.shape != (self.m,):  # lib/fitc_gp.ml:387-390
                raise ValueError("Mean_predictor.calc: number of inducing points disagrees with dimension of "
                                 "coefficients")
        if co_variance_coeffs is not None:
            u = np.asfortranarray(co_variance_coeffs[0], dtype=np.float64)
            r = np.asfortranarray(co_variance_coeffs[1], dtype=np.float64)
            if u.shape != (self.m, self.m) or r.shape != (self.m, self.m):
                raise ValueError("Co_variance_predictor.calc: expected two %d x %d factors" % (self.m, self.m))
        _lib.check(self._lib.gprhip_load_predictor(self._handle(), C.byref(h), _f64_ptr(c) if c is not None else None,
                                                   _f64_ptr(u) if u is not None else None,
                                                   _f64_ptr(r) if r is not None else None))
        del keep

    # ---- staged evaluation (row-sharded across devices; see gpr_amd/dist.py)
    def ar1_len(self):
        return int(self._lib.gprhip_ar1_len(self._handle()))

    def ar2_len(self):
        return int(self._lib.gprhip_ar2_len(self._handle()))

    def eval_pass1(self, ar1_ptr, n_total, *, log_sf2, sigma2, inducing, log_ell=0.0, tproj=None,
                   variational=False, model_only=False, want_grad=True, jitter=CHOLESKY_JITTER,
                   log_hetero_skedasticity=None, log_multiscales_m05=None, reuse_v=False):
        h, keep = self._hypers(log_ell, log_sf2, sigma2, inducing, tproj, variational, model_only, jitter,
                               log_hetero_skedasticity, log_multiscales_m05, reuse_v)
        self._has_ms = log_multiscales_m05 is not None
        self._want_grad = bool(want_grad)
        self._has_tproj = tproj is not None
        self._has_het = log_hetero_skedasticity is not None
        _lib.check(self._lib.gprhip_eval_pass1(self._handle(), C.byref(h), int(want_grad), int(n_total),
                                               C.c_void_p(ar1_ptr)))
        del keep  # the library copies borrowed host buffers before returning

    def eval_pass2(self, ar1_ptr, ar2_ptr):
        _lib.check(self._lib.gprhip_eval_pass2(self._handle(), C.c_void_p(ar1_ptr), C.c_void_p(ar2_ptr)))

    def eval_finish(self, ar2_ptr):
        res = Result()
        nh = self.n_hypers(self._has_tproj, self._has_het, self._has_ms)
        grad = np.empty(nh if self._want_grad else 1, dtype=np.float64)
        coeffs = np.empty(self.m, dtype=np.float64)
        _lib.check(self._lib.gprhip_eval_finish(self._handle(), C.c_void_p(ar2_ptr), C.byref(res),
                                                _f64_ptr(grad), _f64_ptr(coeffs)))
        return Evaluation(res.l1, res.l2, res.l, res.dl_dsigma2 if self._want_grad else None,
                          grad[:res.n_hypers] if self._want_grad else None, coeffs)

    def sync(self):
        _lib.check(self._lib.gprhip_sync(self._handle()))

    def stream(self):
        return self._lib.gprhip_stream(self._handle())

    # ---- diagnostics
    def set_timing(self, level):
        """0: none; 1: HIP events around the dominant kernel (pass-1 SYRK); 2: around every stage."""
        _lib.check(self._lib.gprhip_set_timing(self._handle(), int(level)))

    def condition(self):
        """(cond estimate of K_m + jitter, relative error bound of the mean coefficients) of the current model state."""
        c, b = C.c_double(), C.c_double()
        _lib.check(self._lib.gprhip_condition(self._handle(), C.byref(c), C.byref(b)))
        return c.value, b.value

    def debug_fetch(self, name):
        length = self.m if name == "t" else self.n
        out = np.empty(length, dtype=np.float64)
        _lib.check(self._lib.gprhip_debug_fetch(self._handle(), name.encode(), _f64_ptr(out), length))
        return out

    def debug_fetch_matrix(self, name, rows=None):
        """"km": K_m of the last evaluation (m x m, upper triangle valid, zeros below); "knm_rows": the first `rows`
        rows of K_nm rebuilt with the last evaluation's kernel (rows x m)."""
        if name in ("km", "w_mat"):
            out = np.empty((self.m, self.m), dtype=np.float64, order="F")
        elif name in ("knm_rows", "x_rows"):
            out = np.empty((int(rows), self.m), dtype=np.float64, order="F")
        else:
            raise ValueError("debug_fetch_matrix: unknown name %r" % name)
        _lib.check(self._lib.gprhip_debug_fetch(self._handle(), name.encode(), _f64_ptr(out), out.size))
        return out

    def last_timings(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        k = self._lib.gprhip_last_timings(self._handle(), names, ms, 32)
        return {names[i].decode(): float(ms[i]) for i in range(k)}
