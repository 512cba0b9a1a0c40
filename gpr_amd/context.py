"""Single-process, multi-device evaluation: ctypes mirror of gprhip_ctx_* / gprhip_sharded_* (include/gprhip.h).

This is how the reference's one-process host (bin/ocaml_gpr.ml:176-177, :340-342) reaches the GPUs of a node: the
library shards the training points over the devices of a `Context` and does the exchange steps itself with RCCL
(loaded by the library with dlopen) on its own streams.  gpr_amd/dist.py is the other launch mode -- one process per
GPU under torch.distributed -- over the same staged calls.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import COMM_NONE, COMM_RCCL, COMM_SAME_DEVICE, F64, Result  # noqa: F401
from .problem import CHOLESKY_JITTER, Evaluation, Problem, _f64_ptr


class Context:
    """devices: the HIP devices to shard over.  All distinct -> RCCL; one device named several times -> validation
    mode (the shards share the device, the exchange is a device-local fixed-order sum)."""

    def __init__(self, devices):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        devs = [int(x) for x in devices]
        arr = (C.c_int * len(devs))(*devs)
        _lib.check(self._lib.gprhip_ctx_create(arr, len(devs), C.byref(self._h)))
        self.devices = devs

    @property
    def ndev(self):
        return int(self._lib.gprhip_ctx_ndev(self._h))

    @property
    def comm_mode(self):
        return int(self._lib.gprhip_ctx_comm_mode(self._h))

    def close(self):
        if self._h:
            self._lib.gprhip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BorrowedProblem(Problem):
    """A shard's device problem, owned by the sharded problem it belongs to."""

    def __init__(self, lib, handle, cov_kind, n, D, d, m, device, precision):
        self._lib = lib
        self._h = C.c_void_p(handle)
        self.cov_kind, self.n, self.D, self.d, self.m = cov_kind, int(n), int(D), int(d), int(m)
        self.device, self.precision = device, precision

    def close(self):
        self._h = C.c_void_p()


class ShardedDeviceProblem:
    """The whole FITC problem (n training points) row-sharded over the devices of a Context; `eval` has the
    signature and the results of gpr_amd.Problem.eval."""

    def __init__(self, ctx, cov_kind, n, D, d, m, chunk_rows=0, precision=F64):
        self._lib = _lib.load()
        self.ctx = ctx
        self._h = C.c_void_p()
        self.cov_kind, self.n, self.D, self.d, self.m = cov_kind, int(n), int(D), int(d), int(m)
        self.precision = precision
        _lib.check(self._lib.gprhip_sharded_create(ctx._h, cov_kind, int(precision), self.n, self.D, self.d, self.m,
                                                   int(chunk_rows), C.byref(self._h)))
        self._hyper_builder = Problem._hypers  # same argument checks and layouts as a single-device problem

    def close(self):
        if self._h:
            self._lib.gprhip_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, idx):
        """(device, row_lo, row_hi) of shard idx."""
        dev, lo, hi = C.c_int(), C.c_int64(), C.c_int64()
        _lib.check(self._lib.gprhip_sharded_shard(self._h, int(idx), C.byref(dev), C.byref(lo), C.byref(hi)))
        return dev.value, lo.value, hi.value

    def problem(self, idx=0):
        """Shard idx's device problem (borrowed): predict / covariances / co_variance_coeffs work on it after an
        evaluation; train_stats and per-row debug_fetch names cover that shard's rows."""
        dev, lo, hi = self.shard(idx)
        h = self._lib.gprhip_sharded_problem(self._h, int(idx))
        return _BorrowedProblem(self._lib, h, self.cov_kind, hi - lo, self.D, self.d, self.m, dev, self.precision)

    def set_inputs(self, inputs):
        x = np.asfortranarray(inputs, dtype=np.float64)
        if x.shape != (self.D, self.n):
            raise ValueError("set_inputs: expected shape (%d, %d), got %s" % (self.D, self.n, x.shape))
        _lib.check(self._lib.gprhip_sharded_set_inputs(self._h, _f64_ptr(x), self.D))

    def set_targets(self, targets):
        y = np.ascontiguousarray(targets, dtype=np.float64)
        if y.shape != (self.n,):
            raise ValueError("Trained.calc: Vec.dim targets (%d) <> n (%d)" % (y.shape[0], self.n))
        _lib.check(self._lib.gprhip_sharded_set_targets(self._h, _f64_ptr(y)))

    def n_hypers(self, has_tproj=False, has_hetero=False, has_multiscale=False):
        flags = int(has_tproj) | (int(has_hetero) << 1) | (int(has_multiscale) << 2)
        return int(self._lib.gprhip_n_hypers(self._lib.gprhip_sharded_problem(self._h, 0), flags))

    def eval(self, *, log_sf2, sigma2, inducing, log_ell=0.0, tproj=None, variational=False, model_only=False,
             want_grad=True, jitter=CHOLESKY_JITTER, log_hetero_skedasticity=None, log_multiscales_m05=None,
             reuse_v=False):
        h, keep = self._hyper_builder(self, log_ell, log_sf2, sigma2, inducing, tproj, variational, model_only, jitter,
                                      log_hetero_skedasticity, log_multiscales_m05, reuse_v)
        res = Result()
        nh = self.n_hypers(tproj is not None, log_hetero_skedasticity is not None, log_multiscales_m05 is not None)
        grad = np.empty(nh if want_grad else 1, dtype=np.float64)
        coeffs = np.empty(self.m, dtype=np.float64)
        _lib.check(self._lib.gprhip_sharded_eval(self._h, C.byref(h), int(want_grad), C.byref(res), _f64_ptr(grad),
                                                 _f64_ptr(coeffs)))
        del keep
        return Evaluation(res.l1, res.l2, res.l, res.dl_dsigma2 if want_grad else None,
                          grad[:res.n_hypers] if want_grad else None, coeffs)

    def predict(self, test_inputs, predictive=True, want_variances=True):
        """Means.calc / Variances.calc at test points (D x nt), the test points split over the devices."""
        xt = np.asfortranarray(test_inputs, dtype=np.float64)
        if xt.ndim != 2 or xt.shape[0] != self.D:
            raise ValueError("predict: expected test inputs of shape (%d, nt)" % self.D)
        nt = xt.shape[1]
        means = np.empty(nt, dtype=np.float64)
        var = np.empty(nt, dtype=np.float64) if want_variances else None
        _lib.check(self._lib.gprhip_sharded_predict(self._h, _f64_ptr(xt), self.D, nt, int(predictive), _f64_ptr(means),
                                                    _f64_ptr(var) if want_variances else None))
        return means, var

    def train_stats(self, want_means=False):
        """Residual sums of the whole training set (all shards): ([sse, sum|y-mean|, max|y-mean|, sum y^2], means or None)."""
        sums = np.empty(4, dtype=np.float64)
        means = np.empty(self.n, dtype=np.float64) if want_means else None
        _lib.check(self._lib.gprhip_sharded_train_stats(self._h, _f64_ptr(means) if want_means else None, _f64_ptr(sums)))
        return sums, means

    def set_timing(self, level):
        _lib.check(self._lib.gprhip_sharded_set_timing(self._h, int(level)))

    def comm_stats(self):
        """Exchange steps of the last evaluation: {"collectives", "bytes": [b1, b2], "ms": [t1, t2]}."""
        k = C.c_int()
        b = (C.c_int64 * 2)()
        ms = (C.c_float * 2)()
        _lib.check(self._lib.gprhip_sharded_comm_stats(self._h, C.byref(k), b, ms))
        return {"collectives": k.value, "bytes": [int(b[0]), int(b[1])], "ms": [float(ms[0]), float(ms[1])]}
