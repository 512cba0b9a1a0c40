"""ctypes binding of libgprhip.so (the C ABI declared in include/gprhip.h).

The HIP library is the product: there is no CPU fallback.  Importing this module on a machine
where the shared object is missing raises, and every call that needs a device raises
`GprHipError` when none is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPRHIP_LIBRARY=lab selects the lab build (`make -C gpr_amd/csrc lab`: the production code plus the measured-slower variants
# kept for A/B runs -- GPRHIP_POTRF_CHAIN, GPRHIP_POTRF_LOOKAHEAD, GPRHIP_ROUND_SYNC_US, GPRHIP_COV_OVERLAP; the production
# library ignores those switches).  Read once, when the library is first loaded.
LIB_PATH = os.path.join(_HERE, "libgprhip_lab.so" if os.environ.get("GPRHIP_LIBRARY") == "lab" else "libgprhip.so")

OK, EBADARG, ENOTPOSDEF, EHIP, EOOM, ESTATE, ECOMM, EPRECISION = range(8)
COMM_NONE, COMM_RCCL, COMM_SAME_DEVICE = 0, 1, 2
COV_SE_ISO, COV_SE_FAT = 0, 1
F64, F32_BULK = 0, 1


class GprHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class UntrustworthyCoefficients(GprHipError):
    """fp32-bulk mean coefficients refused (GPRHIP_EPRECISION): K_m is too ill-conditioned for them."""


class NotPositiveDefinite(GprHipError):
    """Counterpart of Lacaml's Failure on potrf info > 0."""


class Hypers(C.Structure):
    _fields_ = [
        ("log_ell", C.c_double),
        ("log_sf2", C.c_double),
        ("sigma2", C.c_double),
        ("inducing", C.POINTER(C.c_double)),
        ("tproj", C.POINTER(C.c_double)),
        ("variational", C.c_int),
        ("model_only", C.c_int),
        ("jitter", C.c_double),
        ("log_hetero_skedasticity", C.POINTER(C.c_double)),
        ("log_multiscales_m05", C.POINTER(C.c_double)),
        ("reuse_v", C.c_int),
    ]


class MemoryPlan(C.Structure):
    """gprhip_memory_plan_t: bytes one shard holds on its device (include/gprhip.h)."""
    _fields_ = [(k, C.c_int64) for k in ("total", "v_store", "chunk_buffers", "slices", "inputs", "row_vectors", "mxm", "rest",
                                         "k_store_optional", "chunk_rows", "kslices")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Result(C.Structure):
    _fields_ = [
        ("l1", C.c_double),
        ("l2", C.c_double),
        ("l", C.c_double),
        ("dl_dsigma2", C.c_double),
        ("n_hypers", C.c_int64),
    ]


# name -> (restype, argtypes); the single source of truth for tests/test_abi.py as well
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
SIGNATURES = {
    "gprhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gprhip_memory_plan": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(MemoryPlan)]),
    "gprhip_problem_create": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                        C.c_int64, C.POINTER(_vp)]),
    "gprhip_problem_create_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           C.c_int64, C.POINTER(_vp)]),
    "gprhip_problem_destroy": (None, [_vp]),
    "gprhip_set_inputs": (C.c_int, [_vp, _dp, C.c_int64]),
    "gprhip_set_targets": (C.c_int, [_vp, _dp]),
    "gprhip_set_inputs_device": (C.c_int, [_vp, _vp]),
    "gprhip_set_targets_device": (C.c_int, [_vp, _vp]),
    "gprhip_n_hypers": (C.c_int64, [_vp, C.c_int]),
    "gprhip_eval": (C.c_int, [_vp, C.POINTER(Hypers), C.c_int, C.POINTER(Result), _dp, _dp]),
    "gprhip_ar1_len": (C.c_int64, [_vp]),
    "gprhip_ar2_len": (C.c_int64, [_vp]),
    "gprhip_exchange_len": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gprhip_exchange_offset": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "gprhip_eval_pass1": (C.c_int, [_vp, C.POINTER(Hypers), C.c_int, C.c_int64, _vp]),
    "gprhip_eval_pass2": (C.c_int, [_vp, _vp, _vp]),
    "gprhip_eval_finish": (C.c_int, [_vp, _vp, C.POINTER(Result), _dp, _dp]),
    "gprhip_sync": (C.c_int, [_vp]),
    "gprhip_stream": (_vp, [_vp]),
    "gprhip_set_timing": (C.c_int, [_vp, C.c_int]),
    "gprhip_predict": (C.c_int, [_vp, _dp, C.c_int64, C.c_int64, C.c_int, _dp, _dp]),
    "gprhip_train_stats": (C.c_int, [_vp, _dp, _dp]),
    "gprhip_covariances": (C.c_int, [_vp, _dp, C.c_int64, C.c_int64, C.c_int, C.c_int, _dp]),
    "gprhip_cov_samples": (C.c_int, [_vp, _dp, C.c_int64, C.c_int64, C.c_double, C.c_double, _dp, _dp,
                                     C.c_int64, _dp]),
    "gprhip_co_variance_coeffs": (C.c_int, [_vp, _dp, _dp]),
    "gprhip_load_predictor": (C.c_int, [_vp, C.POINTER(Hypers), _dp, _dp, _dp]),
    "gprhip_condition": (C.c_int, [_vp, _dp, _dp]),
    "gprhip_debug_fetch": (C.c_int, [_vp, C.c_char_p, _dp, C.c_int64]),
    "gprhip_last_timings": (C.c_int, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
    "gprhip_shard_rows": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gprhip_ctx_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "gprhip_ctx_destroy": (None, [_vp]),
    "gprhip_ctx_ndev": (C.c_int, [_vp]),
    "gprhip_ctx_comm_mode": (C.c_int, [_vp]),
    "gprhip_sharded_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64,
                                        C.POINTER(_vp)]),
    "gprhip_sharded_destroy": (None, [_vp]),
    "gprhip_sharded_shard": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gprhip_sharded_problem": (_vp, [_vp, C.c_int]),
    "gprhip_sharded_set_inputs": (C.c_int, [_vp, _dp, C.c_int64]),
    "gprhip_sharded_set_targets": (C.c_int, [_vp, _dp]),
    "gprhip_sharded_eval": (C.c_int, [_vp, C.POINTER(Hypers), C.c_int, C.POINTER(Result), _dp, _dp]),
    "gprhip_sharded_predict": (C.c_int, [_vp, _dp, C.c_int64, C.c_int64, C.c_int, _dp, _dp]),
    "gprhip_sharded_train_stats": (C.c_int, [_vp, _dp, _dp]),
    "gprhip_sharded_comm_stats": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_float)]),
    "gprhip_sharded_set_timing": (C.c_int, [_vp, C.c_int]),
    "gprhip_last_error": (C.c_char_p, []),
    "gprhip_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libgprhip.so (once) and attach the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gpr_amd: %s not found -- build it with `make -C gpr_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback"
            % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status == OK:
        return
    msg = load().gprhip_last_error().decode("utf-8", "replace")
    if status == ENOTPOSDEF:
        raise NotPositiveDefinite(status, msg)
    if status == EPRECISION:
        raise UntrustworthyCoefficients(status, msg)
    if status == EBADARG:
        raise GprHipError(status, msg)
    raise GprHipError(status, msg or ("gprhip status %d" % status))


def device_count():
    n = C.c_int(0)
    check(load().gprhip_device_count(C.byref(n)))
    return n.value


def memory_plan(cov_kind, n, D, d, m, precision=F64, chunk_rows=0):
    """gprhip_memory_plan: bytes one shard of that shape holds on its device, by part (device-free arithmetic)."""
    plan = MemoryPlan()
    check(load().gprhip_memory_plan(int(cov_kind), int(precision), int(n), int(D), int(d), int(m), int(chunk_rows), C.byref(plan)))
    return plan.as_dict()
