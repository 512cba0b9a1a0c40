"""TEST INFRASTRUCTURE -- ctypes wrapper of oracle/fitc_ref.c (the C restatement of the reference's Cov_se_iso FITC
evaluation over the host's LAPACK).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_build", "libfitc_ref.so")
_lib = None


def build():
    """gcc -O3 -fopenmp (portable code: the build host is not the host that runs it)."""
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(_HERE, "fitc_ref.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", src, "-o", SO, "-ldl", "-lm"])
    return SO


def lapack_path():
    """The LAPACK shared object the restatement calls into: scipy's bundled OpenBLAS."""
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "*openblas*.so*"))
    if not libs:
        raise RuntimeError("fitc_ref: no OpenBLAS found next to scipy")
    return os.path.realpath(libs[0])


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        lib.fitc_ref_iso.restype = C.c_int
        lib.fitc_ref_iso.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, dp, dp, dp, C.c_double, C.c_double,
                                     C.c_double, C.c_int, dp, dp, dp, dp]
        _lib = lib
    return _lib


def usable_cores():
    """Cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota
    (cpu.max / cfs_quota_us) -- more threads than the quota only get throttled (measured on the GPU box: 256 CPUs
    visible, quota 16: dgemm 872 GFLOP/s with 8 threads, 642 with 64; a 16384 x 2048 QR 2.5 s with 8, 7.7 s with 64)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                if f[0] != "max":
                    n = min(n, max(1, int(int(f[0]) / int(f[1]))))
            else:
                q = int(f[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


def iso_eval(X, y, Z, log_ell, log_sf2, sigma2, threads=None):
    """X: d x n, Z: d x m (Fortran), y: n.  Returns dict(l1, l2, l, dl_dsigma2, grad, coeffs, secs)."""
    lib = load()
    X = np.asfortranarray(X, dtype=np.float64)
    Z = np.asfortranarray(Z, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    d, n = X.shape
    m = Z.shape[1]
    out = np.zeros(4)
    grad = np.zeros(2 + d * m)
    coeffs = np.zeros(m)
    secs = np.zeros(8)
    dp = C.POINTER(C.c_double)
    p = lambda a: a.ctypes.data_as(dp)
    rc = lib.fitc_ref_iso(lapack_path().encode(), n, m, d, p(X), p(y), p(Z), float(log_ell), float(log_sf2),
                          float(sigma2), int(threads or usable_cores()), p(out), p(grad), p(coeffs), p(secs))
    if rc != 0:
        raise RuntimeError("fitc_ref_iso failed: %d" % rc)
    return dict(l1=out[0], l2=out[1], l=out[2], dl_dsigma2=out[3], grad=grad, coeffs=coeffs, secs=secs[:6].copy(),
                blas_threads=int(secs[6]), omp_threads=int(secs[7]))
