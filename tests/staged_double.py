"""CPU test double with the staged interface of gpr_amd.Problem (eval_pass1 / eval_pass2 /
eval_finish on caller-owned exchange buffers).

TEST INFRASTRUCTURE ONLY: lets the world_size-2 gloo tests exercise the row-sharding and
all-reduce logic of gpr_amd.dist without a GPU.  It restates in numpy the *build's* two-pass
whitened formulation (B~ = I + V^T S^-1 V by SYRK + potrf instead of the reference's stacked QR;
DESIGN.md section 3) on top of
the oracle's covariance functions, so it doubles as an independent check of that algebra against
the reference-sequence oracle.

Its exchange buffers are the LIBRARY's: lengths from gprhip_exchange_len and the symmetric m x m parts packed as upper
128 x 128 tiles at gprhip_exchange_offset (both device-free entry points of libgprhip.so) -- so the world_size-2 test
reduces buffers of exactly the size and layout the GPU path reduces, and a change of the packed layout on the device side
shows up here.
"""
import ctypes

import numpy as np
import scipy.linalg as sl

from oracle import fitc_oracle as O
from gpr_amd.problem import Evaluation


def _view(ptr, length):
    return np.ctypeslib.as_array((ctypes.c_double * length).from_address(ptr))


class StagedDouble:
    def __init__(self, kernel_factory, n, D, d, m):
        """kernel_factory(log_ell, log_sf2, tproj) -> oracle kernel object.  (Cov_se_iso layouts: D == d.)"""
        from gpr_amd import _lib
        self.kernel_factory, self.n, self.D, self.d, self.m = kernel_factory, n, D, d, m
        self._lib = _lib.load()
        self.mp = (m + 127) // 128 * 128
        # position of every entry of the upper triangle (tile-wise upper: r // 128 <= c // 128) in the packed part
        rr, cc = np.triu_indices(m)
        self._pk_r, self._pk_c = rr, cc
        self._pk_off = np.array([self._lib.gprhip_exchange_offset(m, int(r), int(c)) for r, c in zip(rr, cc)], dtype=np.int64)
        assert self._pk_off.min() >= 0
        self.packed = int(self._lib.gprhip_exchange_len(0, D, d, m, 1)) - self.mp - 4

    def ar1_len(self):
        return int(self._lib.gprhip_exchange_len(0, self.D, self.d, self.m, 1))

    def ar2_len(self):
        return int(self._lib.gprhip_exchange_len(0, self.D, self.d, self.m, 2))

    def _pack(self, buf, sym):
        """upper triangle of the symmetric m x m `sym` into the packed head of an exchange buffer"""
        buf[:self.packed] = 0.0
        buf[self._pk_off] = sym[self._pk_r, self._pk_c]

    def _unpack(self, buf):
        out = np.zeros((self.m, self.m))
        out[self._pk_r, self._pk_c] = buf[self._pk_off]
        return out + np.triu(out, 1).T

    def set_inputs(self, x):
        self.X = np.asfortranarray(x, dtype=np.float64)

    def set_targets(self, y):
        self.y = np.asarray(y, dtype=np.float64)

    def sync(self):
        pass

    def eval_pass1(self, ar1_ptr, n_total, *, log_sf2, sigma2, inducing, log_ell=0.0, tproj=None,
                   variational=False, model_only=False, want_grad=True, jitter=1e-6):
        m = self.m
        self.k = self.kernel_factory(log_ell, log_sf2, tproj)
        self.Z = np.asfortranarray(inducing)
        self.sigma2, self.variational, self.model_only = sigma2, variational, model_only
        self.want_grad, self.n_total, self.jitter = want_grad, n_total, jitter
        km, self.su = O.spec_calc_shared_upper(self.k, self.Z)
        self.km = np.triu(km) + np.triu(km, 1).T
        self.U = np.linalg.cholesky(self.km + jitter * np.eye(m)).T
        self.Ui = sl.solve_triangular(self.U, np.eye(m))
        self.K, self.sc = O.spec_calc_shared_cross(self.k, self.X, self.Z)
        self.V = self.K @ self.Ui
        V = self.V
        self.r = self.k.sf2 - np.sum(V * V, axis=1)
        s = self.r + sigma2
        self.is_ = 1.0 / s
        yy = np.zeros(self.n) if model_only else self.y
        ar1 = _view(ar1_ptr, self.ar1_len())
        ar1[:] = 0.0
        self._pack(ar1, (V * self.is_[:, None]).T @ V)
        pk, mp = self.packed, self.mp
        ar1[pk:pk + m] = V.T @ (self.is_ * yy)
        ar1[pk + mp:] = [np.sum(np.log(s)), np.sum(self.is_ * yy * yy), np.sum(self.is_ * self.r), 0.0]

    def eval_pass2(self, ar1_ptr, ar2_ptr):
        m, d = self.m, self.d
        ar1 = _view(ar1_ptr, self.ar1_len())
        pk, mp = self.packed, self.mp
        self.tail1 = ar1[pk + mp:].copy()
        Bt = np.eye(m) + self._unpack(ar1)
        c = ar1[pk:pk + m]
        self.R = np.linalg.cholesky(Bt).T
        Ri = sl.solve_triangular(self.R, np.eye(m))
        self.b = Ri.T @ c
        self.tt = Ri @ self.b
        self.t = self.Ui @ self.tt
        self.binv = Ri @ Ri.T
        ar2 = _view(ar2_ptr, self.ar2_len())
        ar2[:] = 0.0
        if not self.want_grad:
            return
        V = self.V
        Q = V @ Ri
        q = self.is_ * np.sum(Q * Q, axis=1)
        yy = np.zeros(self.n) if self.model_only else self.y
        res = 0.0 * yy if self.model_only else yy - Q @ self.b
        w = self.is_ * res
        v1 = self.is_ * (2.0 - self.is_ * self.r - q) if self.variational else self.is_ * (1.0 - q)
        v = v1 - w * w
        Xt = self.is_[:, None] * (Q @ Ri.T) - v[:, None] * V - np.outer(w, self.tt)
        Xm = Xt @ self.Ui.T
        E = Xm * self.K
        pts = self.sc["inputs"] if isinstance(self.k, O.SeIsoKernel) else self.sc["projections"]
        sq = np.zeros_like(E)
        for i in range(d):
            df = pts[i, :][:, None] - self.Z[i, :][None, :]
            sq += df * df
        self._pack(ar2, (V * v[:, None]).T @ V)
        col = np.vstack([E.sum(0)[None, :], pts @ E])
        pk, mp = self.packed, self.mp
        colv = ar2[pk:pk + (d + 1) * mp].reshape(d + 1, mp)   # [d + 1][mp]: sum E, sum p_k E
        colv[:, :m] = col
        ar2[pk + (d + 1) * mp:] = [v.sum(), self.is_.sum(), np.sum(w * res), v1.sum(), E.sum(),
                                   np.sum(E * sq), 0.0, 0.0]

    def eval_finish(self, ar2_ptr):
        m, d = self.m, self.d
        ar2 = _view(ar2_ptr, self.ar2_len())
        pk, mp = self.packed, self.mp
        G = self._unpack(ar2)
        col = ar2[pk:pk + (d + 1) * mp].reshape(d + 1, mp)[:, :m]
        tail = ar2[pk + (d + 1) * mp:]
        logdet_bt = 2 * np.sum(np.log(np.diag(self.R)))
        l1 = -0.5 * (logdet_bt + self.tail1[0] + self.n_total * O.LOG_2PI)
        if self.variational:
            l1 += -0.5 * self.tail1[2]
        l2 = 0.0 if self.model_only else -0.5 * (self.tail1[1] - self.b @ self.b)
        if not self.want_grad:
            return Evaluation(l1, l2, l1 + l2, None, None, self.t.copy())
        Wt = np.eye(m) - self.binv - np.outer(self.tt, self.tt) - G
        W = self.Ui @ Wt @ self.Ui.T
        dls2 = -0.5 * (tail[0] - tail[1] if self.variational else tail[0])
        iso = isinstance(self.k, O.SeIsoKernel)
        scale = self.k.inv_ell2 if iso else 1.0
        wk = W * self.km
        g_sf2 = -0.5 * (self.k.sf2 * tail[0] - wk.sum()) - tail[4]
        zq = np.zeros((m, m))
        for i in range(d):
            df = self.Z[i, :][:, None] - self.Z[i, :][None, :]
            zq += df * df
        gi = np.empty((m, d))
        for k_ in range(d):
            zp = np.sum(wk * (self.Z[k_, :][:, None] - self.Z[k_, :][None, :]), axis=0)
            gi[:, k_] = scale * zp - scale * (col[k_ + 1] - self.Z[k_, :] * col[0])
        if iso:
            g_ell = 0.5 * scale * np.sum(wk * zq) - scale * tail[5]
            grad = np.concatenate([[g_ell, g_sf2], gi.reshape(-1)])
        else:
            grad = np.concatenate([[g_sf2], gi.reshape(-1)])
        return Evaluation(l1, l2, l1 + l2, dls2, grad, self.t.copy())
