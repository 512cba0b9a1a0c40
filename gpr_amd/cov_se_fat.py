"""Host-side mirror of Gpr.Cov_se_fat (reference lib/cov_se_fat.ml, lib/cov_se_fat.mli): projection,
heteroskedastic noise and multiscales, each optional, with all their hyper-parameters.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import NamedTuple, Optional, Union

import numpy as np

from ._lib import COV_SE_FAT

COV_KIND = COV_SE_FAT


@dataclass(frozen=True, eq=False)
class Params:
    """Cov_se_fat.Params.params (lib/cov_se_fat.ml:27-49)."""
    d: int
    log_sf2: float
    tproj: Optional[np.ndarray] = None                 # big_dim x d
    log_hetero_skedasticity: Optional[np.ndarray] = None
    log_multiscales_m05: Optional[np.ndarray] = None

    @staticmethod
    def create(d, log_sf2, tproj=None, log_hetero_skedasticity=None, log_multiscales_m05=None):
        if tproj is not None:
            tproj = np.asfortranarray(tproj, dtype=np.float64)
            if tproj.shape[1] != d:  # lib/cov_se_fat.ml:38-48
                raise ValueError("Cov_se_fat.Params.create: tproj projection (%d) disagrees with "
                                 "target dimension d (%d)" % (tproj.shape[1], d))
        if log_multiscales_m05 is not None:
            log_multiscales_m05 = np.asfortranarray(log_multiscales_m05, dtype=np.float64)
            if log_multiscales_m05.shape[0] != d:
                raise ValueError("Cov_se_fat.Params.create: log_multiscales_m05 has %d rows, d = %d"
                                 % (log_multiscales_m05.shape[0], d))
        if log_hetero_skedasticity is not None:
            log_hetero_skedasticity = np.ascontiguousarray(log_hetero_skedasticity, dtype=np.float64)
        return Params(int(d), float(log_sf2), tproj, log_hetero_skedasticity, log_multiscales_m05)


@dataclass(frozen=True, eq=False)
class Kernel:
    """Cov_se_fat.Eval.Kernel.t (lib/cov_se_fat.ml:55-75)."""
    params: Params
    sf2: float

    @staticmethod
    def create(params: Params) -> "Kernel":
        return Kernel(params, math.exp(params.log_sf2))

    def get_params(self) -> Params:
        return self.params


class Inducing_hyper(NamedTuple):
    ind: int
    dim: int


class Proj_hyper(NamedTuple):
    """`Proj {big_dim; small_dim}` (lib/cov_se_fat.ml:263-265), 1-based."""
    big_dim: int
    small_dim: int


class Log_hetero_skedasticity(NamedTuple):
    """`Log_hetero_skedasticity dim` (lib/cov_se_fat.ml:279), 1-based inducing index."""
    dim: int


class Log_multiscale_m05(NamedTuple):
    """`Log_multiscale_m05 {ind; dim}` (lib/cov_se_fat.ml:280), 1-based."""
    ind: int
    dim: int


Hyper = Union[str, Inducing_hyper, Proj_hyper, Log_hetero_skedasticity, Log_multiscale_m05]
LOG_SF2 = "Log_sf2"


def create_default_kernel_params(inputs, n_inducing, rng=None) -> Params:
    """Eval.Inputs.create_default_kernel_params (lib/cov_se_fat.ml:191-213): d = min(big_dim, 10), a random projection
    scaled by the inverse row means of the inputs, log_hetero_skedasticity -5, log_multiscales_m05 0.  The reference
    draws from OCaml's global `Random` state; here a numpy Generator (or seed) supplies the uniform(-1, 1) numbers, in
    the reference's order: tproj row by row, then log_sf2."""
    rng = rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)
    x = np.asarray(inputs, dtype=np.float64)
    big_dim, n_inputs = x.shape
    d = min(big_dim, 10)
    factor = float(n_inputs) / float(big_dim)
    tproj = np.empty((big_dim, d), order="F")
    for r in range(big_dim):
        mean_factor = factor / float(np.sum(x[r, :]))
        tproj[r, :] = mean_factor * (rng.uniform(0.0, 2.0, size=d) - 1.0)
    log_sf2 = float(rng.uniform(0.0, 2.0) - 1.0)
    return Params.create(d, log_sf2, tproj=tproj, log_hetero_skedasticity=np.full(n_inducing, -5.0),
                         log_multiscales_m05=np.zeros((d, n_inducing), order="F"))


def project(kernel: Kernel, inputs):
    """Eval.Inputs.project (lib/cov_se_fat.ml:215-218): tproj^T inputs, or the inputs themselves without a projection."""
    x = np.asfortranarray(inputs, dtype=np.float64)
    tp = kernel.params.tproj
    return x if tp is None else np.asfortranarray(tp.T @ x)


def create_inducing(kernel: Kernel, inputs):
    """Eval.Inputs.create_inducing = project (lib/cov_se_fat.ml:220): inducing points live in the projected space."""
    return project(kernel, inputs)


def kernel_space_dim(kernel: Kernel, inputs) -> int:
    return kernel.params.d


def tproj_of(kernel: Kernel):
    return kernel.params.tproj


class HyperModule:
    """Cov_se_fat.Deriv.Hyper (lib/cov_se_fat.ml:287-407): [Log_sf2; inducing (ind-major); Proj (big-major)]."""

    @staticmethod
    def get_all(kernel: Kernel, inducing, _inputs=None):
        d = kernel.params.d
        m = inducing.shape[1]
        hypers = [LOG_SF2]
        for ind in range(1, m + 1):
            for dim in range(1, d + 1):
                hypers.append(Inducing_hyper(ind, dim))
        tproj = kernel.params.tproj
        if tproj is not None:
            for big in range(1, tproj.shape[0] + 1):
                for small in range(1, d + 1):
                    hypers.append(Proj_hyper(big, small))
        if kernel.params.log_hetero_skedasticity is not None:
            for i in range(1, m + 1):
                hypers.append(Log_hetero_skedasticity(i))
        if kernel.params.log_multiscales_m05 is not None:
            for ind in range(1, m + 1):
                for dim in range(1, d + 1):
                    hypers.append(Log_multiscale_m05(ind, dim))
        return hypers

    @staticmethod
    def get_value(kernel: Kernel, inducing, _inputs, hyper):
        if hyper == LOG_SF2:
            return kernel.params.log_sf2
        if isinstance(hyper, Proj_hyper):
            if kernel.params.tproj is None:  # lib/cov_se_fat.ml:344-347
                raise RuntimeError("Deriv.Hyper.option_get_value: tproj not supported")
            return float(kernel.params.tproj[hyper.big_dim - 1, hyper.small_dim - 1])
        if isinstance(hyper, Log_hetero_skedasticity):
            if kernel.params.log_hetero_skedasticity is None:
                raise RuntimeError("Deriv.Hyper.option_get_value: log_hetero_skedasticity not supported")
            return float(kernel.params.log_hetero_skedasticity[hyper.dim - 1])
        if isinstance(hyper, Log_multiscale_m05):
            if kernel.params.log_multiscales_m05 is None:
                raise RuntimeError("Deriv.Hyper.option_get_value: log_multiscales_m05 not supported")
            return float(kernel.params.log_multiscales_m05[hyper.dim - 1, hyper.ind - 1])
        return float(inducing[hyper.dim - 1, hyper.ind - 1])

    @staticmethod
    def set_values(kernel: Kernel, inducing, inputs, hypers, values):
        log_sf2 = kernel.params.log_sf2
        tproj = None
        new_inducing = None
        log_het = None
        log_ms = None
        for h, v in zip(hypers, values):
            if h == LOG_SF2:
                log_sf2 = float(v)
            elif isinstance(h, Log_multiscale_m05):
                if log_ms is None:
                    if kernel.params.log_multiscales_m05 is None:
                        raise RuntimeError("Deriv.Hyper.option_get_value: log_multiscales_m05 not supported")
                    log_ms = np.array(kernel.params.log_multiscales_m05, dtype=np.float64, order="F", copy=True)
                log_ms[h.dim - 1, h.ind - 1] = v
            elif isinstance(h, Log_hetero_skedasticity):
                if log_het is None:
                    if kernel.params.log_hetero_skedasticity is None:
                        raise RuntimeError("Deriv.Hyper.option_get_value: log_hetero_skedasticity not supported")
                    log_het = np.array(kernel.params.log_hetero_skedasticity, dtype=np.float64, copy=True)
                log_het[h.dim - 1] = v
            elif isinstance(h, Proj_hyper):
                if tproj is None:
                    if kernel.params.tproj is None:
                        raise RuntimeError("Deriv.Hyper.option_get_value: tproj not supported")
                    tproj = np.array(kernel.params.tproj, dtype=np.float64, order="F", copy=True)
                tproj[h.big_dim - 1, h.small_dim - 1] = v
            else:
                if new_inducing is None:
                    new_inducing = np.array(inducing, dtype=np.float64, order="F", copy=True)
                new_inducing[h.dim - 1, h.ind - 1] = v
        params = Params(kernel.params.d, log_sf2, kernel.params.tproj if tproj is None else tproj,
                        kernel.params.log_hetero_skedasticity if log_het is None else log_het,
                        kernel.params.log_multiscales_m05 if log_ms is None else log_ms)
        return Kernel.create(params), (inducing if new_inducing is None else new_inducing), inputs

    @staticmethod
    def index_of(kernel: Kernel, inducing, hyper) -> int:
        d = kernel.params.d
        m = inducing.shape[1]
        if hyper == LOG_SF2:
            return 0
        nproj = 0 if kernel.params.tproj is None else kernel.params.tproj.shape[0] * d
        if isinstance(hyper, Proj_hyper):
            return 1 + d * m + (hyper.big_dim - 1) * d + (hyper.small_dim - 1)
        if isinstance(hyper, Log_hetero_skedasticity):
            return 1 + d * m + nproj + (hyper.dim - 1)
        if isinstance(hyper, Log_multiscale_m05):
            nhet = 0 if kernel.params.log_hetero_skedasticity is None else m
            return 1 + d * m + nproj + nhet + (hyper.ind - 1) * d + (hyper.dim - 1)
        return 1 + (hyper.ind - 1) * d + (hyper.dim - 1)


def eval_args(kernel: Kernel):
    return dict(log_ell=0.0, log_sf2=kernel.params.log_sf2, tproj=kernel.params.tproj,
                log_hetero_skedasticity=kernel.params.log_hetero_skedasticity,
                log_multiscales_m05=kernel.params.log_multiscales_m05)
