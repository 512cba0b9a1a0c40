"""Host-side mirror of Gpr.Cov_se_iso (reference lib/cov_se_iso.ml, lib/cov_se_iso.mli).

Only the parts of the Specs.Deriv instance that are *data* live here (parameters, the
hyper-parameter enumeration and get/set).  The covariance arithmetic itself runs in HIP
(gpr_amd/csrc/cov_kernels.hip); there is deliberately no host implementation of it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import NamedTuple, Union

import numpy as np

from ._lib import COV_SE_ISO

COV_KIND = COV_SE_ISO


@dataclass(frozen=True)
class Params:
    """Cov_se_iso.Params.t (lib/cov_se_iso.ml:23-25)."""
    log_ell: float
    log_sf2: float


@dataclass(frozen=True)
class Kernel:
    """Cov_se_iso.Eval.Kernel.t (lib/cov_se_iso.ml:33-44)."""
    params: Params
    inv_ell2: float
    inv_ell2_05: float
    log_sf2: float
    sf2: float

    @staticmethod
    def create(params: Params) -> "Kernel":
        inv_ell2 = math.exp(-2.0 * params.log_ell)
        return Kernel(params, inv_ell2, -0.5 * inv_ell2, params.log_sf2, math.exp(params.log_sf2))

    def get_params(self) -> Params:
        return self.params


class Inducing_hyper(NamedTuple):
    """`Inducing_hyper {ind; dim}` -- both 1-based like the reference (lib/cov_se_iso.ml:27)."""
    ind: int
    dim: int


Hyper = Union[str, Inducing_hyper]  # "Log_ell" | "Log_sf2" | Inducing_hyper
LOG_ELL, LOG_SF2 = "Log_ell", "Log_sf2"


def create_default_kernel_params(_inputs=None, n_inducing=None) -> Params:
    """lib/cov_se_iso.ml:122-123."""
    return Params(log_ell=0.0, log_sf2=0.0)


def create_inducing(_kernel, inputs):
    """Eval.Inputs.create_inducing (lib/cov_se_iso.ml:120): chosen inputs are the inducing points."""
    return np.asfortranarray(inputs, dtype=np.float64)


def kernel_space_dim(kernel, inputs) -> int:
    return inputs.shape[0]


def tproj_of(kernel):
    return None


class HyperModule:
    """Cov_se_iso.Deriv.Hyper (lib/cov_se_iso.ml:185-230)."""

    @staticmethod
    def get_all(_kernel, inducing, _inputs=None):
        d, m = inducing.shape
        hypers = [LOG_ELL, LOG_SF2]
        for ind in range(1, m + 1):
            for dim in range(1, d + 1):
                hypers.append(Inducing_hyper(ind, dim))
        return hypers

    @staticmethod
    def get_value(kernel: Kernel, inducing, _inputs, hyper):
        if hyper == LOG_ELL:
            return kernel.params.log_ell
        if hyper == LOG_SF2:
            return kernel.params.log_sf2
        return float(inducing[hyper.dim - 1, hyper.ind - 1])

    @staticmethod
    def set_values(kernel: Kernel, inducing, inputs, hypers, values):
        log_ell, log_sf2 = kernel.params.log_ell, kernel.params.log_sf2
        new_inducing = None
        for h, v in zip(hypers, values):
            if h == LOG_ELL:
                log_ell = float(v)
            elif h == LOG_SF2:
                log_sf2 = float(v)
            else:
                if new_inducing is None:  # lazy copy, lib/cov_se_iso.ml:213-219
                    new_inducing = np.array(inducing, dtype=np.float64, order="F", copy=True)
                new_inducing[h.dim - 1, h.ind - 1] = v
        new_kernel = Kernel.create(Params(log_ell=log_ell, log_sf2=log_sf2))
        return new_kernel, (inducing if new_inducing is None else new_inducing), inputs

    @staticmethod
    def index_of(kernel, inducing, hyper) -> int:
        """Position of `hyper` in get_all's order (the layout of the device gradient vector)."""
        d = inducing.shape[0]
        if hyper == LOG_ELL:
            return 0
        if hyper == LOG_SF2:
            return 1
        return 2 + (hyper.ind - 1) * d + (hyper.dim - 1)


def eval_args(kernel: Kernel):
    """Scalar hyper-parameters handed to the device evaluation."""
    return dict(log_ell=kernel.params.log_ell, log_sf2=kernel.params.log_sf2, tproj=None,
                log_hetero_skedasticity=None, log_multiscales_m05=None)
