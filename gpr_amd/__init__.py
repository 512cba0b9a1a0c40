"""gpr_amd -- MI355X-native FITC/SPGP Gaussian-process core behind mmottl/gpr's module surface.

The compute path is the hand-written HIP library gpr_amd/libgprhip.so (C ABI: include/gprhip.h);
this package is the host-side mirror of the reference's functor interface for that path.
"""
from ._lib import (COV_SE_FAT, COV_SE_ISO, F32_BULK, F64, GprHipError, NotPositiveDefinite,  # noqa: F401
                   UntrustworthyCoefficients,
                   device_count, load, memory_plan)
from .problem import CHOLESKY_JITTER, Evaluation, Problem  # noqa: F401
from .context import Context, ShardedDeviceProblem  # noqa: F401
