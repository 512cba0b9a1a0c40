import os
import sys

import pytest

# torch first: it ships its own copy of the HIP runtime (torch/lib/libamdhip64.so), libgprhip.so is linked against the
# system one (/opt/rocm/lib).  Whichever is loaded first serves both; loaded in the other order (the library first, torch's
# GPU initialisation later -- e.g. `pytest tests/test_gpu_parity.py -k device_resident` after any other GPU test) torch
# reports "No HIP GPUs are available".  bench.py imports torch first for the same reason.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover  (CPU-only tooling without torch)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_available():
    import gpr_amd
    return gpr_amd.device_count() > 0
