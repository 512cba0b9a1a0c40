import os
import sys

import pytest

# torch first: it bundles its own HIP runtime, libgprhip.so is linked against the system one, and only this order lets one
# copy serve both -- INTEGRATION.md, "hosts that also load torch"; the library refuses the other order by name
# (gprhip_problem_create: "two HIP runtimes are mapped ...", tests/test_abi.py::test_two_hip_runtimes_are_refused_by_name).
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
