import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import cpu_ext
    cpu_ext.build()
    return cpu_ext


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["exact", "split16"])
def mathmode(request):
    """Runs a test once per arithmetic mode of the ST-GCN kernels (pose2room_amd.p2rnet.math_mode): the exact-fp32 default
    and the opt-in split16 mode.  The golden tests (G3-G10) take this fixture: same fixtures, same tolerances, both modes."""
    from pose2room_amd.p2rnet import math_mode
    with math_mode.use(request.param):
        yield request.param
    math_mode.reset()
