import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    from tests import margins
    margins.dump(ROOT)            # (how far inside their bounds the tolerance checks landed: gpurun_out/parity_margins.json)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _drain_device_between_tests(request):
    """GPU tests build and drop HIP graphs, streams and private memory pools: finish all device work and collect
    the garbage of a test before the next one starts (a graph destroyed by a later, unrelated allocation while its
    last replay is still in flight has taken the whole run down on ROCm 7.2)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
            from diffgfdn_amd.losses import raise_on_unit_grad_violation
            raise_on_unit_grad_violation()       # (every decay-loss backward of the test kept its unit-gradient promise)
