import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture(autouse=True)
def _work_queues_left_clean(request):
    """After every GPU test: the residual-step work-queue slots of the library are all zero again (tts_diag_queue_nonzero)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import torch
    if not torch.cuda.is_available():
        return
    from ims_toucan_prosody_variance_amd import capi
    dirty = capi.lib().tts_diag_queue_nonzero()
    assert dirty == 0, f"{dirty} non-zero words left in the residual-step work queues (negative: HIP error)"
