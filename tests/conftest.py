import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _clean_device_status(request):
    """The library's device status word (mfg_status) is sticky by design; a -m gpu test that provokes it on purpose must
    not fail the tests after it."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        try:
            import torch
            if torch.cuda.is_available():
                from discrete_mean_field_game_amd import ops
                ops.clear_status()
        except Exception:
            pass
