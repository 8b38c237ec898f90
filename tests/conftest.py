import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # the product's setting (comic_amd/__init__.py), before any GPU call
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must fail loudly (not skip) on a GPU box without the HIP library;
    # on a CPU-only container they are deselected by the driver's `-m "not gpu"`.
    pass


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_parked_tensors(request):
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from tests import gpu_util
        gpu_util.release()
