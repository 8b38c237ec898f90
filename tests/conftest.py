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


@pytest.fixture(scope='session', autouse=True)
def _background_launches():
    """COMIC_TEST_NOISE=1: a thread keeps a stream of its own busy with GEMMs for the whole session.  The suite's results must not
    depend on how the kernels of different streams interleave; under this load a missing dependency between two lanes shows
    within a run or two (the race of the scheduled cnn_finetune backward: 1e-4 differences in every pair instead of one suite
    run in three)."""
    if os.environ.get('COMIC_TEST_NOISE') != '1':
        yield
        return
    import threading
    import torch
    stop = threading.Event()

    def loop():
        s = torch.cuda.Stream()
        a = torch.randn(2048, 2048, device='cuda:0')
        while not stop.is_set():
            with torch.cuda.stream(s):
                for _ in range(16):
                    a @ a
            s.synchronize()
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    yield
    stop.set()
    th.join(timeout=10)


@pytest.fixture(autouse=True)
def _release_parked_tensors(request):
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from tests import gpu_util
        gpu_util.release()
