import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True, scope="module")
def _isolate_modules():
    """Process-global state must not leak from one test file into the next (measured: the full -m gpu suite took 18 min
    in one process against 5.5 min for the same files run separately): FLOWTRON_* / FT_* environment switches and the torch
    thread count are restored, and the caching allocator's pool (the full-width tests leave > 100 GB cached) is returned."""
    import gc
    import torch
    env = {k: v for k, v in os.environ.items() if k.startswith(("FLOWTRON_", "FT_"))}
    threads = torch.get_num_threads()
    yield
    for k in [k for k in os.environ if k.startswith(("FLOWTRON_", "FT_"))]:
        if k not in env:
            del os.environ[k]
    os.environ.update(env)
    torch.set_num_threads(threads)
    gc.collect()
    try:                                             # hand freed heap back to the OS (the oracle's autograd graphs are tens of GB)
        import ctypes
        ctypes.CDLL("libc.so.6").malloc_trim(0)
    except Exception:
        pass
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def pytest_collection_modifyitems(config, items):
    """The training-loop tests (DataLoader workers + per-batch host->device traffic) run FIRST: after the full-width oracle
    tests of the other files the same tests take 110-170 s each instead of 7-13 s (measured three times; the time is spent
    waiting in the batch's `.cuda()`), while nothing that runs after them is affected."""
    first = [it for it in items if "test_gpu_train_loop" in it.nodeid]
    rest = [it for it in items if "test_gpu_train_loop" not in it.nodeid]
    items[:] = first + rest
