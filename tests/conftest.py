import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The arena's hunt for its kinds of HBM is bounded per call by default (32 slabs / 50 ms: the product's first call
    # must be cheap, and the next calls continue the hunt).  The placement tests of this suite ask what the arena does
    # once it KNOWS the device's kinds, so the test session lifts the bound like bench.py does; the bounded default
    # itself is checked in a fresh process (test_gpu_perf.py: test_arena_default_hunt_is_bounded).
    os.environ.setdefault("PRT_ARENA_HUNT", "full")
    # the library's fault-injection hook (PRT_TEST_FAIL_UPDATE) exists only in processes that ask for it before the
    # library's first call (csrc/prt.hip: prt_system_update)
    os.environ.setdefault("PRT_TEST_HOOKS", "1")
    # A process that has hundreds of GiB of device memory mapped must not write a core file if it ever dies:
    # once a crash inside the HIP runtime filled the box's disk that way and took the following run with it.
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, resource.getrlimit(resource.RLIMIT_CORE)[1]))
    except (ImportError, ValueError, OSError):
        pass


# PRT_TESTS_ENGINE_ON_HOST=1 python -m pytest tests -m gpu ...: the `-m gpu` tests with the product's Python (engine.py, the
# drop-in layer) on the HOST build of libprt's sources (tests/hostemu/engine_on_host.py) -- a mode of the TEST SUITE for a
# box without a GPU (tests/test_hostemu_campaigns.py runs a selection this way); the product itself has no such mode.
ENGINE_ON_HOST = os.environ.get("PRT_TESTS_ENGINE_ON_HOST", "0") == "1"


@pytest.fixture(scope="session", autouse=ENGINE_ON_HOST)
def _engine_on_host():
    if not ENGINE_ON_HOST:
        yield None
        return
    from hostemu.engine_on_host import engine_on_host
    with engine_on_host() as engine:
        yield engine


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if ENGINE_ON_HOST:
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no HIP device is visible")
    return torch.device("cuda", 0)
