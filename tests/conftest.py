import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def qe():
    """The product package (ctypes binding over libqe_gpu.so)."""
    import importlib
    try:
        # torch brings a HIP runtime of its own; loaded after the engine's, a process ends up with two and the
        # second to initialise finds no device.  Loaded first, both share torch's.
        import torch  # noqa: F401
    except ImportError:
        pass
    return importlib.import_module("query-engine_amd")


@pytest.fixture(scope="session")
def ctx(qe):
    """One context on device 0; fails loudly (no CPU fallback) when there is no GPU."""
    c = qe.Context(0)
    yield c
    c.close()
