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


@pytest.fixture(scope="session")
def ctx():
    """A device context on cuda:0.  Fails loudly when libsylph_hip.so or the GPU is missing."""
    import sylph_amd
    try:   # tests that also use torch on the GPU: let torch bring its HIP runtime up first, as bench.py does
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    c = sylph_amd.Context(0)
    yield c
    c.close()
