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


@pytest.fixture(scope="session", autouse=True)
def _gpu_runtime_first(request):
    """When GPU tests are selected, THIS process brings its GPU runtime up before any test runs: several tests start child processes
    on the device (the CLI, two ranks on one GPU), and a process that initialises HIP for the first time right behind them has been seen
    to find no device (gpurun call 23: `pytest tests/test_gpu_cli.py tests/test_gpu_fastq.py`, HIP error 100 at the first context)."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        request.getfixturevalue("ctx")
