import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _cap_threads():
    # The GPU box reports 256 logical CPUs; an OpenMP pool that wide makes the small oracle
    # convolutions crawl (minutes instead of seconds).  32 threads is what bench.py uses too.
    import torch

    torch.set_num_threads(min(32, os.cpu_count() or 1))


def pytest_configure(config):
    _cap_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")
