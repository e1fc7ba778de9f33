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


# The library parses its DDPM_* switches once per process (ddpm_reload_env parses them again): tests that flip a switch through
# monkeypatch get the reload for free, including the undo at teardown.
def _reload_switches():
    try:
        from ddpm_ood_amd import _lib

        _lib.reload_env()
    except Exception:
        pass


_setenv, _delenv, _undo = pytest.MonkeyPatch.setenv, pytest.MonkeyPatch.delenv, pytest.MonkeyPatch.undo


def _setenv_reload(self, name, value, prepend=None):
    _setenv(self, name, value, prepend)
    if name.startswith("DDPM_"):
        _reload_switches()


def _delenv_reload(self, name, raising=True):
    _delenv(self, name, raising)
    if name.startswith("DDPM_"):
        _reload_switches()


def _undo_reload(self):
    _undo(self)
    _reload_switches()


pytest.MonkeyPatch.setenv = _setenv_reload
pytest.MonkeyPatch.delenv = _delenv_reload
pytest.MonkeyPatch.undo = _undo_reload
