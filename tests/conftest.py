import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "whisper-medusa_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """libwm.so built in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    sys.path.insert(0, PKG)
    import build as wm_build
    return wm_build.build(verbose=False)


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)
