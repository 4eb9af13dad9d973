import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically when no device is present (the CPU tier is `-m "not gpu"`).
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """make sure both shared libraries exist (prebuilt files travel to the GPU box; make is a no-op there)"""
    lib = os.path.join(ROOT, "smelter_amd", "libsmelter_hip.so")
    orc = os.path.join(ROOT, "oracle", "liborc.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        subprocess.check_call(["make", "-s", "-j8", "-C", ROOT, "all"])
    yield


def nerr(a, b):
    """max-normalised error: max|a-b| / max|b|"""
    import numpy as np
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))
