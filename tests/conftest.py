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
    """Both shared libraries are (re)built incrementally whenever a compiler is present, so a stale prebuilt binary can never be
    tested against newer sources (the .so files are git-ignored artefacts that travel to the GPU box; there make finds them up
    to date and does nothing).  Without hipcc the prebuilt files must exist."""
    import shutil
    lib = os.path.join(ROOT, "smelter_amd", "libsmelter_hip.so")
    orc = os.path.join(ROOT, "oracle", "liborc.so")
    have_hipcc = os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")
    if have_hipcc and shutil.which("make"):
        subprocess.check_call(["make", "-s", "-j8", "-C", ROOT, "all"])
    assert os.path.exists(lib) and os.path.exists(orc), "libsmelter_hip.so / liborc.so missing and no compiler to build them"
    yield


def nerr(a, b):
    """max-normalised error: max|a-b| / max|b|"""
    import numpy as np
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))
