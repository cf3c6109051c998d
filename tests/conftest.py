import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
PKG = os.path.join(ROOT, "hai719-raytracing_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _ensure_built():
    need = [os.path.join(PKG, "libhrt_host.so"), os.path.join(PKG, "libhrt.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], check=True)


@pytest.fixture(scope="session")
def hrt():
    _ensure_built()
    return importlib.import_module("hai719-raytracing_amd")


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def gpu(hrt):
    """Initialised HIP library; fails loudly (no skip, no fallback) when there is no GPU."""
    hrt.init(0)
    return hrt
