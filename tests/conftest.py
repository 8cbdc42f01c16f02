import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library (HIP).  GPU tests call through this C ABI only."""
    from frb_baseband_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY host emulator build of the engine + generic kernels (tests/emu)."""
    import subprocess
    from frb_baseband_amd import _lib
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu_dir])
    return _lib.load(os.path.join(emu_dir, "libfrbch_emu.so"))
