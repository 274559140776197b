import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _make(directory, target=None, env=None):
    cmd = ["make", "-C", directory] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/liboracle.so), built on demand."""
    from tests import _util
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        _make(os.path.join(ROOT, "oracle"))
    return _util.Oracle(path)


@pytest.fixture(scope="session")
def emu_lib():
    """libmrzgpu compiled by g++ against the wave64 emulator (tests/emu)."""
    import modern_rzip_amd as m
    path = os.path.join(ROOT, "tests", "emu", "libmrzgpu_emu.so")
    if not os.path.exists(path):
        _make(os.path.join(ROOT, "modern-rzip_amd", "csrc"), "emu")
    return m.load_library(path)


@pytest.fixture(scope="session")
def gpu_lib():
    """The real libmrzgpu.so; GPU tests fail loudly if it is missing."""
    import modern_rzip_amd as m
    return m.load_library()
