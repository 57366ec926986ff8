"""pytest configuration: markers, oracle / product library loaders.

The oracle (oracle/*.so) is test infrastructure; the product library
(linreg-mpc_amd/csrc/liblinreg_gc.so) is what is under test.  Nothing here
reads /root/reference at run time.
"""
import ctypes
import os
import subprocess
import sys

import pytest

try:                   # before anything loads liblinreg_gc.so: one HIP runtime per process (torch ships its own
    import torch       # libamdhip64 under the same SONAME; whichever loads first serves both), as bench.py does
except ImportError:    # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))


# a party whose peers never come up gives up after this many seconds (host/net.c; default 300): the multi-process tests
# wait 120-300 s for their children, and a run that cannot connect should fail with the parties' own messages, not with
# a TimeoutExpired
os.environ.setdefault("LINREG_CONNECT_TIMEOUT", "60")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _make(target_dir, target):
    path = os.path.join(target_dir, target)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", target_dir, target])
    return path


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.load()


@pytest.fixture(scope="session")
def gccpu():
    import gccpu as g
    return g.load()


@pytest.fixture(scope="session")
def lgc():
    """the product binding; builds liblinreg_gc.so in-tree if it is missing"""
    so = os.path.join(ROOT, "linreg-mpc_amd", "csrc", "liblinreg_gc.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so)])
    import linreg_gc
    return linreg_gc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
