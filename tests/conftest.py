import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: (re)build it when a C compiler is around."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield
