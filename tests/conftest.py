import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-quantity_amd")
QUANTITY = os.path.join(PKG, "quantity")          # holds the drop-in top-level modules common/, tools/, model/
GOLDEN = os.path.join(ROOT, "tests", "golden")

for p in (ROOT, QUANTITY, GOLDEN, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (oracle/fq_oracle.c via ctypes). Test infrastructure only."""
    from oracle import fq_oracle
    fq_oracle.build()
    return fq_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
