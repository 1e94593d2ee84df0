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


def _ensure_native_lib():
    """The HIP library is a build artefact (git-ignored).  hipcc cross-compiles gfx950 without a GPU, so
    build it on demand when a fresh checkout runs the tests; the product itself never builds lazily."""
    import subprocess
    csrc = os.path.join(PKG, "csrc")
    lib = os.path.join(PKG, "lib", "libfq_hip.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.isfile(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", csrc, "-s"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_native_lib()


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need the MI355X: on a box without one (the build container) they are skipped, not failed."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (oracle/fq_oracle.c via ctypes). Test infrastructure only."""
    from oracle import fq_oracle
    fq_oracle.build()
    return fq_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
