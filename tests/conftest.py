import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the product (HIP library + CLI) and the oracle are built; a no-op when up to date."""
    import subprocess
    lib = os.path.join(ROOT, "lz77_amd", "liblz77_mi355x.so")
    cli = os.path.join(ROOT, "lz77_amd", "lz77")
    var = os.path.join(ROOT, "lz77_amd", "liblz77_mi355x_variants.so")
    if not (os.path.exists(lib) and os.path.exists(cli) and os.path.exists(var)):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "lz77_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liblz77_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")
