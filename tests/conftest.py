"""pytest configuration: `gpu` marker, import paths, and building the oracle (checker) libraries."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: build it here, never from product code
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    # the product's native pieces (hipcc / g++ cross-compile without a GPU; no-ops when up to date)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "openmm-velocityverlet_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "platforms", "hip")], check=True)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
