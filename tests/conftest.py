import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    """Build the CPU oracle (test infrastructure) once per session."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    return os.path.join(ROOT, "oracle")
