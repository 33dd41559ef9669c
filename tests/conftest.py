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


def fortran_exe(name):
    """Path of a binary of the Fortran drop-in layer (ttcross_amd/fortran/build).  Where amdflang exists the layer is
    part of the product: a missing binary is built on the spot and a build failure FAILS the test; only a box without
    the compiler skips."""
    import shutil
    fdir = os.path.join(ROOT, "ttcross_amd", "fortran")
    exe = os.path.join(fdir, "build", name)
    if not os.path.exists(exe):
        if not (shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang")):
            pytest.skip("no amdflang on this box: the Fortran drop-in layer cannot be built")
        p = subprocess.run(["make", "-s", "-C", fdir], capture_output=True, text=True)
        assert p.returncode == 0 and os.path.exists(exe), "Fortran drop-in layer failed to build:\n" + p.stdout[-2000:] + p.stderr[-2000:]
    return exe
