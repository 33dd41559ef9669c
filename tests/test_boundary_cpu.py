"""CPU-side checks of the drop-in boundary: the reference's own drivers must compile and link UNCHANGED against the
modules of ttcross_amd/fortran, and the integrands' exp must be the run-time library's."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def test_reference_drivers_compile_and_link_unchanged():
    """/root/reference/test_crs_{ising,stdnorm,mvn}.f90 against ttcross_amd/fortran/build + libttx.so
    (oracle/Makefile target `dropin`).  Only where the reference sources and amdflang exist (the build container)."""
    if not os.path.isdir("/root/reference/lib") or not (shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang")):
        pytest.skip("reference sources or amdflang not present on this box")
    fdir = os.path.join(ROOT, "ttcross_amd", "fortran")
    p = subprocess.run(["make", "-s", "-C", fdir], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for d in ("ising", "stdnorm", "mvn"):
        exe = os.path.join(ROOT, "oracle", "_ref", f"dropin_test_crs_{d}")
        if os.path.exists(exe):
            os.remove(exe)
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dropin"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for d in ("ising", "stdnorm", "mvn"):
        exe = os.path.join(ROOT, "oracle", "_ref", f"dropin_test_crs_{d}")
        assert os.path.exists(exe)
        syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
        assert "ttx_create" in syms and "ttx_run" in syms          # bound to the engine, not to a CPU path


def test_exp_host_instantiation_is_the_runtime_libm():
    """ttx_exp.h (the integrands' exp on the device) instantiated on the host == libm's exp, bit for bit: random
    arguments over the whole range the integrands can produce, the subnormal tail, and the special values."""
    from ttcross_amd import engine as E
    libm = ctypes.CDLL("libm.so.6")
    libm.exp.restype = ctypes.c_double
    libm.exp.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(7)
    x = np.concatenate([-rng.random(150000) * 800.0, (rng.random(50000) - 0.5) * 1500.0, -745.2 + rng.random(50000) * 40.0,
                        (rng.random(20000) - 0.5) * 1e-9, -rng.random(50000) * 40.0,
                        [0.0, -0.0, 1e-300, -1e-300, 709.78, 709.79, 710.0, -745.13, -745.14, -746.0, 1e308, -1e308,
                         float("inf"), -float("inf")]])
    got = E.exp_host(x)
    want = np.array([libm.exp(v) for v in x])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.isnan(E.exp_host(np.array([float("nan")]))[0])


def test_exp_table_is_reproducible():
    """ttx_exp_tab.h is what gen_exp_table.py generates from first principles."""
    import tempfile
    csrc = os.path.join(ROOT, "ttcross_amd", "csrc")
    with tempfile.TemporaryDirectory() as t:
        out = os.path.join(t, "tab.h")
        subprocess.run(["python3", os.path.join(csrc, "gen_exp_table.py"), out], check=True)
        assert open(out).read() == open(os.path.join(csrc, "ttx_exp_tab.h")).read()
