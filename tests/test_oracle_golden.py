"""The oracle (oracle/ttx_oracle.c) against golden logs of the GENUINE reference (tests/golden/, made by
make_golden.sh from oracle/_ref) and against the analytic Ising values of test_crs_ising.f90:71-100.

What "pinned" means here: the oracle reproduces the reference's per-sweep (erank, n_evals, val) EXACTLY for
as long as the pivots stay above rounding noise (the reference sums dgemv/ddot in MKL's order, the oracle
in netlib order, so near-ties in the argmax eventually resolve differently), and the converged integral
to <= 1e-13 relative.  `EXACT_PREFIX` holds the number of leading sweeps required to match exactly."""
import os
import subprocess

import pytest

from golden_util import GOLDEN, golden_cases, parse_log

# name -> (min exactly-matching leading sweeps, rel tol on the final value)
EXACT_PREFIX = {
    "ising_C_6_33_20_2": (20, 1e-14), "ising_C_6_33_10_1": (10, 1e-14), "ising_C_5_17_8_0": (8, 1e-14),
    "ising_C_8_25_12_3": (12, 1e-14), "ising_D_6_33_12_2": (12, 1e-14), "ising_E_5_33_12_2": (12, 1e-14),
    "ising_D_12_33_10_2": (10, 1e-14), "ising_C_16_51_32_2": (17, 1e-13), "ising_C_64_51_32_2": (8, 1e-13),
    "ising_C_6_33_20_2_np2": (17, 1e-14), "ising_C_6_33_20_2_np4": (20, 1e-14),
    "ising_C_16_51_32_2_np8": (18, 1e-13), "ising_C_64_51_32_2_np8": (7, 1e-13),
    "ising_D_8_33_10_2_np3": (10, 1e-14),
    # pivoting = 0 down to the noise floor: acceptance depends on amax, which lib/dmrgg.f90:492-513 leaves alone
    "ising_C_16_33_24_0": (24, 1e-14), "ising_C_16_33_24_0_np5": (17, 1e-13),
    "ising_E_8_25_10_1": (10, 1e-14), "ising_D_10_17_8_0": (8, 1e-14), "ising_E_8_25_10_2_np3": (10, 1e-14),
    "ising_C_12_9_40_3": (9, 1e-14),     # tiny modes: pivots reach rounding noise after 9 sweeps (37 in all)
    "mvn_6_33_12_2": (8, 1e-3),      # inv_cov/det come from LAPACK in the reference; not converged at r=12
    "stdnorm_4_33_10_2": (2, 1e-13),
}


@pytest.mark.parametrize("name,argv", golden_cases(), ids=[c[0] for c in golden_cases()])
def test_oracle_matches_reference_log(oracle_built, name, argv):
    out = subprocess.run([os.path.join(oracle_built, "ttx_oracle")] + argv, capture_output=True, text=True,
                         check=True).stdout
    g_rows, g_val, _ = parse_log(open(os.path.join(GOLDEN, name + ".txt")).read())
    o_rows, o_val, _ = parse_log(out)
    need, tol = EXACT_PREFIX[name]
    assert len(g_rows) == len(o_rows), "number of sweeps differs"
    k = 0
    for a, b in zip(g_rows, o_rows):
        if a["erank"] == b["erank"] and a["neval"] == b["neval"] and abs(a["val"] - b["val"]) <= 2e-13 * abs(a["val"]):
            k += 1
        else:
            break
    assert k >= need, f"only {k} leading sweeps match the reference exactly (need {need})"
    assert abs(g_val - o_val) <= tol * abs(g_val)


def test_flang_rng_stream(oracle_built):
    """ttxo_flang_draw == amdflang's random_number, draw for draw (tests/golden/flang_rng.txt)."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(oracle_built, "libttx_oracle.so"))
    lib.ttxo_flang_draw.restype = ctypes.c_double
    lib.ttxo_flang_draw.argtypes = [ctypes.c_uint64]
    import struct
    want = [struct.unpack(">d", bytes.fromhex(l.strip()))[0] for l in open(os.path.join(GOLDEN, "flang_rng.txt"))]
    got = [lib.ttxo_flang_draw(k) for k in range(len(want))]
    assert got == want


@pytest.mark.parametrize("m,digits", [(5, 9.0), (6, 9.5), (8, 9.0)])
def test_oracle_ising_analytic(oracle_built, m, digits):
    """Known-answer: C_m against Bailey's table (the `correct digits` line of the reference driver)."""
    out = subprocess.run([os.path.join(oracle_built, "ttx_oracle"), "ising", "C", str(m), "33", "16", "2"],
                         capture_output=True, text=True, check=True).stdout
    d = [float(l.split(":")[1]) for l in out.splitlines() if l.startswith("correct digits")][0]
    assert d >= digits


@pytest.mark.parametrize("kind,m,n,r,piv,nproc", [("d", 24, 17, 8, 2, 1), ("e", 30, 9, 6, 3, 3), ("d", 40, 33, 6, 1, 2)])
def test_unit_skip_changes_no_bit(oracle_built, kind, m, n, r, piv, nproc):
    """ttxo_set_unit_skip (used only to make the D_256 fixture in reasonable time) leaves out pair factors that are exactly 1:
    tapes, every per-sweep record, the cores and the integral must be bit-identical with and without it."""
    import numpy as np
    import oracle_lib as O
    from ttcross_amd import drivers as D
    s = D.ising_setup(kind, m, n)
    runs = []
    for on in (0, 1):
        O.lib().ttxo_set_unit_skip(on)
        try:
            runs.append(O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=nproc))
        finally:
            O.lib().ttxo_set_unit_skip(0)
    a, b = runs
    assert np.array_equal(a["tapes"], b["tapes"]) and a["neval"] == b["neval"] and a["value"] == b["value"]
    assert a["sweeps"] == b["sweeps"]
    for x, y in zip(a["cores"], b["cores"]):
        assert np.array_equal(x, y)
