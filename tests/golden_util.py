"""Parsing of the reference's per-sweep log lines (lib/dmrgg.f90:971-1008) for golden comparisons."""
import glob
import os
import re

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def parse_log(txt):
    rows = []
    for line in txt.splitlines():
        if "n_evals" not in line:
            continue
        t = line.split()
        it = int(re.match(r"\d+", t[0]).group())
        rows.append({"it": it, "dir": t[0][-2:], "erank": float(t[2]), "neval": int(t[6]),
                     "val": float(t[-1]) if " val " in line else None})
    fin = [l for l in txt.splitlines() if l.startswith("computed value")]
    value = float(fin[0].split()[2]) if fin else None
    nev = [l for l in txt.splitlines() if l.startswith("...with")]
    neval = int(nev[0].split()[1]) if nev else None
    return rows, value, neval


# golden logs whose oracle run takes minutes (single thread): compared through the stored oracle fixtures
# tests/golden/oracle_*.npz (make_oracle_fixture.py) instead of a live oracle run
LONG = {"mvn_128_33_50_2", "ising_D_256_101_64_5", "ising_D_256_101_64_5_np8"}


def golden_cases(include_long=False):
    """[(name, driver argv incl. nproc)] for every fixture made by make_golden.sh."""
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "*.txt"))):
        b = os.path.basename(f)[:-4]
        if b in LONG and not include_long:
            continue
        if b.split("_")[0] not in ("ising", "stdnorm", "mvn"):      # driver logs only (not accchk_/ttops_/zquad_/flang_rng)
            continue
        t = b.split("_")
        nproc = 1
        if t[-1].startswith("np"):
            nproc = int(t[-1][2:])
            t = t[:-1]
        out.append((b, t + [str(nproc)]))
    return out
