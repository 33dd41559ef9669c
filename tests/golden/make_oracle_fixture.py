#!/usr/bin/env python3
"""Per-sweep fixtures of the CPU ORACLE for runs that are too long to repeat inside the test suites.

    python tests/golden/make_oracle_fixture.py mvn 128 33 50 2 1      # -> tests/golden/oracle_mvn_128_33_50_2_np1.npz
    python tests/golden/make_oracle_fixture.py mvn 128 33 50 2 4
    python tests/golden/make_oracle_fixture.py d 256 101 64 5 8       # -> tests/golden/oracle_ising_D_256_101_64_5_np8.npz (~20 min)

The oracle (oracle/ttx_oracle.c, single thread) takes minutes at BASELINE config 4, so it runs HERE, in the
build container, and the GPU tests compare against the stored records: per sweep (it, erank, neval, val, amax,
pivotmax), the pivot tapes, the final ranks, neval and the integral.  Inputs are rebuilt by the test from the
same driver set-up (ttcross_amd.drivers.box_setup + oracle_lib.mvn_init), so only outputs are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle_lib as O  # noqa: E402
from ttcross_amd import drivers as D  # noqa: E402


def main():
    kind, d, n, r, piv, nproc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    tag = f"{kind}_{d}_{n}_{r}_{piv}"
    if kind in ("c", "d", "e"):       # Ising C/D/E: `d` is the integral's index (test_crs_ising KIND INDEX N RANK PIV)
        s = D.ising_setup(kind, d, n)
        tag = f"ising_{kind.upper()}_{d}_{n}_{r}_{piv}"
        O.lib().ttxo_set_unit_skip(1)          # bit-neutral shortcut for nodes in [0,1] (oracle/ttx_oracle.c); D_256 would take hours without
    else:
        s = D.box_setup(kind, d, n)
    if kind == "mvn":
        s["aux"] = O.mvn_init(d)
    oo = O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=nproc)
    sw = oo["sweeps"]
    out = os.path.join(HERE, f"oracle_{tag}_np{nproc}.npz")
    np.savez_compressed(out, it=np.array([a["it"] for a in sw]), erank=np.array([a["erank"] for a in sw]),
                        neval=np.array([a["neval"] for a in sw], dtype=np.int64), val=np.array([a["val"] for a in sw]),
                        amax=np.array([a["amax"] for a in sw]), pivotmax=np.array([a["pivotmax"] for a in sw]),
                        tapes=oo["tapes"].astype(np.int16), r=oo["r"], total_neval=np.int64(oo["neval"]), value=np.float64(oo["value"]),
                        seconds=np.float64(oo["seconds"]))
    print(out, len(sw), oo["neval"], repr(oo["value"]), f"{oo['seconds']:.1f} s")


if __name__ == "__main__":
    main()
