#!/usr/bin/env python3
"""Recipe for the multi-rank reference build (test infrastructure, see ttx_oracle.h).

The fp64 sweep of the reference (lib/dmrgg.f90) lost its "share blocks to the RIGHT" step; the
multiprecision twin still has it (lib/dmrggmp.f90:572-629).  This script writes, into oracle/_ref/
only, a copy of lib/dmrgg.f90 with that block re-inserted after the "moving left" deallocation
(lib/dmrgg.f90:954-958).  The block is taken from the reference tree at build time and
transliterated to fp64 by text substitution -- no reference source is stored in this repository.

usage: make_ref_mpi.py /root/reference oracle/_ref/dmrgg_mpi.f90
"""
import re
import sys


def main(ref, out):
    src = open(f"{ref}/lib/dmrgg.f90").read().split("\n")
    mp = open(f"{ref}/lib/dmrggmp.f90").read().split("\n")

    # locate the block in the MP twin: from the 'RIGHT' comment to the matching deallocate check
    start = next(i for i, l in enumerate(mp) if "share blocks to the RIGHT" in l)
    end = next(i for i in range(start, len(mp)) if "fail to deallocate after moving right" in mp[i])
    block = mp[start:end + 1]

    subs = [
        (r"\bmpcopy\b", "dcopy"),
        (r"\bmp2_lual\b", "d2_lual"),
        (r"\bmp_dmrgg_fun\b", "dmrgg_fun"),
        (r"\*mpwds6", ""),
        (r"dble\(logten\*log\(abs\((.*)\)\)\)\)", r"dabs(\1))"),
    ]
    fixed = []
    for line in block:
        for pat, rep in subs:
            line = re.sub(pat, rep, line)
        fixed.append(line)

    # insertion point: after the error check that follows the LEFT-phase deallocate
    k = next(i for i, l in enumerate(src) if "fail to deallocate after moving left" in l)
    k = next(i for i in range(k, len(src)) if src[i].strip() == "end if") + 1
    patched = src[:k] + [""] + fixed + src[k:]
    open(out, "w").write("\n".join(patched))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
