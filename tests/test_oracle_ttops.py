"""Oracle restatement of the tt_lib utilities (ort / svd / norm / dot / tijk) against the output of the GENUINE
reference (tests/golden/ttops_*.txt, made by tests/golden/ref_ttops.f90 through make_golden.sh).  The TT is the one
dtt_dmrgg builds for Ising C_m; LAPACK's dgesvd is restated as a Jacobi SVD, hence rounding-level tolerances."""
import os

import numpy as np
import pytest

import oracle_lib as O
from golden_util import GOLDEN


def _fixture(name):
    out = {}
    for line in open(os.path.join(GOLDEN, name)):
        t = line.split()
        if not t:
            continue
        out.setdefault(t[0], []).append(t[1:])
    return out


def _ising_tt(m, n, r, piv):
    from ttcross_amd import drivers as D
    s = D.ising_setup("c", m, n)
    oo = O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=piv, accuracy=s["acc"])        # no quad, as the fixture driver
    return oo["cores"], n


def probe_indices(m, n, k):
    return [(5 * k + 3 * i + i * i * k) % n + 1 for i in range(1, m)]


@pytest.mark.parametrize("m,n,r,piv", [(6, 33, 12, 2), (10, 25, 16, 2)])
def test_oracle_ttops_vs_reference(oracle_built, m, n, r, piv):
    fx = _fixture(f"ttops_C_{m}_{n}_{r}_{piv}.txt")
    cores, n = _ising_tt(m, n, r, piv)
    tt = O.OracleTT(cores)
    assert list(tt.ranks) == [int(x) for x in fx["ranks0"][0]]
    nrm0 = float(fx["norm0"][0][0])
    assert abs(tt.norm() - nrm0) <= 1e-12 * nrm0
    assert abs(tt.dot(tt) - float(fx["dot00"][0][0])) <= 1e-12 * nrm0 ** 2
    t1 = O.OracleTT(cores)
    t1.ort()
    assert list(t1.ranks) == [int(x) for x in fx["ranks_ort"][0]]
    assert abs(t1.norm() - float(fx["norm_ort"][0][0])) <= 1e-12 * nrm0
    for case, (tol, rmax) in enumerate([(1e-4, 0), (1e-8, 0), (1e-12, 5)], start=1):
        t2 = O.OracleTT(cores)
        t2.svd(tol, rmax)
        assert list(t2.ranks) == [int(x) for x in fx["ranks_svd"][case - 1][1:]], f"ranks after svd case {case}"
        assert abs(t2.norm() - float(fx["norm_svd"][case - 1][1])) <= 1e-11 * nrm0
        assert abs(tt.dot(t2) - float(fx["dot_svd"][case - 1][1])) <= 1e-11 * nrm0 ** 2
        for k in range(1, 5):
            row = [x for x in fx["elem"] if int(x[0]) == case and int(x[1]) == k][0]
            ind = probe_indices(m, n, k)
            scale = abs(float(row[2])) + 1e-300
            assert abs(tt.ijk(ind) - float(row[2])) <= 1e-12 * scale
            assert abs(t2.ijk(ind) - float(row[3])) <= 1e-7 * scale     # the rounded TT is defined up to ~tol*norm per element
