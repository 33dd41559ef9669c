"""One NaN / Inf case of the BUILT-IN device integrands in a process of its own (tests/test_gpu_boundary.py starts one per case,
so that a GPU fault -- what the first such input caused in round 2 -- fails that case instead of taking the test runner down).

    python tests/nan_worker.py CASE        exit code 0: the run ended, every pivot in range

CASE = <integrand>_<where>[_fast]: c_node (NaN among the nodes of Ising C; TTX_SWEEP picks cluster / fused / chain),
d_node, e_node (Ising D / E: the division kernels), d_weight (NaN weight: nodes stay in [0,1], so the short division, the wave
teams and -- with _fast -- the table evaluators run), mvn_inf (an infinite entry of the inverse covariance)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from ttcross_amd import drivers as D
    from ttcross_amd import engine as E
    case = sys.argv[1]
    fast = case.endswith("_fast")
    base = case[:-5] if fast else case
    piv, nproc = int(sys.argv[2]), int(sys.argv[3])
    if base.startswith("mvn"):
        d, n, r = 7, 9, 6
        s = D.box_setup("mvn", d, n)
        s["aux"] = s["aux"].copy()
        s["aux"][d + 1 + d * 2] = np.inf                 # inv_cov(2,3); det (and the normalisation check) untouched
    else:
        kind = base[0]
        m, n, r = (12, 9, 6) if kind == "c" else (22, 9, 6)
        s = D.ising_setup(kind, m, n)
        s["par"] = s["par"].copy()
        nn = s["n"][0]
        s["par"][3 if base.endswith("node") else nn + 2] = np.nan
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], aux=s["aux"], nproc=nproc,
                   arith="fast" if fast else None)
    tt.run()
    tp = tt.tapes()
    act = tp[:, 1:tt.d, :]
    nmax = max(s["n"])
    ok = tp.shape[0] >= 1 and bool(((act == -1) | ((act >= 1) & (act <= max(nmax, r + 1)))).all()) and all(1 <= rk <= r for rk in tt.ranks())
    print(f"{case} piv={piv} nproc={nproc} path={tt.sweep_path()} arith={tt.arith} sweeps={tp.shape[0]} ranks_max={int(max(tt.ranks()))} {'OK' if ok else 'BAD'}", flush=True)
    tt.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
