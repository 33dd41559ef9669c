"""More edge probes of tt_lib on uploaded trains: rank-deficient / zero cores, rank 128, accchk sizes."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as O
from ttcross_amd import engine as E, drivers as D
rng = np.random.default_rng(9)
def tt(n, r): return [rng.standard_normal((r[k], n[k], r[k + 1])) for k in range(len(n))]
def show(name, f):
    try:
        print("ok     ", name, "->", f(), flush=True)
    except E.TTXError as e:
        print("refused", name, "--", str(e)[:150], flush=True)
    except Exception as e:      # noqa: BLE001
        print("PYERR  ", name, "--", type(e).__name__, str(e)[:150], flush=True)
z = [np.zeros((1, 4, 3)), np.zeros((3, 5, 2)), np.zeros((2, 3, 1))]
show("zero train: norm / ort ranks / svd ranks", lambda: (E.TTCross.from_cores(z).norm(), list(E.TTCross.from_cores(z).ort().ranks()), list(E.TTCross.from_cores(z).svd(1e-8, 0).ranks())))
dup = tt([6, 6, 6], [1, 4, 4, 1]); dup[1][:, :, 1] = dup[1][:, :, 0]; dup[1][:, :, 3] = 0.0        # rank-deficient middle core
def chk(c, tol):
    t = E.TTCross.from_cores(c).svd(tol, 0); o = O.OracleTT(c); o.svd(tol, 0)
    return list(t.ranks()), list(o.ranks), abs(t.norm() - o.norm()) <= 1e-10 * max(o.norm(), 1e-300)
show("rank-deficient core: svd ranks vs oracle, norm", lambda: chk(dup, 1e-10))
big = tt([9, 9, 9, 9], [1, 9, 81, 9, 1])
show("ranks up to 81: svd 1e-12 vs oracle", lambda: chk(big, 1e-12))
b128 = tt([16, 16, 16], [1, 16, 128, 1]); b128[1] = rng.standard_normal((16, 16, 128))
show("rank 128 (capped by 16*16=256 rows): ort ranks, norm match", lambda: (list(E.TTCross.from_cores(b128).ort().ranks()), abs(E.TTCross.from_cores(b128).ort().norm() - O.OracleTT(b128).norm()) <= 1e-10 * O.OracleTT(b128).norm()))
s = D.ising_setup("c", 6, 17)
c6 = E.TTCross(s["n"], s["fun_id"], s["par"], 8, pivoting=2, accuracy=s["acc"], quad=s["quad"], tru=s["tru"]).run()
show("accchk nlot=1", lambda: c6.accchk(1)["einf"])
show("accchk nlot=0", lambda: c6.accchk(0))
show("accchk nlot=2e6", lambda: c6.accchk(2000000)["einf"])
show("accchk nlot<0", lambda: c6.accchk(-5))
show("quad after svd equals quad before (1e-9)", lambda: abs(c6.quad(s["quad"]) - E.TTCross.from_cores([c6.core(k) for k in range(1, 6)]).svd(1e-13, 0).quad(s["quad"])) <= 1e-9 * abs(c6.quad(s["quad"])))
