"""Exact vs TTX_ARITH=fast on a few cases: leading identical sweeps, value differences, wall time (development probe)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from ttcross_amd import drivers as D, engine as E

def run(s, r, piv, nproc, arith):
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=nproc, arith=arith)
    tt.run()          # warm
    tt.run()
    return tt

def cmp(name, s, r, piv, nproc):
    a = run(s, r, piv, nproc, "exact"); b = run(s, r, piv, nproc, "fast")
    assert b.arith == "fast", b.arith
    ra, rb = a.sweeps(), b.sweeps()
    ta, tb = a.tapes(), b.tapes()
    k = 0
    for x, y in zip(ra[1:], rb[1:]):
        i = x["it"] - 1
        if i < len(tb) and np.array_equal(ta[i], tb[i]) and x["neval"] == y["neval"]: k += 1
        else: break
    dv = max(abs(x["val"] - y["val"]) / max(abs(x["val"]), 1e-300) for x, y in zip(ra[:k + 1], rb[:k + 1]))
    va, vb = a.quad(s["quad"]), b.quad(s["quad"])
    print(f"{name}: sweeps {len(ra)-1}/{len(rb)-1} identical-leading {k} max-rel-dval(leading) {dv:.2e} integral rel diff {abs(va-vb)/abs(va):.2e} "
          f"neval {a.neval}/{b.neval} time exact {a.seconds*1e3:.1f} ms fast {b.seconds*1e3:.1f} ms", flush=True)

# point evaluator
rng = np.random.default_rng(1)
for kind, m, n in [("d", 12, 33), ("e", 40, 17), ("d", 100, 33)]:
    s = D.ising_setup(kind, m, n)
    ind = rng.integers(1, n + 1, size=(2000, m - 1)).astype(np.int32)
    fe = E.k_eval(s["fun_id"], s["n"], s["par"], ind); ff = E.k_eval(s["fun_id"], s["n"], s["par"], ind, arith="fast")
    ok = fe != 0
    print(kind, m, "point evaluator max rel", np.max(np.abs(ff[ok] - fe[ok]) / np.abs(fe[ok])), "zeros", int((~ok).sum()), "fast-nonzero-where-exact-zero", int((ff[~ok] != 0).sum()), flush=True)

cases = [("d6", D.ising_setup("d", 6, 33), 12, 2, 1), ("e5", D.ising_setup("e", 5, 33), 12, 2, 1), ("d12", D.ising_setup("d", 12, 33), 10, 2, 1),
         ("d32", D.ising_setup("d", 32, 33), 12, 2, 1), ("d60g4", D.ising_setup("d", 60, 9), 6, 2, 4), ("e9g7", D.ising_setup("e", 9, 33), 12, 3, 7),
         ("d4full", D.ising_setup("d", 4, 11), 5, -1, 1), ("d100", D.ising_setup("d", 100, 17), 10, 3, 1),
         ("d64r24", D.ising_setup("d", 64, 51), 24, 2, 8),
         ("mvn6", D.box_setup("mvn", 6, 33), 12, 2, 1), ("mvn9g2", D.box_setup("mvn", 9, 17), 10, 3, 2), ("mvn32", D.box_setup("mvn", 32, 33), 20, 2, 4)]
only = sys.argv[1:] 
for c in cases:
    if only and c[0] not in only: continue
    cmp(*c)
