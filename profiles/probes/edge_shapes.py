"""Edge-shape probe (one case per process): ragged / tiny mode sizes, d = 1, 2, groups = d - 1, accuracy switches."""
import sys, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from ttcross_amd import drivers as D, engine as E
n_list = json.loads(sys.argv[1]); ident = float(sys.argv[2]); r = int(sys.argv[3]); piv = int(sys.argv[4]); ng = int(sys.argv[5])
acc = float(sys.argv[6]) if len(sys.argv) > 6 else 500 * D.EPS
nmax = n_list[0]
try:
    x, w = D.lgwt(nmax)
    par = np.zeros(2 * nmax + 1)
    par[:nmax] = (x + 1.0) / 2
    par[nmax:2 * nmax] = 0.5 * w * float(max(nmax // 2, 1))
    par[2 * nmax] = ident
    quad = [np.full(n, 1.0 / float(max(nmax // 2, 1))) for n in n_list]
    tt = E.TTCross(list(n_list), E.TTX_FUN_ISING, par, r, pivoting=piv, accuracy=acc, quad=quad, nproc=ng)
    tt.run()
    print("ok", n_list, ident, r, piv, ng, "neval", tt.neval, "sweeps", len(tt.sweeps()), "ranks", list(tt.ranks()), "value", tt.quad(quad), flush=True)
except E.TTXError as e:
    print("refused", n_list, ident, r, piv, ng, "--", str(e)[:170], flush=True)
