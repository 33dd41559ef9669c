"""Edge-size probe: each case in its own process (a GPU fault must not take the others down); prints ok / the error message."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from ttcross_amd import drivers as D, engine as E
kind, d, n, r, piv, ng = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
try:
    s = D.ising_setup(kind, d + 1, n) if kind in ("c", "d", "e") else D.box_setup(kind, d, n)
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=ng)
    tt.run()
    print("ok", kind, d, n, r, piv, ng, "neval", tt.neval, "sweeps", len(tt.sweeps()), "value", tt.quad(s["quad"]), flush=True)
except E.TTXError as e:
    print("refused", kind, d, n, r, piv, ng, "--", str(e)[:160], flush=True)
