import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O
from ttcross_amd import drivers as D, engine as E
kind = sys.argv[1]; d = int(sys.argv[2])
if kind in ("c", "d", "e"):
    s = D.ising_setup(kind, d + 1, 3)
else:
    s = D.box_setup(kind, d, 2)
    if kind == "mvn": s["aux"] = O.mvn_init(d)
tt = E.TTCross(s["n"], s["fun_id"], s["par"], 2, pivoting=1, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=1)
tt.run()
print("ok", kind, d, tt.neval, flush=True)
