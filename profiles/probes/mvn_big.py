import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O
from ttcross_amd import drivers as D, engine as E
d = int(sys.argv[1]); n = int(sys.argv[2]); r = int(sys.argv[3])
s = D.box_setup("mvn", d, n); s["aux"] = O.mvn_init(d)
tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=1, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=int(sys.argv[4]) if len(sys.argv) > 4 else 1)
print("created", d, flush=True)
tt.run()
oo = O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=1, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=int(sys.argv[4]) if len(sys.argv) > 4 else 1)
print("ok", d, tt.quad(s["quad"]) == oo["value"], tt.neval == oo["neval"], flush=True)
