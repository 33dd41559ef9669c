"""One run of a bench workload with the -DTTX_STAMPS build (TTX_LIB): phase stamps of k_lottery / k_halfstep go to stderr."""
import sys
sys.path.insert(0, ".")
import bench
from ttcross_amd import drivers as D, engine as E
wl = sys.argv[1]
argv, desc = bench.WORKLOADS[wl]
s = D.ising_setup(argv[1], argv[2], argv[3]) if argv[0] == "ising" else D.box_setup(argv[0], argv[2], argv[3])
groups = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if wl == "mvn128" else 8)
tt = E.TTCross(s["n"], s["fun_id"], s["par"], argv[4], pivoting=argv[5], accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=groups)
tt.run()
print(desc, "arith", tt.arith, "ms", tt.seconds * 1e3, "sweeps", len(tt.sweeps()) - 1, "neval", tt.neval)
