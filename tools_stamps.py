import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ttcross_amd import engine as E
E.lib_path = lambda: os.path.join(os.path.dirname(os.path.abspath(__file__)), "ttcross_amd", "lib", "libttx_stamps.so")
from ttcross_amd import drivers as D
g = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = D.ising_setup("c", 64, 51)
tt = E.TTCross(s["n"], s["fun_id"], s["par"], 32, pivoting=2, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=g)
tt.run(); tt.run()
print(tt.seconds)
