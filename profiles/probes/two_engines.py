"""Two engines on one GPU driven from two host threads at the same time (the cluster kernels of both must be co-resident or fall
back): results must equal the single-engine run."""
import sys, threading, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from ttcross_amd import drivers as D, engine as E
s = D.ising_setup("c", 64, 51)
mk = lambda: E.TTCross(s["n"], s["fun_id"], s["par"], 32, pivoting=2, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=8)
ref = mk().run(); vref, nref = ref.quad(s["quad"]), ref.neval
res = {}
def work(k):
    tt = mk()
    vals = set()
    for _ in range(30):
        tt.run(); vals.add((tt.quad(s["quad"]), tt.neval))
    res[k] = (vals, tt.cluster_fallbacks, tt.sweep_path())
t0 = time.time()
ths = [threading.Thread(target=work, args=(k,)) for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
[t.start() for t in ths]; [t.join() for t in ths]
print("seconds", round(time.time() - t0, 2))
for k, (vals, fb, path) in res.items():
    print("engine", k, "identical to the single run:", vals == {(vref, nref)}, "fallbacks", fb, "path", path)
