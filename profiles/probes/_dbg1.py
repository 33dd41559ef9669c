import sys
import numpy as np
sys.path.insert(0, ".")
from ttcross_amd import drivers as D, engine as E
s = D.ising_setup("d", 100, 17)
T = {}
for ar in ("exact", "fast"):
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], 10, pivoting=3, accuracy=s["acc"], quad=s["quad"], arith=ar).run()
    T[ar] = tt.tapes()
for it in range(3):
    df = [p for p in range(1, 99) if not np.array_equal(T["exact"][it][p], T["fast"][it][p])]
    print("sweep", it + 1, "differing bonds", df[:20])
    for p in df[:4]:
        print("  bond", p, T["exact"][it][p].tolist(), T["fast"][it][p].tolist())
print(T["exact"][0][1:99].tolist())
