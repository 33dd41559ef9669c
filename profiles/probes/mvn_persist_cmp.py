"""mvn, TTX_ARITH=fast: persistent incremental tables (default) against tables rebuilt per bond step (TTX_FAST_PERSIST=0), one process per mode."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys; sys.path.insert(0, %r)
from ttcross_amd import drivers as D, engine as E
d, n, r, piv, ng = map(int, sys.argv[1:6])
s = D.box_setup("mvn", d, n)
tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=ng, arith="fast").run()
for a in tt.sweeps()[:8]: print(a["it"], a["neval"], repr(a["val"]), a["erank"])
print("final", len(tt.sweeps()), repr(tt.quad(s["quad"])))
''' % ROOT
for case in [(6, 33, 12, 2, 1), (9, 17, 10, 3, 2), (32, 33, 20, 2, 4), (12, 17, 8, 1, 3)]:
    outs = []
    for pers in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", CODE] + [str(c) for c in case], capture_output=True, text=True, env=dict(os.environ, TTX_FAST_PERSIST=pers))
        outs.append(p.stdout.strip().splitlines() if p.returncode == 0 else ["ERR " + p.stderr[-400:]])
    print("== case", case)
    for a, b in zip(*outs): print("  ", a, "|", b)
