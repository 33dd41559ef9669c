"""Edge probe of the tt_lib entry points on uploaded trains (all cases in one process: no kernel here can fault on bad sizes -- the
host validates; a crash would show as a non-zero exit)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from ttcross_amd import engine as E
rng = np.random.default_rng(5)
def tt(n, r): return [rng.standard_normal((r[k], n[k], r[k + 1])) for k in range(len(n))]
def show(name, f):
    try:
        print("ok     ", name, "->", f(), flush=True)
    except E.TTXError as e:
        print("refused", name, "--", str(e)[:150], flush=True)
    except Exception as e:      # noqa: BLE001
        print("PYERR  ", name, "--", type(e).__name__, str(e)[:150], flush=True)
a = tt([4, 5, 3], [1, 3, 2, 1]); b = tt([4, 5, 3], [1, 2, 2, 1]); c = tt([4, 5], [1, 2, 1]); one = tt([6], [1, 1]); r1 = tt([3, 3, 3, 3], [1, 1, 1, 1, 1])
A, B, C, R1 = (E.TTCross.from_cores(x) for x in (a, b, c, r1))
show("norm of a rank-1 train", lambda: R1.norm())
show("ort of a rank-1 train", lambda: list(E.TTCross.from_cores(r1).ort().ranks()))
show("svd tol=1 (everything may go)", lambda: list(E.TTCross.from_cores(a).svd(1.0, 0).ranks()))
show("svd rmax=1", lambda: list(E.TTCross.from_cores(a).svd(1e-12, 1).ranks()))
show("svd tol=0", lambda: list(E.TTCross.from_cores(a).svd(0.0, 0).ranks()))
show("single-core train", lambda: E.TTCross.from_cores(one).norm())
show("dot of trains with different ranks", lambda: A.dot(B))
show("dot of trains with different shapes", lambda: A.dot(C))
show("tijk inside", lambda: A.tijk([1, 1, 1]))
show("tijk index 0", lambda: A.tijk([0, 1, 1]))
show("tijk index past n", lambda: A.tijk([1, 6, 1]))
show("tijk wrong length", lambda: A.tijk([1, 1]))
show("quad with short weights", lambda: A.quad([np.ones(4), np.ones(4), np.ones(3)]))
show("quad with a missing mode", lambda: A.quad([np.ones(4), np.ones(5)]))
show("core 0 / core d+1", lambda: (A.core(0).shape, A.core(4).shape))
show("run() on an uploaded train", lambda: A.run())
show("accchk on an uploaded train", lambda: A.accchk(10))
big = tt([3, 3], [1, 200, 1])
show("upload with rank 200 (> 128)", lambda: list(E.TTCross.from_cores(big).ranks()))
show("zquad on a small train", lambda: abs(A.zquad([np.ones(4, complex), np.ones(5, complex) * 1j, np.ones(3, complex)])) if hasattr(A, "zquad") else "n/a")
