"""pivoting = -1 (full superblock search, lib/dmrgg.f90:341-408) at r = 32, n = 51: the column-by-column path that keeps the
reference's dgemm order (bit-exact) against the dense path (TTX_FULLPIV=mfma: one evaluation pass + fp64 MFMA GEMM fused
with the arg-max).  Run on the GPU box:  python profiles/fullpiv_timing.py [INDEX] > profiles/r02_fullpiv_timing.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ttcross_amd import drivers as D  # noqa: E402
from ttcross_amd import engine as E  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = D.ising_setup("c", m, 51)
res = {}
for mode in ("exact", "mfma"):
    os.environ["TTX_FULLPIV"] = mode
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], 32, pivoting=-1, accuracy=s["acc"], quad=s["quad"], tru=s["tru"])
    tt.run()
    t0 = time.perf_counter()
    tt.run()
    dt = time.perf_counter() - t0
    tt.set_profile(True)
    tt.run()
    ks = tt.kernel_stats()
    res[mode] = (dt, tt.neval, tt.quad(s["quad"]), len(tt.sweeps()) - 1, ks["halfstep"]["ms"])
    print(f"{mode:6s} C_{m} n=51 r=32 piv=-1: {dt*1e3:9.1f} ms per run, {tt.neval} evaluations ({tt.neval/dt/1e9:.2f} G evals/s), {res[mode][3]} sweeps, "
          f"search kernels {ks['halfstep']['ms']:.1f} ms, integral {res[mode][2]:.16e}, rel. err {abs(1 - res[mode][2] / s['tru']):.2e}")
    tt.close()
print(f"dense / exact run time: {res['mfma'][0] / res['exact'][0]:.3f};  integrals differ by {abs(res['mfma'][2] - res['exact'][2]) / abs(res['exact'][2]):.2e} relative")
