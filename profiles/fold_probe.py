import ctypes, sys
sys.path.insert(0,'/root/repo')
from ttcross_amd import engine as E
L=E.load_library()
L.ttx_k_fold_probe.argtypes=[ctypes.c_int32,ctypes.c_int32,ctypes.c_int32,ctypes.POINTER(ctypes.c_double)]
o=(ctypes.c_double*7)()
for nblk in (1,1024):
    for ln in (128,250):
        L.ttx_k_fold_probe(0,nblk,ln,o)
        print(f"blocks {nblk:5d} len {ln:4d}: mul-chain {o[0]:.2f} ns/elem  sum-chain {o[1]:.2f} ns/elem  division run {o[2]:.2f} ns/pair | independent x8: fma_f64 {o[3]:.2f} rcp_f64 {o[4]:.2f} rcp_f32 {o[5]:.2f} fdiv_unit {o[6]:.2f} ns/instr")
print(E.k_latency_probe())
