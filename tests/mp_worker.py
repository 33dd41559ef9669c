"""Multi-process worker for the bond-group split over several engine processes (one per GPU in production).
Launched by torch.distributed.run; on a single-GPU box all ranks share device 0 and talk through the
host-callback transport over gloo (RCCL refuses two ranks on one device) -- the engine code path above the
transport is the one the 8-GPU job runs.  Rank 0 checks the job against the oracle with the same number of
bond groups; every rank checks the cores it holds.  Exit code != 0 on any mismatch.

    python -m torch.distributed.run --nproc-per-node W tests/mp_worker.py KIND M N R PIV NGROUPS [gloo|rccl]
    RANK=r WORLD_SIZE=W TTX_SHM_NAME=x python tests/mp_worker.py KIND M N R PIV NGROUPS shm     (one process per rank, no torch)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    # TTX_MP_REPEAT=k: the whole job k times in this process under the SAME transport name (a second dtt_dmrgg call of one job)
    reps = int(os.environ.get("TTX_MP_REPEAT", "1"))
    rc = 0
    for _ in range(reps):
        rc |= one_job()
    sys.exit(rc)


def one_job():
    kind, m, n, r, piv, ng = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    transport = sys.argv[7] if len(sys.argv) > 7 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist = None
    dev = 0
    if transport != "shm":
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        dev = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
        dist.init_process_group("gloo")
    import oracle_lib as O
    from ttcross_amd import drivers as D
    from ttcross_amd import engine as E
    s = D.ising_setup(kind, m, n)
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"],
                   nproc=ng, device=dev, world_rank=rank, world_size=world)
    if transport == "rccl":
        tt.comm_init(dist)
    elif transport == "shm":
        tt.comm_init_shm(os.environ.get("TTX_SHM_NAME", "ttx_test"))
    else:
        tt.set_dist_transport(dist)
    tt.run()
    val = tt.quad(s["quad"])
    oo = O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=ng)
    bad = []
    if not np.array_equal(tt.tapes()[:, 1:tt.d], oo["tapes"][:, 1:tt.d]):
        bad.append("tapes")
    gs, os_ = tt.sweeps(), oo["sweeps"]
    if len(gs) != len(os_):
        bad.append("nsweeps")
    for a, b in zip(gs, os_):
        for f in ("neval", "erank", "val", "amax", "pivotmax"):
            if a[f] != b[f]:
                bad.append(f"sweep{a['it']}.{f}: {a[f]!r} vs {b[f]!r}")
    if tt.neval != oo["neval"]:
        bad.append("neval")
    if not np.array_equal(tt.ranks(), oo["r"]):
        bad.append("ranks")
    if val != oo["value"]:
        bad.append(f"value {val!r} vs {oo['value']!r}")
    ncores = 0
    L = E.load_library()
    for k in range(1, tt.d + 1):
        if L.ttx_core_size(tt._h, k) > 0:
            ncores += 1
            if not np.array_equal(tt.core(k), oo["cores"][k - 1]):
                bad.append(f"core{k}")
    print(f"[rank {rank}/{world}] groups={ng} transport={transport} value={val:.16e} neval={tt.neval} cores_held={ncores} "
          f"path={tt.sweep_path()} time={tt.seconds*1e3:.2f}ms {'OK' if not bad else 'MISMATCH ' + '; '.join(bad[:6])}", flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if os.environ.get("TTX_MP_LEAK") == "1":     # die without closing: the shared-memory segment of this job stays behind under its name
        sys.stdout.flush()
        os._exit(1 if bad else 0)
    tt.close()
    return 1 if bad else 0


if __name__ == "__main__":
    main()
