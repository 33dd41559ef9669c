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
    fast = os.environ.get("TTX_MP_FAST") == "1"       # TTX_ARITH=fast: the checker is the single-process engine in the same mode (same groups)
    s = D.box_setup("mvn", m, n) if kind == "mvn" else D.ising_setup(kind, m, n)
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s.get("aux"),
                   nproc=ng, device=dev, world_rank=rank, world_size=world, arith="fast" if fast else "exact")
    if transport == "rccl":
        tt.comm_init(dist)
    elif transport == "shm":
        tt.comm_init_shm(os.environ.get("TTX_SHM_NAME", "ttx_test"))
    else:
        tt.set_dist_transport(dist)
    tt.run()
    val = tt.quad(s["quad"])
    if fast:
        one = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s.get("aux"), nproc=ng, device=dev,
                        arith="fast").run()
        oo = dict(tapes=one.tapes(), sweeps=one.sweeps(), neval=one.neval, r=one.ranks(), value=one.quad(s["quad"]),
                  cores=[one.core(k) for k in range(1, one.d + 1)])
        one.close()
    else:
        oo = O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=ng, aux=s.get("aux"))
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
    if os.environ.get("TTX_MP_UTILS") == "1":
        # collective post-processing on the multi-process engine against the SAME job as one process (validated against the oracle by
        # the single-process tests): dtt_accchk exact, norm / dot_product 1e-12, ztt_quad 1e-13, dtt_write byte-identical
        import tempfile
        one = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=ng, device=dev).run()
        rng = np.random.default_rng(5)
        wz = rng.standard_normal((3, int(np.sum(s["n"])))) + 1j * rng.standard_normal((3, int(np.sum(s["n"]))))
        a1, a2 = tt.accchk(700), one.accchk(700)
        if any(a1[k] != a2[k] for k in ("einf", "efro", "ainf", "afro")) or not np.array_equal(a1["pivot"], a2["pivot"]):
            bad.append(f"accchk {a1} vs {a2}")
        n1, n2 = tt.norm(), one.norm()
        if abs(n1 - n2) > 1e-12 * abs(n2):
            bad.append(f"norm {n1!r} vs {n2!r}")
        d1, d2 = tt.dot(one), one.dot(one)
        if abs(d1 - d2) > 1e-12 * abs(d2):
            bad.append(f"dot {d1!r} vs {d2!r}")
        z1, z2 = tt.zquad(wz), one.zquad(wz)
        if np.max(np.abs(z1 - z2)) > 1e-13 * np.max(np.abs(z2)):
            bad.append(f"zquad {z1} vs {z2}")
        rep = tt.replicate()
        if not all(np.array_equal(rep.core(k), one.core(k)) for k in range(1, tt.d + 1)):
            bad.append("replica cores")
        rep.close()
        with tempfile.TemporaryDirectory() as td:
            f1, f2 = os.path.join(td, f"a{rank}.tt"), os.path.join(td, f"b{rank}.tt")
            tt.write(f1)
            if rank == 0:
                one.write(f2)
                b1, b2 = open(f1, "rb").read(), open(f2, "rb").read()
                if b1 != b2:
                    k = next((i for i in range(min(len(b1), len(b2))) if b1[i] != b2[i]), -1)
                    bad.append(f"dtt_write file: sizes {len(b1)} / {len(b2)}, first difference at byte {k}")
        one.close()
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
