#!/usr/bin/env python3
"""bench.py -- fiber evaluations per second of the dtt_dmrgg greedy-cross sweep on MI355X.

A "step" is one complete dtt_dmrgg run (initial cross, all sweeps to the reference's stop rule, finalisation)
on the workload BASELINE.json's metric is quoted on: Ising C_64, n=51, maxrank 32, pivoting 2 (63 cores).
`value` = neval / wall(dtt_dmrgg), the reference's own figure of merit (test_crs_ising.f90:146-156), with all
inputs resident in HBM.  For N>1 the SAME problem is split over N GPUs by bond groups (strong scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c64|c16|d32|d256|mvn128|ort|svd...] [--arith exact|fast]

Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (the rook half-step:
fiber evaluation + residual + arg-max), `cpu_baseline` = the genuine reference (oracle/_ref, kind
"reference") or the C oracle (kind "port") timed on the host cores on the same workload.
"""
import argparse
import json
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (driver argv, description)
    "c64": (("ising", "c", 64, 51, 32, 2), "Ising C_64 n=51 r=32 piv=2 (d=63)"),
    "c16": (("ising", "c", 16, 51, 32, 2), "Ising C_16 n=51 r=32 piv=2 (d=15)"),
    "d32": (("ising", "d", 32, 51, 24, 2), "Ising D_32 n=51 r=24 piv=2 (d=31)"),
    # BASELINE config 5 at full size: 0.53 s per step exact (3.7 s with TTX_DE_CUT=0), 0.21 s with --arith fast on one MI355X
    "d256": (("ising", "d", 256, 101, 64, 5), "Ising D_256 n=101 r=64 piv=5 (d=255)"),
    # BASELINE config 4 at full size (4 bond groups by default): 1.2 s per step exact, 0.15 s with --arith fast
    "mvn128": (("mvn", "mvn", 128, 33, 50, 2), "mvn d=128 n=33 r=50 piv=2 (multivariate-normal density, test_crs_mvn)"),
}
# tt_lib utilities on the RESULT train of a sweep: `ort` = dtt_ort (left-to-right Householder QR), `svd` = dtt_svd
# (ort + truncated SVD right-to-left, tol 1e-10); the sweep that produces the train is not timed
UTIL_WORKLOADS = {"ort": "c64", "svd": "c64", "ort_d64": "d64", "svd_d64": "d64"}
WORKLOADS["d64"] = (("ising", "d", 64, 51, 32, 2), "Ising D_64 n=51 r=32 piv=2 (d=63)")
LONG_WORKLOADS = {"d256", "mvn128"}     # one CPU run takes minutes: cpu_baseline times a bounded prefix of one run
FP64_VECTOR_PEAK_TFLOPS = 78.0      # MI355X fp64 vector (non-matrix) peak, SURVEY 8(d)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _time_reference(cmd, env, budget_s, max_runs=15):
    """Repeated full runs of a reference binary for ~budget_s; [(neval, internal seconds)]."""
    runs = []
    t0 = time.time()
    try:
        while time.time() - t0 < budget_s and len(runs) < max_runs:
            out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600).stdout
            mm = re.search(r"\.\.\.with\s+(\d+) evaluations completed in\s+([0-9.E+-]+) sec", out)
            if not mm:
                return []
            runs.append((int(mm.group(1)), float(mm.group(2))))
    except Exception:  # noqa: BLE001
        return []
    return runs


def cpu_baseline(argv, groups, budget_s=24.0):
    """The CPU path timed on this box's host cores on the same workload (a reported baseline, not the target).
    Returns (cpu_baseline, cpu_baseline_single):
      * cpu_baseline        -- the SAME decomposition as the GPU run: the reference under `mpiexec -np <groups>`, one
                               OpenMP thread per rank (oracle/_ref/test_crs_ising_mpi: the build with the right-going
                               boundary exchange re-inserted -- the unpatched fp64 source aborts on more than one rank);
                               falls back to the single-process figure when mpiexec or the binary is missing;
      * cpu_baseline_single -- the unmodified reference as one process, best of {8, 1} OpenMP threads.
    Without oracle/_ref: the C oracle (kind "port", one thread)."""
    kind, m, n, r, piv = argv[1].upper(), argv[2], argv[3], argv[4], argv[5]
    cores = os.cpu_count() or 1
    args = [kind, str(m), str(n), str(r), str(piv)]
    exe = os.path.join(ROOT, "oracle", "_ref", "test_crs_ising")
    exe_mpi = os.path.join(ROOT, "oracle", "_ref", "test_crs_ising_mpi")
    mpiexec = "/opt/conda/bin/mpiexec"
    single = multi = None
    if os.path.exists(exe):
        # the reference's OpenMP regions are tiny (a fiber of <= r*n evaluations): more threads than ~8 only add
        # fork/join cost, so the best of {8, 1} threads is reported (thread count stated in `cores`)
        best = None
        for thr in sorted({min(cores, 8), 1}, reverse=True):
            env = dict(os.environ, OMP_NUM_THREADS=str(thr), MKL_THREADING_LAYER="SEQUENTIAL", OMP_PROC_BIND="close")
            runs = _time_reference([exe] + args, env, budget_s / 4)
            if runs:
                rate = statistics.median([a / b for a, b in runs])
                if best is None or rate > best[0]:
                    best = (rate, thr, len(runs), statistics.median([b for _, b in runs]))
        if best:
            single = {"value": best[0], "unit": "evals/s", "cores": best[1], "kind": "reference",
                      "sample": f"{best[2]} full runs of test_crs_ising {' '.join(args)} (genuine reference, unmodified sources, amdflang -O2 -fopenmp + "
                                f"MKL sequential, 1 process x OMP_NUM_THREADS={best[1]} of {cores} host cores); median neval/internal time; median time {best[3]:.4f} s"}
    if os.path.exists(exe_mpi) and os.path.exists(mpiexec) and groups > 1:
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_THREADING_LAYER="SEQUENTIAL")
        runs = _time_reference([mpiexec, "-np", str(groups), exe_mpi] + args, env, budget_s / 2)
        if runs:
            multi = {"value": statistics.median([a / b for a, b in runs]), "unit": "evals/s", "cores": groups, "kind": "reference",
                     "sample": f"{len(runs)} full runs of mpiexec -np {groups} test_crs_ising {' '.join(args)} (reference with the right-going boundary exchange "
                               f"re-inserted from lib/dmrggmp.f90:572-629, 1 OpenMP thread per rank, {cores} host cores): the decomposition of the GPU run; "
                               f"median neval/internal time; median time {statistics.median([b for _, b in runs]):.4f} s"}
    if single is None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        from ttcross_amd import drivers as D
        s = D.ising_setup(argv[1], m, n)
        runs = []
        t0 = time.time()
        while time.time() - t0 < budget_s / 2 and len(runs) < 25:
            o = O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"])
            runs.append((o["neval"], o["seconds"]))
        single = {"value": statistics.median([a / b for a, b in runs]), "unit": "evals/s", "cores": 1, "kind": "port",
                  "sample": f"{len(runs)} full runs of the C oracle (oracle/ttx_oracle.c, 1 thread) on the same workload"}
    return (multi or single), single


def cpu_baseline_prefix(argv, limit_s=None):
    """Workloads whose CPU run takes minutes (D_256: 474 s, mvn 128: 51 s on 8 cores): the genuine reference is started on all
    host cores of this box and stopped after limit_s; the rate is n_evals / time of the last per-sweep line it printed
    (lib/dmrgg.f90:971-1008).  The cost per evaluation does not depend on the sweep, so the prefix rate stands for the run."""
    import signal
    cores = min(os.cpu_count() or 1, 64)
    if limit_s is None:      # mvn: the whole run fits (its low-rank sweeps are fork/join-bound, a short prefix would flatter the GPU)
        limit_s = 30.0 if argv[0] == "ising" else 100.0
    if argv[0] == "ising":
        cmd = [os.path.join(ROOT, "oracle", "_ref", "test_crs_ising"), argv[1].upper()] + [str(x) for x in argv[2:]]
    else:
        cmd = [os.path.join(ROOT, "oracle", "_ref", "test_crs_mvn")] + [str(x) for x in argv[2:]]
    if not os.path.exists(cmd[0]):
        return None
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_THREADING_LAYER="SEQUENTIAL", OMP_PROC_BIND="close")
    last = None
    try:
        import select
        # the Fortran run time flushes per line only on a terminal; the GPU boxes have no pty devices, so a preloaded
        # isatty() shim (oracle/tty_shim.c, bench infrastructure) makes it treat the pipe like one
        shim = os.path.join(ROOT, "oracle", "tty_shim.so")
        if not os.path.exists(shim):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "tty_shim.so"], check=False)
        if os.path.exists(shim):
            env["LD_PRELOAD"] = shim
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        fd = p.stdout.fileno()
        t0 = time.time()
        buf = ""
        while time.time() - t0 < limit_s:
            if select.select([fd], [], [], 0.5)[0]:
                chunk = os.read(fd, 65536)
                if not chunk:
                    break
                buf += chunk.decode(errors="replace")
                *lines, buf = buf.split("\n")
                for line in lines:
                    mm = re.search(r"time:\s*([0-9.E+-]+)\s+n_evals:\s*(\d+)", line)
                    if mm and float(mm.group(1)) > 0:
                        last = (int(mm.group(2)), float(mm.group(1)), line.split()[0])
            elif p.poll() is not None:
                break
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
            try:
                p.wait(timeout=10)
            except Exception:  # noqa: BLE001
                p.kill()
    except Exception as e:  # noqa: BLE001
        print(f"cpu_baseline_prefix: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    if not last:
        print("cpu_baseline_prefix: the reference printed no per-sweep line within the limit", file=sys.stderr)
        return None
    return {"value": last[0] / last[1], "unit": "evals/s", "cores": cores, "kind": "reference",
            "sample": f"the first {last[1]:.1f} s of {' '.join(os.path.basename(c) if i == 0 else c for i, c in enumerate(cmd))} (genuine reference, amdflang -O2 -fopenmp + MKL sequential, "
                      f"OMP_NUM_THREADS={cores}): {last[0]} evaluations up to sweep {last[2]}"}


def bench_utility(a, D, E):
    """`--workload ort|svd[_d64]`: dtt_ort / dtt_svd (lib/tt.f90:130-198, 307-368) on the finalised train of a sweep, one GPU.
    A step = one call on a fresh copy of that train (the upload of the copy is not timed).  Reported against the HBM
    roofline with the algorithmic traffic of one pass over the cores per stage (ort: read + write every core once;
    svd: ort plus a second read + write), and the Householder flop count against the fp64 MATRIX peak for reference."""
    import numpy as np
    base = UTIL_WORKLOADS[a.workload]
    argv, desc = WORKLOADS[base]
    s = D.ising_setup(argv[1], argv[2], argv[3])
    src = E.TTCross(s["n"], s["fun_id"], s["par"], argv[4], pivoting=argv[5], accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=8).run()
    cores = [src.core(k) for k in range(1, src.d + 1)]
    r = src.ranks()
    op = a.workload.split("_")[0]
    times = []
    out_ranks = None
    for it in range(a.warmup + a.steps):
        t = E.TTCross.from_cores(cores)
        t.tijk([1] * src.d)                 # first utility call allocates the engine's scratch buffers: not part of the step
        t0 = time.perf_counter()
        if op == "ort":
            t.ort()
        else:
            t.svd(1e-10)
        dt = time.perf_counter() - t0
        if it >= a.warmup:
            times.append(dt)
        out_ranks = t.ranks()
        nrm = t.norm()
        t.close()
    sz = sum(c.size for c in cores)
    passes = 2 if op == "ort" else 4
    ms = 1e3 * sum(times) / len(times)
    flops = sum(2.0 * (r[k] * s["n"][k]) * r[k + 1] ** 2 * 2 for k in range(src.d))      # QR + forming Q, per core
    line = {"metric": f"tt_lib {('dtt_ort' if op == 'ort' else 'dtt_svd')} on the result train of {desc}", "value": src.d / (ms * 1e-3), "unit": "cores/s",
            "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "the finalised cores of a dtt_dmrgg run (synthetic integrand)",
            "config": {"workload": a.workload, "cores": src.d, "doubles": int(sz), "ranks_in_max": int(max(r)), "ranks_out_max": int(max(out_ranks)), "norm": nrm},
            "roofline": {"kernel": "k_qr (+ k_gemm_mfma, k_jacobi_svd)", "bound": "hbm", "achieved": passes * 8.0 * sz / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": passes * 8.0 * sz / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_step": passes * 8.0 * sz, "householder_flops_per_step": flops,
                         "note": "63 dependent per-core stages of < 0.5 MB each: launch- and latency-bound, see DESIGN.md section 8"}}
    if not a.no_cpu_baseline:
        # cpu_baseline: the GENUINE reference's dtt_ort / dtt_svd (LAPACK through MKL sequential, as the reference links it) on the
        # IDENTICAL train -- written here with ttx_write in the reference's stream format, read there with its own dtt_read --
        # by oracle/_ref/ref_tt_timing (our own small timing driver around the reference's modules; bench infrastructure)
        exe = os.path.join(ROOT, "oracle", "_ref", "ref_tt_timing")
        if os.path.exists(exe):
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                fn = os.path.join(td, "train.tt")
                t = E.TTCross.from_cores(cores)
                t.write(fn)
                t.close()
                env = dict(os.environ, OMP_NUM_THREADS="1", MKL_THREADING_LAYER="SEQUENTIAL")
                try:
                    o = subprocess.run([exe, fn, op, "1e-10", "12"], capture_output=True, text=True, env=env, timeout=300).stdout
                    mm = re.search(r"calls\s+(\d+)\s+median_ms\s+([0-9.]+)\s+ranks_max\s+(\d+)", o)
                    if mm:
                        cms = float(mm.group(2))
                        line["cpu_baseline"] = {"value": src.d / (cms * 1e-3), "unit": "cores/s", "cores": 1, "kind": "reference", "ms_per_step": cms,
                                                "ranks_out_max": int(mm.group(3)),
                                                "sample": f"{mm.group(1)} calls of the genuine reference's {'dtt_ort' if op == 'ort' else 'dtt_svd(tol=1e-10)'} (lib/tt.f90, LAPACK via MKL sequential, "
                                                          "1 thread) on the identical train, read with its own dtt_read; median time per call"}
                        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
                except Exception as e:  # noqa: BLE001
                    print(f"cpu_baseline (tt_lib): {type(e).__name__}: {e}", file=sys.stderr)
    print(json.dumps(line))


def tt_own(nproc, d):
    """share(1, d-1, nproc) of lib/default.f90:78-97: own(0:nproc)."""
    first, last = 1, d - 1
    return [first + int(float(last - first + 1) * p / nproc) for p in range(nproc)] + [last + 1]


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def supervise(a, rank, world):
    """N > 1: every rank started by torch.distributed.run becomes a SUPERVISOR that never touches the GPU and runs the
    measurement in a child process (`--child`), one attempt per transport: first the in-library RCCL transport, then the
    host-staged gloo transport (the one the multi-process tests cover).  An attempt whose child crashes or does not finish
    within --attempt-timeout on ANY rank is stopped on all ranks (the supervisors talk over their own gloo group / store)
    and the next transport is tried, so that a fault in the RCCL path costs one time-out, not the bench line.
    The child does the barrier + synchronize bracketing and the max over ranks itself; rank 0's JSON line is relayed."""
    import signal
    import torch.distributed as dist
    dist.init_process_group("gloo")
    store = dist.distributed_c10d._get_default_store()
    attempts = ["gloo"] if a.backend == "gloo" else ["nccl", "gloo"]
    log = []
    for k, backend in enumerate(attempts):
        port = [_free_port() if rank == 0 else 0]
        dist.broadcast_object_list(port, src=0)
        # the children rendezvous among themselves on a fresh port; without the launcher's TORCHELASTIC_* variables
        # (TORCHELASTIC_USE_AGENT_STORE would make them look for the launcher's store on that port)
        env = {k_: v for k_, v in os.environ.items() if not k_.startswith("TORCHELASTIC_")}
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]))
        argv = [x for x in sys.argv[1:]]
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--child", "--backend", backend, "--attempt", str(k)]
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
        key = f"bench_attempt_{k}_failed"
        t0 = time.time()
        why = None
        while child.poll() is None:
            time.sleep(0.5)
            if time.time() - t0 > a.attempt_timeout:
                why = f"no result within {a.attempt_timeout:.0f} s"
            elif store.check([key]):
                why = "stopped: the attempt failed on another rank"
            if why:
                try:
                    os.killpg(child.pid, signal.SIGKILL)        # exactly the process group started above
                except ProcessLookupError:
                    pass
                child.wait()
                break
        out = child.stdout.read() if child.stdout else ""
        line = None
        for ln in out.splitlines():
            if ln.startswith("{"):
                line = ln
        bad = why is not None or child.returncode != 0 or (rank == 0 and line is None)
        if bad:
            store.set(key, "1")
            if why is None:
                why = f"child exit code {child.returncode}" if child.returncode != 0 else "child printed no result line"
            print(f"[bench supervisor, rank {rank}] attempt {k} ({backend}): {why}", file=sys.stderr, flush=True)
        flag = [1 if bad else 0]
        allflags = [None] * world
        dist.all_gather_object(allflags, flag)
        failed = [i for i, f in enumerate(allflags) if f[0]]
        log.append({"backend": backend, "ok": not failed, "failed_ranks": failed, "seconds": round(time.time() - t0, 1)})
        if not failed:
            if rank == 0:
                res = json.loads(line)
                res["attempts"] = log
                print(json.dumps(res), flush=True)
            dist.destroy_process_group()
            return 0
    dist.destroy_process_group()
    if rank == 0:
        print(f"[bench supervisor] every transport failed: {log}", file=sys.stderr, flush=True)
    return 1


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher: this process, which has not touched the GPU and never will, starts
    the N ranks itself as FRESH child processes (never a re-exec) with the environment torch.distributed.run would give them
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), relays rank 0's result line and returns the
    worst exit code.  Each rank then supervises its measurement as under the launcher (supervise())."""
    import signal
    port = _free_port()
    procs = []
    for rk in range(a.gpus):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rk == 0 else subprocess.DEVNULL, text=True, start_new_session=True))
    deadline = time.time() + 2 * a.attempt_timeout + 120
    rc = 0
    out0 = ""
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        for p in procs[1:]:
            p.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        rc = 1
        print("[bench] the ranks did not finish in time; stopping them", file=sys.stderr, flush=True)
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)            # exactly the process groups started above
            except ProcessLookupError:
                pass
            p.wait()
        rc = max(rc, abs(p.returncode or 0))
    for ln in out0.splitlines():
        if ln.startswith("{"):
            print(ln, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c64", choices=sorted(WORKLOADS) + sorted(UTIL_WORKLOADS))
    ap.add_argument("--no-extras", action="store_true", help="skip the k2_streaming and single_group side measurements (profiler runs: only the workload's own launches)")
    ap.add_argument("--groups", type=int, default=0, help="bond groups = MPI ranks of the reference's domain split; default 8 (config 3 of BASELINE.json) at every N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arith", default="exact", choices=["exact", "fast"], help="TTX_ARITH: exact (default; every fp64 operation of the integrand in the reference's order) "
                    "or fast (Ising D/E, mvn: re-associated O(d) evaluation, tolerance-checked)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N>1 (gloo: rehearsal with several ranks on one GPU)")
    ap.add_argument("--attempt-timeout", type=float, default=420.0, help="N>1: seconds one transport attempt may take before the supervisors stop it")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)        # set by supervise(): this process does the measurement
    ap.add_argument("--attempt", type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world == 1 and a.gpus > 1 and a.workload not in UTIL_WORKLOADS:
        raise SystemExit(self_launch(a))       # no launcher: start the ranks here (fresh children; this process stays off the GPU)
    if world > 1 and not a.child and a.workload not in UTIL_WORKLOADS:
        raise SystemExit(supervise(a, rank, world))
    if a.child and os.environ.get("TTX_BENCH_TEST_FAULT"):       # tests/test_dist_cpu.py: "<attempt>:<rank>:hang|crash|dry"
        for spec in os.environ["TTX_BENCH_TEST_FAULT"].split(","):
            att, rk, what = spec.split(":")
            if (att == "*" or int(att) == a.attempt) and (rk == "*" or int(rk) == rank):
                if what == "hang":
                    time.sleep(3600)
                if what == "crash":
                    raise SystemExit(7)
                if what == "dry":       # no engine: checks the supervisors' protocol only
                    import torch.distributed as dist
                    dist.init_process_group("gloo")
                    dist.barrier()
                    if rank == 0:
                        print(json.dumps({"metric": "dry", "value": 1.0, "n_gpus": world, "backend": a.backend}))
                    dist.destroy_process_group()
                    return
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if a.backend == "gloo":
            local = local % max(torch.cuda.device_count(), 1)
            dist.init_process_group("gloo")
            gloo_pg = None
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            try:        # only needed if the in-library RCCL transport cannot be brought up
                gloo_pg = dist.new_group(backend="gloo")
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] no gloo group for the fallback transport: {e}", file=sys.stderr, flush=True)
                gloo_pg = None

    from ttcross_amd import drivers as D
    from ttcross_amd import engine as E

    if a.workload in UTIL_WORKLOADS:
        return bench_utility(a, D, E)
    argv, desc = WORKLOADS[a.workload]
    s = D.ising_setup(argv[1], argv[2], argv[3]) if argv[0] == "ising" else D.box_setup(argv[0], argv[2], argv[3])
    groups = a.groups or max(4 if a.workload == "mvn128" else 8, world)
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], argv[4], pivoting=argv[5], accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"],
                   nproc=groups, device=local, world_rank=rank, world_size=world, arith=a.arith) if world > 1 else \
        E.TTCross(s["n"], s["fun_id"], s["par"], argv[4], pivoting=argv[5], accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"],
                  nproc=groups, device=local, arith=a.arith)
    transport = "none (single process)"
    if world > 1:
        # data path between GPUs: in-library RCCL point-to-point / all-reduce on the engine's stream; if RCCL cannot
        # be initialised on ANY rank, all ranks fall back together to the host-staged gloo transport
        import torch
        ok = 1
        try:
            if a.backend == "gloo":
                raise RuntimeError("gloo rehearsal: RCCL not attempted")
            tt.comm_init(dist)
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] RCCL transport unavailable: {e}", file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            transport = "rccl (ncclSend/ncclRecv + ncclAllReduce over xGMI)"
        else:
            tt.close()
            tt = E.TTCross(s["n"], s["fun_id"], s["par"], argv[4], pivoting=argv[5], accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"],
                           nproc=groups, device=local, world_rank=rank, world_size=world, arith=a.arith)
            if a.backend == "nccl" and gloo_pg is None:
                raise SystemExit("neither the RCCL transport nor a gloo fallback group is available")
            tt.set_dist_transport(dist, group=gloo_pg)
            transport = "gloo host-staged fallback"

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        tt.run()
    barrier()
    t0 = time.perf_counter()
    neval = 0
    for _ in range(a.steps):
        tt.run()            # ends with a stream synchronisation: all device work of the step is complete
        neval += tt.neval
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = tt.quad(s["quad"])
    nsweeps = len(tt.sweeps()) - 1
    # N > 1: the result depends only on the number of bond groups, not on how they are spread over GPUs -- rank 0 repeats
    # the job as ONE process and the integral must come out bit-identical (the transport between GPUs moved the right bytes)
    multi_check = None
    if world > 1 and rank == 0:
        try:
            t1 = E.TTCross(s["n"], s["fun_id"], s["par"], argv[4], pivoting=argv[5], accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"],
                           nproc=groups, device=local, arith=a.arith)
            t1.run()
            v1 = t1.quad(s["quad"])
            multi_check = "integral identical to the single-process run of the same bond groups" if (v1 == value and t1.neval == tt.neval) else \
                          f"DIFFERS from the single-process run: {value!r} / {tt.neval} vs {v1!r} / {t1.neval}"
            t1.close()
        except Exception as e:  # noqa: BLE001
            multi_check = f"single-process check failed: {e}"

    # roofline pass: the same step with every launch bracketed by HIP events on the engine's stream
    tt.set_profile(True)
    agg = {}
    for _ in range(max(1, min(a.steps, 3))):
        tt.run()
        for k, v in tt.kernel_stats().items():
            g = agg.setdefault(k, dict(launches=0, ms=0.0, bytes=0.0))
            for f in g:
                g[f] += v[f]
    tt.set_profile(False)
    hs = agg["halfstep"]
    avg_us = 1e3 * hs["ms"] / max(hs["launches"], 1)
    bytes_per_launch = hs["bytes"] / max(hs["launches"], 1)
    achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0

    # K2 in isolation on an HBM-resident factor (the same residual + arg-max code): where bandwidth, not launch
    # latency, is the limit.  And the reference's 1-rank decomposition (1 bond group) for comparison.
    k2 = None
    one_group = None
    if rank == 0 and world == 1 and not a.no_extras:
        m_rows, r_cols = 1 << 22, 32
        ms, by = E.k_residual_bench(m_rows, r_cols, 20, device=local)
        k2 = {"kernel": "k_resid_argmax_stream (K2 residual + arg-max, same arithmetic)", "rows": m_rows, "rank": r_cols,
              "factor_bytes": 8 * m_rows * r_cols, "avg_launch_us": 1e3 * ms, "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS}
        if groups != 1:
            t1 = E.TTCross(s["n"], s["fun_id"], s["par"], argv[4], pivoting=argv[5], accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"],
                           nproc=1, device=local, arith=a.arith)
            t1.run()
            tb = time.perf_counter()
            for _ in range(3):
                t1.run()
            d1 = (time.perf_counter() - tb) / 3
            one_group = {"bond_groups": 1, "ms_per_step": 1e3 * d1, "value": t1.neval / d1, "neval_per_step": t1.neval,
                         "integral": t1.quad(s["quad"])}
            t1.close()
    if rank != 0:
        return
    path = tt.sweep_path()
    arith = tt.arith
    heavy = (argv[0] == "ising" and argv[1] in ("d", "e")) or argv[0] == "mvn"
    # the chain path's half-step kernel by integrand and arithmetic: the wave-per-pivot kernels of Ising D/E and mvn (exact), the generic
    # kernel with the table evaluators of ttx_fast.h (fast), the generic kernel otherwise
    de_cut = argv[0] == "ising" and argv[1] in ("d", "e") and os.environ.get("TTX_DE_CUT", "1") != "0"       # nodes of the drivers lie in [0,1]
    chain_k = "k_halfstep" if (arith == "fast" and heavy) else ("k_halfstep_dec" if de_cut else "k_halfstep_de / k_halfstep_det") if (argv[0] == "ising" and argv[1] in ("d", "e")) else \
              "k_halfstep_mvn" if argv[0] == "mvn" else "k_halfstep"
    kname = {"chain": chain_k, "fused": "k_sweep_fused", "cluster": "k_sweep_cluster"}[path]
    kdesc = {"chain": chain_k + (" with the table evaluators of TTX_ARITH=fast (one rook half-step: O(d) fiber evaluation + residual K2 + arg-max)" if (arith == "fast" and heavy)
                                 else " (one rook half-step: fiber evaluation + residual + arg-max)"),
             "fused": "k_sweep_fused (whole sweep of a bond group in one workgroup; bytes = its rook half-steps)",
             "cluster": "k_sweep_cluster (whole sweep of a bond group by a cluster of workgroups; bytes = its rook half-steps)"}[path]
    # HBM traffic of that kernel per launch: FETCH_SIZE + WRITE_SIZE from the committed rocprofv3 --pmc passes of THIS command and THIS
    # round's kernels (profiles/measure_r03.sh writes profiles/r03_pmc_fetch_write_<workload>_<arith>_g<groups>.csv); the file's git blob id
    # is recorded so that the number can be traced to the committed measurement.  Absent file: null.
    traffic = None
    traffic_source = None
    try:
        import csv
        pm = os.path.join(ROOT, "profiles", f"r03_pmc_fetch_write_{a.workload}_{arith}_g{groups}.csv")
        if os.path.exists(pm) and world == 1:
            f = w = None
            first = kname.split(" ")[0]
            for row in csv.reader(l for l in open(pm) if not l.startswith("#")):
                nm = row[1].replace("void ", "") if len(row) >= 6 else ""
                if nm == first or nm.startswith(first + "<") or nm.startswith(first + "("):
                    if row[2] == "FETCH_SIZE" and f is None:
                        f = float(row[5])
                    if row[2] == "WRITE_SIZE" and w is None:
                        w = float(row[5])
            if f is not None and w is not None:
                # gfx950: FETCH_SIZE tallies a 128-byte request as 64 bytes (MI355X_MICROARCH.md); calibrated on THIS access pattern --
                # 8-byte lanes sweeping factor slabs, the K2 streaming kernel on a known byte count
                # (profiles/r03_pmc_fetch_write_calib_k2_stream.csv: 553.7 MB reported for 1107.3 MB read, WRITE_SIZE exact) -> 2 x FETCH + WRITE
                traffic = (2.0 * f + w) * 1024.0
                blob = subprocess.run(["git", "hash-object", pm], capture_output=True, text=True, cwd=ROOT).stdout.strip()
                traffic_source = (f"{os.path.relpath(pm, ROOT)} (git blob {blob[:12] or 'n/a'}): 2 x mean FETCH_SIZE + mean WRITE_SIZE per launch of {first} (gfx950 counts a "
                                  "128-byte read request as 64 bytes; factor calibrated on the K2 streaming kernel, profiles/r03_calib_k2_stream.txt), separate rocprofv3 --pmc "
                                  "passes of this command with this round's kernels (not collected in this run)")
    except Exception:  # noqa: BLE001
        traffic = None
    hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    out = {
        "metric": "fiber evals/s (neval / wall time of dtt_dmrgg), " + desc,
        "value": neval / dt, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic (integrand evaluated on the fly; Gauss-Legendre nodes/weights; flang-compatible lottery RNG stream)",
        "multi_gpu_check": multi_check,
        "config": {"workload": desc, "driver": ("test_crs_ising " + " ".join(str(x) for x in argv[1:])) if argv[0] == "ising" else ("test_crs_mvn " + " ".join(str(x) for x in argv[2:])), "bond_groups": groups, "transport": transport,
                   "arith": arith, "sweep_path": path,
                   "neval_per_step": neval // a.steps, "sweeps": nsweeps, "integral": value,
                   "rel_err_vs_analytic": abs(1 - value / s["tru"]) if s["tru"] else None},
        # `bound` names what limits the dominant kernel on THIS path: "hbm" where the half-step is its residual sweep K2 (fast mode of
        # the heavy integrands, generic chain), "fp64-valu" where it is the integrand's dependent fp64 chain (exact Ising D/E, mvn),
        # "latency" for the whole-sweep kernels (a launch moves ~4 MB in ~300 us: achieved / peak are then the measured launch time and
        # the modelled floor of its dependent chain, filled in below; the HBM view stays in `hbm`)
        "roofline": dict({"kernel": kdesc, "bound": "hbm"}, **hbm, traffic=traffic, traffic_source=traffic_source,
                         avg_launch_us=avg_us, launches_per_step=hs["launches"] / max(1, min(a.steps, 3)),
                         algorithmic_bytes_per_launch=bytes_per_launch),
        "kernel_ms_per_step": {k: v["ms"] / max(1, min(a.steps, 3)) for k, v in agg.items()},
    }
    # K1 (fiber evaluation) as fp64 vector work, SURVEY 8(d): algorithmic flops per evaluation of the integrand
    dd = len(s["n"])
    fl = (5 * dd + 1) if argv[1] == "c" else (2 * dd * dd + dd + 20) if argv[0] == "mvn" else (6 * dd * (dd + 1) // 2 + 5 * dd)
    runs_p = max(1, min(a.steps, 3))
    k1_ms = (agg["halfstep"]["ms"] + agg["lottery"]["ms"]) / runs_p            # the kernels that evaluate (profile pass)
    if arith == "fast" and heavy:
        out["k1_evaluation"] = {"flops_per_eval_reference_formula": fl, "note": "TTX_ARITH=fast does not execute the reference's O(d^2) formula: per fiber element it multiplies the "
                                "~18 x 18 (after the 2^-54 cut, triangular) range factors through the free dimension, everything else comes from per-pivot tables (ttx_fast.h); "
                                "the evaluation is a few microseconds per half-step and the half-step is bound by its residual sweep (roofline.bound = hbm)"}
    else:
        out["k1_evaluation"] = {"flops_per_eval": fl, "achieved": (neval / dt) * fl / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": (neval / dt) * fl / 1e12 / FP64_VECTOR_PEAK_TFLOPS}
        if heavy and path == "chain" and k1_ms > 0:       # exact Ising D/E / mvn: the half-step IS the integrand's fp64 chain
            ach = (neval / a.steps) * fl / (k1_ms * 1e-3) / 1e12
            out["roofline"].update({"bound": "fp64-valu", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS,
                                    "hbm": hbm, "note": "algorithmic flops of the reference's formula per evaluation x evaluations per step / time of the evaluating kernels "
                                                        "(lottery + half-steps, HIP events); every product / sum kept in the reference's order"})
            if de_cut:
                out["roofline"]["note"] += ("; with all nodes in [0,1] the rows of the pair triangle end at the unit cut (running product <= 2^-54: the factor is "
                                            "exactly 1): the kernels EXECUTE ~12 % of the formula's factors at D_256 (instrumented oracle) -- `achieved` counts the "
                                            "formula's flops, i.e. the work the reference performs for the same bits; executed_frac_of_formula is the measured share")
                out["roofline"]["executed_frac_of_formula"] = 0.12 if argv[2] >= 128 else None
    # The HBM roofline is the wrong yardstick for the sweep kernel at BASELINE sizes (a launch moves 3.8 MB in ~300 us): what
    # bounds it is a chain of DEPENDENT operations.  Latency model of one launch, from unit costs measured in this run on
    # one wave (ttx_k_latency_probe) and the operation counts of the algorithm:
    #   bond steps in sequence per launch = max own bonds of a group;  per bond step: 1 lottery + H half-steps + 1 append.
    #   one evaluation of the Ising C integrand with the prefix-state scheme: ~(A+2)+(B+2) running-sum steps + m weight
    #   multiplies + 1 division on the critical path; a residual adds r dependent multiply-adds; every half-step needs the
    #   factor row from L2 and the partners' arg-max records back through L2 (2 dependent round trips), a lottery 2 more
    #   (pivot lists, candidate factor rows), an append 2 (neighbour LU, barrier).
    if rank == 0 and world == 1 and argv[1] == "c":
        try:
            lat = E.k_latency_probe(local)
            dd = len(s["n"])
            nb_seq = max(b - a_ for a_, b in zip(tt_own(groups, dd)[:-1], tt_own(groups, dd)[1:]))
            runs_ = max(1, min(a.steps, 3))
            launches = hs["launches"] / runs_
            hsteps = (tt.resid_halfsteps / groups) / max(launches, 1)        # residual half-steps per group and launch
            rbar = argv[4] / 2.0
            t_eval = (dd + 4) * lat["fp64_mul_add_ns"] + dd * lat["fp64_mul_ns"] + lat["fp64_div_ns"]
            t_half = t_eval + rbar * lat["fp64_mul_add_ns"] + 2 * lat["l2_roundtrip_ns"] + 12 * lat["lds_read_ns"]
            t_lot = t_eval + rbar * lat["fp64_mul_add_ns"] + 2 * lat["l2_roundtrip_ns"] + 12 * lat["lds_read_ns"]
            t_app = rbar * lat["fp64_mul_add_ns"] + 2 * lat["l2_roundtrip_ns"]
            per_bond = hsteps / nb_seq
            model_us = 1e-3 * nb_seq * (t_lot + (per_bond + 1) * t_half + t_app)
            out["latency_model"] = {
                "kernel": kname, "unit_latencies_ns": lat, "bond_steps_in_sequence_per_launch": nb_seq,
                "residual_halfsteps_per_bond_step": per_bond, "dependent_ops_per_evaluation": 2 * dd + 5,
                "modelled_floor_us_per_launch": model_us, "measured_us_per_launch": avg_us, "frac_of_floor": model_us / avg_us if avg_us else None,
                "note": "floor = bond steps x (lottery + half-steps + append), each = dependent fp64 chain of one evaluation + residual + "
                        "2 L2 round trips + 12 LDS reads at the unit latencies measured on this device in this run"}
            if path in ("cluster", "fused") and avg_us > 0:
                out["roofline"].update({"bound": "latency", "achieved": avg_us, "peak": model_us, "unit": "us per launch (peak = modelled floor of the dependent chain; lower is better)",
                                        "frac": model_us / avg_us, "hbm": hbm})
        except Exception as e:  # noqa: BLE001
            out["latency_model"] = {"error": str(e)}
    if k2:
        k2["note"] = "MICROBENCHMARK, not on the product path: the same residual + arg-max code on a synthetic 1 GB factor (no BASELINE config has a factor above 3.3 MB)"
        out["k2_streaming"] = k2
    if one_group:
        out["single_group"] = one_group
    if not a.no_cpu_baseline and world == 1:
        if a.workload in LONG_WORKLOADS:
            out["cpu_baseline"] = cpu_baseline_prefix(argv)
        else:
            out["cpu_baseline"], out["cpu_baseline_single"] = cpu_baseline(argv, groups)
        if out["cpu_baseline"]:
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
