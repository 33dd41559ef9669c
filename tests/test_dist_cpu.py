"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the host transport layer the engine uses between GPUs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,nproc", [(2, 8), (2, 3), (3, 8)])
def test_gloo_transport_and_group_split(world, nproc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world * 10 + nproc), os.path.join(ROOT, "tests", "dist_cpu_worker.py"), str(nproc)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count(" OK") == world


def test_group_split_covers_all_groups():
    from ttcross_amd import engine as E
    for nproc in range(1, 20):
        for world in range(1, nproc + 1):
            seen = []
            for r in range(world):
                g0, G = E.split_groups(nproc, r, world)
                assert G >= 1
                seen += list(range(g0, g0 + G))
            assert seen == list(range(nproc))


def _run_bench_supervisors(fault, extra=()):
    """bench.py --gpus 2 as the driver launches it; the children are stand-ins (no GPU here): protocol only."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TTX_BENCH_TEST_FAULT=fault)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29733", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


def test_bench_supervisor_relays_the_first_good_attempt():
    out, lines = _run_bench_supervisors("*:*:dry")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["backend"] == "nccl" and lines[0]["attempts"] == [
        {"backend": "nccl", "ok": True, "failed_ranks": [], "seconds": lines[0]["attempts"][0]["seconds"]}]


@pytest.mark.parametrize("fault", ["hang", "crash"])
def test_bench_supervisor_falls_back_when_a_rank_of_the_first_transport_fails(fault):
    out, lines = _run_bench_supervisors(f"0:1:{fault},*:*:dry", ("--attempt-timeout", "15"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["backend"] == "gloo"
    att = lines[0]["attempts"]
    assert [x["backend"] for x in att] == ["nccl", "gloo"] and not att[0]["ok"] and 1 in att[0]["failed_ranks"] and att[1]["ok"]
