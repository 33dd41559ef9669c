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
