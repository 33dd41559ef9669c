"""CPU worker (gloo, no GPU): drives the host side of the N>1 path -- the ttx_transport thunks the engine calls
per sweep (right-going then left-going neighbour messages, MAX and SUM all-reduces) and the bond-group split --
exactly in the order ttx_engine.hip::xfer_neighbours / allreduce_dev use them."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from ttcross_amd import engine as E
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nproc = int(sys.argv[1])
    sendrecv, allreduce = E.make_dist_transport(dist)
    g0, G = E.split_groups(nproc, rank, world)
    left, right = E.neighbour_ranks(nproc, rank, world)
    assert G >= 1 and (left == -1) == (rank == 0) and (right == -1) == (rank == world - 1)
    # neighbour messages: payload encodes (sender rank, direction)
    MSZ = 4096
    sR = np.full(MSZ, 10 * rank + 1, dtype=np.uint8); sL = np.full(MSZ, 10 * rank + 2, dtype=np.uint8)
    rL = np.zeros(MSZ, dtype=np.uint8); rR = np.zeros(MSZ, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert sendrecv(None, right, p(sR), MSZ, left, p(rL), MSZ) == 0      # right-going
    assert sendrecv(None, left, p(sL), MSZ, right, p(rR), MSZ) == 0      # left-going
    if left >= 0:
        assert (rL == 10 * left + 1).all()
    else:
        assert (rL == 0).all()
    if right >= 0:
        assert (rR == 10 * right + 2).all()
    else:
        assert (rR == 0).all()
    # all-reduces: MAX of (amax, pivotmax, -pivotmin), SUM of the per-sweep summary with disjoint slots
    red = np.array([1.0 + rank, -5.0 * rank, -999e9 if rank else -0.25, 0.0])
    assert allreduce(None, red.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 4, 1) == 0
    assert red.tolist() == [float(world), 0.0, -0.25, 0.0]
    summ = np.zeros(8 + nproc)
    summ[0] = 100 + rank
    summ[8 + g0: 8 + g0 + G] = np.arange(g0, g0 + G) + 0.5
    assert allreduce(None, summ.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), summ.size, 0) == 0
    assert summ[0] == sum(100 + r for r in range(world)) and np.array_equal(summ[8:], np.arange(nproc) + 0.5)
    dist.barrier()
    print(f"[rank {rank}] groups {g0}..{g0+G-1} neighbours ({left},{right}) OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
