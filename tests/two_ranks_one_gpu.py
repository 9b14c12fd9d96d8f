#!/usr/bin/env python3
"""One RANK of the world > 1 path of the C ABI on a box with ONE GPU (test infrastructure; started by
tests/test_multi_rank_one_gpu.py as two fresh child processes, ranks 0 and 1, both on device 0).

The collectives go through tests/fake_rccl (RRT_RCCL_LIB -> rrt_comm_use_library): RCCL itself refuses two ranks on one device.
What runs for real: rrt_comm_init with world = 2, the size-check all-reduce, rrt_gather with two slabs, rrt_gather_fetch of the
OTHER rank's queries (the rank * bytes_per_rank offset, the slab's self-describing tail) -- compared with the CPU oracle -- and the
RRT_E_COMM refusals (slabs of different sizes; another batch's slabs).

    python tests/two_ranks_one_gpu.py RANK WORLD ID_FILE
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from rrtplanner_amd import _ffi, hostprep, multi  # noqa: E402
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pairs  # noqa: E402


def main():
    rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    ctx = _ffi.Context(0)  # every rank on the one GPU
    multi.init_comm(ctx, rank, world, path=idfile)
    og = perlin_occupancygrid(256, 256, seed=4)
    og8 = oracle.og_u8(og)
    ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    assert ctx.allreduce([float(rank + 1), -float(rank)], "max").tolist() == [float(world), 0.0]
    assert ctx.allreduce([float(rank + 1)], "sum").tolist() == [world * (world + 1) / 2.0]
    ctx.barrier()
    r2 = hostprep.radius_threshold(30)
    batches = []
    for bi, (Q, n) in enumerate(((3, 2500), (3, 1800))):  # two batches of three queries per rank
        pairs = random_connected_pairs(og, np.random.default_rng(9 + bi), Q * world)
        b = _ffi.Batch(ctx, Q, n, team=8)  # (capped teams: two processes share the device, each claims CUs on its own)
        keep, refs = [], {}
        for g in range(Q * world):  # global query g runs on rank g % world, slot g // world
            xs, xg = pairs[g]
            samples = hostprep.draw_free_samples(np.random.default_rng(100 * bi + g), free, n)
            refs[g] = oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=r2, logs=False)
            if multi.owner_of(g, world) == rank:
                qu, k = _ffi.make_query(1, n, xs, xg, samples, r2_rewire=r2)
                keep.append(k)
                b.set_query(multi.local_slot(g, world), qu)
        b.launch()
        b.sync()
        ptr, nbytes = b.gather()
        ctx.sync()
        assert ptr and nbytes == b.result_block()[1]
        for g in range(Q * world):  # every rank reads EVERY query, its own and the other ranks'
            st, ro = refs[g]
            res = b.gather_fetch(multi.owner_of(g, world), multi.local_slot(g, world))
            live = ro.j + (1 if ro.found else 0)
            assert (res.status, res.j, res.vgoal, res.found) == (st, ro.j, ro.vgoal, ro.found), (rank, bi, g)
            assert np.array_equal(res.pts[:live], ro.pts[:live]) and np.array_equal(res.parent[:live], ro.parent[:live]), (rank, bi, g)
            assert np.array_equal(res.vcost[:live], ro.vcost[:live]), (rank, bi, g)
        batches.append(b)
        ctx.barrier()
    # the gathered slabs are the SECOND batch's: the first batch is refused, not served another batch's trees
    try:
        batches[0].gather_fetch((rank + 1) % world, 0)
        raise SystemExit("another batch's slabs were served")
    except _ffi.RRTError as e:
        assert e.code == _ffi.RRT_E_COMM, e
    try:
        batches[1].gather_fetch(world, 0)
        raise SystemExit("a rank outside the world was served")
    except _ffi.RRTError as e:
        assert e.code == _ffi.RRT_E_ARG, e
    # slabs of different sizes: every rank gets RRT_E_COMM from the size check, nobody hangs in the all-gather
    odd = _ffi.Batch(ctx, 3 if rank == 0 else 2, 1800, team=8)
    try:
        odd.gather()
        raise SystemExit("ranks with different slabs gathered")
    except _ffi.RRTError as e:
        assert e.code == _ffi.RRT_E_COMM and "different sizes" in str(e), e
    ctx.barrier()
    odd.close()
    for b in batches:
        b.close()
    ctx.comm_destroy()
    ctx.close()
    print(f"RANK {rank} of {world} OK", flush=True)


if __name__ == "__main__":
    main()
