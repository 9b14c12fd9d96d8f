#!/usr/bin/env python3
"""Diagnostic: per-phase shader cycles of rrt_expand_kernel (wave 0 of query 0) from the stamped build.

    make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so
    RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so python tools/stamps.py [--n 50000] [--alg 1] [--queries 1]

Never quote the stamped build's run time; read the shares."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--alg", type=int, default=1)
ap.add_argument("--queries", type=int, default=1)
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--team", type=int, default=None, help="CUs per query: 1, 2, 4 (default: as many as fit)")
ap.add_argument("--serial", action="store_true")
a = ap.parse_args()
og = perlin_occupancygrid(a.grid, a.grid, seed=1)
free = np.argwhere(og == 0)
ctx = _ffi.Context(0); ctx.set_grid(hostprep.og_nonzero(og))
b = _ffi.Batch(ctx, a.queries, a.n, serial=a.serial, team=a.team)
sg = np.random.default_rng(7); keep = []
for q in range(a.queries):
    xs, xg = random_connected_pair(og, sg)
    s = hostprep.draw_free_samples(np.random.default_rng(q), free, a.n)
    qu, k = _ffi.make_query(a.alg, a.n, xs, xg, s, r2_rewire=64 * 64, goal_d2=hostprep.goal_threshold(12),
                            Cmat=hostprep.rotation_to_world_frame(xs, xg)); keep.append(k)
    b.set_query(q, qu)
for rep in range(2):
    b.rearm(); b.launch(); b.sync()
ms = b.elapsed_ms()
r = b.get_result(0, arrays=False)
cyc = b.debug_cycles(0)
names = (["A scan+wave-reduce", "barrier 1", "B nearest+LoS+dup", "C choose parent", "D insert", "go2goal"] if a.serial else
         ["A scan + reductions", "barrier (scan skew)", "B owner phase (wave 0)", "barrier (owner tail)", "C wait members + commit", "go2goal"])
tot = sum(cyc) or 1
print(f"kernel {ms:.2f} ms ({b.kernel_name()}), n={a.n}, nodes={r.c.j}, iters/s={a.n/ms*1e3:.0f}, status={r.c.status}")
w = cyc[6:]; cyc = cyc[:6]
print("per-wave owner-phase cyc/iter:", [round(x / a.n) for x in w[:16]])
print("per-wave near-set part cyc/iter:", [round(x / a.n) for x in w[16:]])
if os.environ.get("RRT_STAMPS_PIPE"):
    nblk = (a.n + 63) // 64
    print("committer cyc/block [after the end-of-block barrier, part A, rounds: re-resolutions + barriers, ordered loop, publish, "
          "wave-0 re-resolutions without near-set redo, samples in rounds, loop top to part A, rounds: lists between them, store pass]:",
          [round(x / nblk, 1) for x in w[:10]])
    print("committer wave-0 resolves: with near-set redo: %d x %.0f cyc; others: %d x %.0f cyc; rounds/block %.2f" % (w[11], w[10] / max(w[11], 1), w[5], w[12] / max(w[5], 1), (w[13] & 0xffffffff) / nblk))
    print("committer: samples settled by their own lanes (no round): %.2f per block" % ((w[13] >> 32) / nblk))
    print("committer: fetch of the next block's records %.0f cyc/block, %.0f of them waiting for the workers' flags; wave 0 at the end-of-block barrier %.0f cyc/block" % ((w[14] & 0xffffffff) / nblk, (w[14] >> 32) / nblk, w[15] / nblk))
    print("worker 1  cyc/block [resolve, hand over, go wait, take]:", [round(x / nblk, 1) for x in w[16:20]])
if os.environ.get("RRT_STAMPS_RAW"):
    print("raw wcyc[0:16]:", list(w[:16]))
    print("raw wcyc[16:]:", list(w[16:]))
for nm, c in zip(names, cyc):
    print(f"  {nm:22s} {c:14d} cyc  {100*c/tot:5.1f}%  {c/a.n:9.1f} cyc/iter")
print(f"  total stamped cycles {tot} = {tot/a.n:.0f} cyc/iter ; los_cand={r.c.n_los_cand} near={r.c.sum_near}")
