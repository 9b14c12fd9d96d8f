#!/usr/bin/env python3
"""Run one query batch (for rocprofv3 PMC / kernel-trace runs).  python tools/one_query.py --alg 1 --n 50000 --queries 1"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000); ap.add_argument("--alg", type=int, default=1)
ap.add_argument("--queries", type=int, default=1); ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
og = perlin_occupancygrid(a.grid, a.grid, seed=1); free = np.argwhere(og == 0)
ctx = _ffi.Context(0); ctx.set_grid(hostprep.og_nonzero(og))
b = _ffi.Batch(ctx, a.queries, a.n); sg = np.random.default_rng(7); keep = []
for q in range(a.queries):
    xs, xg = random_connected_pair(og, sg)
    s = hostprep.draw_free_samples(np.random.default_rng(q), free, a.n)
    qu, k = _ffi.make_query(a.alg, a.n, xs, xg, s, r2_rewire=64 * 64, goal_d2=hostprep.goal_threshold(12), Cmat=hostprep.rotation_to_world_frame(xs, xg)); keep.append(k)
    b.set_query(q, qu)
for r in range(a.reps):
    b.rearm(); b.launch(); b.sync()
    print("kernel ms", b.elapsed_ms(), "status", b.get_result(0, arrays=False).c.status, "j", b.get_result(0, arrays=False).c.j)
