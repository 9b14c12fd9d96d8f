#!/usr/bin/env python3
"""Repeat one query many times on every team size and compare every result array with the first run: a rare hand-off race
would show up as a difference.   python tools/stress_team.py [--n 30000] [--reps 25] [--alg 1]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=30000)
ap.add_argument("--reps", type=int, default=25)
ap.add_argument("--alg", type=int, default=1)
ap.add_argument("--queries", type=int, default=3)
a = ap.parse_args()
og = perlin_occupancygrid(1024, 1024, seed=1)
free = np.argwhere(og == 0)
ctx = _ffi.Context(0)
ctx.set_grid(hostprep.og_nonzero(og))
sg = np.random.default_rng(7)
qs = []
for q in range(a.queries):
    xs, xg = random_connected_pair(og, sg)
    s = hostprep.draw_free_samples(np.random.default_rng(q), free, a.n)
    qs.append(_ffi.make_query(a.alg, a.n, xs, xg, s, r2_rewire=64 * 64, goal_d2=hostprep.goal_threshold(12),
                              Cmat=hostprep.rotation_to_world_frame(xs, xg)))
ref = None
bad = 0
for team, pipe in ((None, True), (None, False), (32, True), (16, True), (16, False), (8, True), (4, True), (4, False), (3, True), (2, True), (2, False),
                   (1, True)):
    b = _ffi.Batch(ctx, a.queries, a.n, team=team, pipe=pipe)
    for q, (qu, keep) in enumerate(qs):
        b.set_query(q, qu)
    for rep in range(a.reps):
        b.rearm(); b.launch(); b.sync()
        out = []
        for q in range(a.queries):
            r = b.get_result(q)
            live = r.j + (1 if r.found else 0)
            out.append((r.status, r.j, r.vgoal, r.pts[:live].copy(), r.parent[:live].copy(), r.vcost[:live].copy()))
        if ref is None:
            ref = out
        for q in range(a.queries):
            same = out[q][:3] == ref[q][:3] and all(np.array_equal(x, y) for x, y in zip(out[q][3:], ref[q][3:]))
            if not same:
                bad += 1
                print(f"MISMATCH team={team} pipe={pipe} rep={rep} query={q}: status/j/vgoal {out[q][:3]} vs {ref[q][:3]}")
    print(f"team cap {team} pipe {pipe}: cus/query {b.team()[0]}, fallbacks {b.team()[1]}, {a.reps} repetitions, last kernel {b.elapsed_ms():.2f} ms")
    b.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
