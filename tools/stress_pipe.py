#!/usr/bin/env python3
"""Stress: batches of queries on one CU each (rrt_pipe_kernel) or on small teams, every tree against the oracle.

    python tools/stress_pipe.py [rounds] [team] [queries]

Random grid sizes, obstacle densities, radii (also below a cell and beyond the grid), RRTStandard / RRT* mixed, n up to 6000."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
TEAM = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # 1: rrt_pipe_kernel; 2: the two-worker team with 32 samples per member
NQ = int(sys.argv[3]) if len(sys.argv) > 3 else 256
RMIN, RMAX = float(os.environ.get("STRESS_RMIN", 0)), float(os.environ.get("STRESS_RMAX", 1e9))  # (radii below 16 keep a team of two on 16 samples per member)
rng = np.random.default_rng(20261004)
ctx = _ffi.Context(0)
bad = 0
t0 = time.time()
for rd in range(rounds):
    w, h = int(rng.integers(64, 700)), int(rng.integers(64, 700))
    og = perlin_occupancygrid(w, h, seed=int(rng.integers(0, 1000)))
    og8 = oracle.og_u8(og)
    free = np.argwhere(og8 == 0)
    if free.shape[0] < 100:
        continue
    ctx.set_grid(og8)
    Q, n = NQ, int(rng.choice([300, 1500, 6000]))
    b = _ffi.Batch(ctx, Q, n, team=TEAM)
    keep, refs = [], []
    for q in range(Q):
        xs, xg = random_connected_pair(og, rng)
        samples = hostprep.draw_free_samples(np.random.default_rng(int(rng.integers(0, 1 << 30))), free, n)
        alg = int(rng.integers(0, 2))
        r2 = hostprep.radius_threshold(float(rng.choice([r for r in (3, 12, 25, 40, 64, 150, 2000) if RMIN <= r <= RMAX])))
        qu, k = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2)
        keep.append(k)
        b.set_query(q, qu)
        refs.append(oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2))
    b.launch(); b.sync()
    assert TEAM != 1 or b.kernel_name() == "rrt_pipe_kernel", b.kernel_name()
    for q in range(Q):
        res = b.get_result(q)
        st, ro = refs[q]
        live = ro.j + (1 if ro.found else 0)
        ok = (res.status == st and res.j == ro.j and res.vgoal == ro.vgoal and np.array_equal(res.pts[:live], ro.pts[:live]) and
              np.array_equal(res.parent[:live], ro.parent[:live]) and np.array_equal(res.vcost[:live], ro.vcost[:live]) and
              res.sum_j == ro.sum_j and res.sum_cells_nn == ro.sum_cells_nn and res.sum_near == ro.sum_near)
        if not ok:
            bad += 1
            print(f"MISMATCH round {rd} query {q}: grid {w}x{h} n {n} j {res.j} vs {ro.j}")
    print(f"round {rd}: grid {w}x{h}, n = {n}, {Q} queries, {b.kernel_name()} {b.elapsed_ms():.1f} ms, mismatches so far {bad}", flush=True)
    b.close()
print(f"{rounds} rounds in {time.time() - t0:.0f} s: {'OK' if bad == 0 else 'FAILED'} ({bad} mismatches)")
sys.exit(1 if bad else 0)
