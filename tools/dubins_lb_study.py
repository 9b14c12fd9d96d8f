#!/usr/bin/env python3
"""CPU only: how much of the Dubins choose-parent pricing the chord lower bound removes (oracle/dubins_oracle.c counters).
    python3 tools/dubins_lb_study.py [--n 100000]    (BASELINE configs[4] shape, query 0 of bench.py --config 5)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import oracle  # noqa: E402
from rrtplanner_amd import hostprep  # noqa: E402
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pairs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=None)
a = ap.parse_args()
cfg = dict(bench.CONFIGS[5])
n = a.n or cfg["n"]
og = perlin_occupancygrid(cfg["grid"], cfg["grid"], thresh=0.33, seed=cfg["grid_seed"])
og8 = hostprep.og_nonzero(og)
free = np.argwhere(og == 0)
xs, xg = random_connected_pairs(og, np.random.default_rng(7), 1)[0]
rng = np.random.default_rng(0)
samples = hostprep.draw_free_samples(rng, free, n)
heads = rng.integers(0, cfg["nh"], size=n)
ps, pg = (int(xs[0]), int(xs[1]), 0), (int(xg[0]), int(xg[1]), 5 % cfg["nh"])
t0 = time.time()
st, r = oracle.dubins_plan(og8, n, 1, ps, pg, samples, heads, r2_rewire=hostprep.radius_threshold(cfg["r_rewire"]), rho=cfg["rho"], nh=cfg["nh"], logs=False, counters=True)
dt = time.time() - t0
acc = r.j - 1
print(f"Dubins-RRT* {cfg['grid']}^2 n={n} r_rewire={cfg['r_rewire']} rho={cfg['rho']} nh={cfg['nh']}: status {st}, {acc} nodes, {dt:.1f} s (oracle with counters)")
print(f"  near-set entries of accepted samples (= word evaluations of the index-order walk) {r.sum_near}")
print(f"  near-set entries of REJECTED samples (priced and thrown away by a kernel that prices before it knows) {r.near_of_rejected}  (+{100 * r.near_of_rejected / max(1, r.sum_near):.1f} %)")
print(f"  entries whose chord bound is not below the cost through the nearest vertex {r.lb_static_skip}  ({100 * r.lb_static_skip / max(1, r.sum_near):.1f} % need no evaluation at all)")
print(f"  evaluations of the (bound, index)-ordered search with tightening {r.lb_evals}  ({100 * r.lb_evals / max(1, r.sum_near):.1f} % of the entries; {r.lb_evals / max(1, acc):.1f} per accepted sample)")
print(f"  sweeps of that search {r.lb_sweeps} ({r.lb_sweeps / max(1, acc):.2f} per accepted sample); index-order walk: cells read {r.sum_cells_cand}")
print(f"  bound violations {r.lb_violations} (must be 0), searches ending elsewhere than the walk {r.lb_mismatch} (must be 0)")
