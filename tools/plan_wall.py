#!/usr/bin/env python3
"""End-to-end wall time of planner.plan() (host sample draw, upload, kernel, download, DiGraph) vs the kernel alone:
the PCIe- and host-inclusive rate quoted in DESIGN.md.   python tools/plan_wall.py [--n 50000] [--alg star|informed|standard]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrtplanner_amd import RRTStandard, RRTStar, RRTStarInformed
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--alg", default="star")
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
og = perlin_occupancygrid(1024, 1024, seed=1)
xs, xg = random_connected_pair(og, np.random.default_rng(7))
mk = {"standard": lambda: RRTStandard(og, a.n, pbar=False, seed=0), "star": lambda: RRTStar(og, a.n, 64, pbar=False, seed=0),
      "informed": lambda: RRTStarInformed(og, a.n, 64, 12, pbar=False, seed=0)}[a.alg]
p = mk()
T = None
for r in range(a.reps):
    del T  # freeing a materialised 50 000-node graph takes longer than planning the next one: keep it out of the timing
    t0 = time.perf_counter()
    T, gv = p.plan(xs, xg)
    t1 = time.perf_counter()
    path = p.route2gv(T, gv)
    t2 = time.perf_counter()
    segs = p.vertices_as_ndarray(T, path)
    t3 = time.perf_counter()
    nn = T.number_of_nodes()  # first touch of the graph itself: fills the dictionaries
    t4 = time.perf_counter()
    print(f"rep {r}: plan() {1e3 * (t1 - t0):.1f} ms, route2gv {1e3 * (t2 - t1):.2f} ms, vertices_as_ndarray {1e3 * (t3 - t2):.2f} ms, "
          f"materialising the DiGraph ({nn} nodes) {1e3 * (t4 - t3):.1f} ms, path {len(path)} vertices, j {p.last_stats['j']}")
