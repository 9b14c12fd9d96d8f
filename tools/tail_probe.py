#!/usr/bin/env python3
"""Diagnostic: how long each query of a one-CU-per-query batch runs (the batch ends with its slowest query).

    RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so python tools/tail_probe.py [--config 5] [--queries 256]

Per query: wave 0's stamped cycles (its whole life in the kernel) and the vertices it inserted; the workload is bench.py's."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pairs

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=5)
ap.add_argument("--queries", type=int, default=256)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
Q, n, alg = a.queries, cfg["n"], cfg["alg"]
og = perlin_occupancygrid(cfg["grid"], cfg["grid"], thresh=0.33, seed=cfg.get("grid_seed", 1))
og8 = hostprep.og_nonzero(og)
free = np.argwhere(og == 0)
pairs = random_connected_pairs(og, np.random.default_rng(7), Q)
r2 = hostprep.radius_threshold(cfg["r_rewire"])
ctx = _ffi.Context(0); ctx.set_grid(og8)
dub = alg >= _ffi.ALG_DUBINS
b = _ffi.Batch(ctx, Q, n, team=1, dubins=dub)
keep = []
for g in range(Q):
    xs, xg = pairs[g]
    rng = np.random.default_rng(g)
    samples = hostprep.draw_free_samples(rng, free, n)
    if dub:
        heads = rng.integers(0, cfg["nh"], size=n)
        ps, pg = (int(xs[0]), int(xs[1]), (7 * g) % cfg["nh"]), (int(xg[0]), int(xg[1]), (13 * g + 5) % cfg["nh"])
        qu, k = _ffi.make_query(alg, n, ps, pg, samples, r2_rewire=r2, headings=heads, rho=cfg["rho"], nh=cfg["nh"])
    else:
        qu, k = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2)
    keep.append(k); b.set_query(g, qu)
for rep in range(2):
    b.rearm(); b.launch(); b.sync()
ms = b.elapsed_ms()
cyc = np.array([sum(b.debug_cycles(g)[:6]) for g in range(Q)], dtype=np.float64)
js = np.array([b.get_result(g, arrays=False).c.j for g in range(Q)])
near = np.array([b.get_result(g, arrays=False).c.sum_near for g in range(Q)], dtype=np.float64) / n
t = cyc / cyc.max() * ms
print(f"{b.kernel_name()}: {Q} queries, kernel {ms:.1f} ms")
print("per-query time (ms): min %.1f  median %.1f  mean %.1f  p90 %.1f  max %.1f" % (t.min(), np.median(t), t.mean(), np.percentile(t, 90), t.max()))
print("vertices: min %d median %d max %d;  near set per sample: min %.0f median %.0f max %.0f" % (js.min(), np.median(js), js.max(), near.min(), np.median(near), near.max()))
print("correlation of time with vertices %.2f, with near-set size %.2f" % (np.corrcoef(t, js)[0, 1], np.corrcoef(t, near)[0, 1]))
order = np.argsort(-t)[:5]
print("slowest:", [(int(g), round(float(t[g]), 1), int(js[g]), round(float(near[g]))) for g in order])
print("CU time used / CU time held = %.2f" % (t.sum() / (Q * t.max())))
med = int(np.argsort(t)[Q // 2])
c = b.debug_cycles(med)[:6]
print("phases of the median query %d (wave 0, %% of its stamped cycles):" % med, [round(100.0 * x / max(sum(c), 1), 1) for x in c], " cycles per iteration:", [round(x / n) for x in c])
