#!/usr/bin/env python3
"""Where the wall time of one bench step goes on the host side: per call of the step (rearm / launch / sync), over K steps.
    python3 tools/step_gap.py [--steps 20] [--config 2]
Prints per-call mean / max microseconds, the kernel's HIP-event time and the step's wall time."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rrtplanner_amd import _ffi, hostprep  # noqa: E402
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pairs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
Q, n = cfg["queries"], cfg["n"]
og = perlin_occupancygrid(cfg["grid"], cfg["grid"], seed=1)
og8 = hostprep.og_nonzero(og)
free = np.argwhere(og == 0)
pairs = random_connected_pairs(og, np.random.default_rng(7), Q)
ctx = _ffi.Context(0)
ctx.set_grid(og8)
b = _ffi.Batch(ctx, Q, n)
keep = []
for q in range(Q):
    qu, k = _ffi.make_query(cfg["alg"], n, pairs[q][0], pairs[q][1], hostprep.draw_free_samples(np.random.default_rng(q), free, n),
                            r2_rewire=hostprep.radius_threshold(cfg["r_rewire"]))
    keep.append(k)
    b.set_query(q, qu)
for _ in range(3):
    b.rearm(); b.launch(); b.sync()
ctx.sync()
T = {k: [] for k in ("rearm", "launch", "sync", "elapsed", "step", "kernel")}
for _ in range(a.steps):
    t0 = time.perf_counter(); b.rearm()
    t1 = time.perf_counter(); b.launch()
    t2 = time.perf_counter(); b.sync()
    t3 = time.perf_counter(); ms = b.elapsed_ms()
    t4 = time.perf_counter()
    for k, v in (("rearm", t1 - t0), ("launch", t2 - t1), ("sync", t3 - t2), ("elapsed", t4 - t3), ("step", t4 - t0), ("kernel", ms * 1e-3)):
        T[k].append(v * 1e6)
print(f"config {a.config}: {Q} queries, n = {n}, kernel {b.kernel_name()}, {a.steps} steps (microseconds)")
for k, v in T.items():
    print(f"  {k:8s} mean {np.mean(v):10.1f}  min {np.min(v):10.1f}  max {np.max(v):10.1f}")
print(f"  step - kernel: mean {np.mean(T['step']) - np.mean(T['kernel']):.1f} us   (sync - kernel: {np.mean(T['sync']) - np.mean(T['kernel']):.1f} us)")
