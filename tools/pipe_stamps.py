#!/usr/bin/env python3
"""Diagnostic: per-phase shader cycles of rrt_pipe_kernel (wave 0 of query 0) from the stamped build.

    make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so
    RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so python tools/pipe_stamps.py [--n 50000] [--alg 1] [--queries 1] [--grid 1024]

Never quote the stamped build's run time; read the shares."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--alg", type=int, default=1)
ap.add_argument("--queries", type=int, default=1)
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--r", type=float, default=64.0)
ap.add_argument("--team", type=int, default=1, help="cap on the CUs per query (0: as many as fit -- the team pipeline for a few CUs per query)")
a = ap.parse_args()
og = perlin_occupancygrid(a.grid, a.grid, seed=1)
free = np.argwhere(og == 0)
ctx = _ffi.Context(0); ctx.set_grid(hostprep.og_nonzero(og))
b = _ffi.Batch(ctx, a.queries, a.n, team=(a.team or None))
sg = np.random.default_rng(7); keep = []
for q in range(a.queries):
    xs, xg = random_connected_pair(og, sg)
    s = hostprep.draw_free_samples(np.random.default_rng(q), free, a.n)
    qu, k = _ffi.make_query(a.alg, a.n, xs, xg, s, r2_rewire=hostprep.radius_threshold(a.r)); keep.append(k)
    b.set_query(q, qu)
for rep in range(2):
    b.rearm(); b.launch(); b.sync()
ms = b.elapsed_ms()
r = b.get_result(0, arrays=False)
cyc = b.debug_cycles(0)
n = a.n
print(f"kernel {ms:.2f} ms ({b.kernel_name()}), n={n}, nodes={r.c.j}, iters/s={n/ms*1e3:.0f}, status={r.c.status}")
names = ["first record stream", "one price per lane", "lines of sight (nearest + priced)", "pass 2 (stream, prices, lines)", "waiting (window full)", "deposit"]
tot = sum(cyc[:6]) or 1
for nm, c in zip(names, cyc[:6]): print("  %-34s %12d  %5.1f%%  %8.1f cyc/sample of wave 0" % (nm, c, 100 * c / tot, c * 15 / n))
print("wave 0's stamped cycles / kernel time = %.2f GHz" % (tot / (ms * 1e6)))
print("near %.1f  los_cand %.2f per sample" % (r.c.sum_near / n, r.c.n_los_cand / n))
print("samples resolved again: %d (%.2f %%)" % (cyc[6], 100.0 * cyc[6] / n))
passes, one_cyc = cyc[8] & ((1 << 24) - 1), cyc[8] >> 24
tot_r = (cyc[7] + one_cyc + cyc[9] + cyc[10]) or 1
print("the retiring wave: waiting for the head %.1f %%, publishing %.1f %% (%d publications, %.0f cycles each), passes %.1f %% (%d passes, %.2f heads and %.0f cycles each), heads on their own %.1f %%" % (
    100.0 * cyc[9] / tot_r, 100.0 * cyc[10] / tot_r, cyc[13], cyc[10] / max(cyc[13], 1), 100.0 * cyc[7] / tot_r, passes, cyc[12] / max(passes, 1), cyc[7] / max(passes, 1),
    100.0 * one_cyc / tot_r))
hn, hc = cyc[14:19], cyc[19:24]
print("wave 0's samples by the time they took it (cycles): " + ", ".join("%s: %d (%.1f %% of its time)" % (nm, a_, 100.0 * c_ / max(sum(hc), 1)) for nm, a_, c_ in zip(("< 20 k", "20 - 40 k", "40 - 80 k", "80 - 160 k", "> 160 k"), hn, hc)))
print("wave 0 found the window full for %d of its samples (%d sleeps)" % (cyc[24], cyc[25]))
