#!/usr/bin/env python3
"""Diagnostic: per-phase shader cycles of the Dubins kernels (wave 0 of one query) from the stamped build.

    make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so
    RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so python tools/dubins_stamps.py [serial]

Default: rrt_dubins_block_kernel (16 samples per round); `serial`: rrt_expand_kernel<false, true>.

BASELINE config 5's shape (2048 x 2048, n = 100 000, r_rewire = 64, rho = 8, 64 headings), one query.  Never quote the stamped
build's run time; read the shares."""
import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair
grid, n = 2048, 100000
og = perlin_occupancygrid(grid, grid, seed=1)
og8 = hostprep.og_nonzero(og)
xs, xg = random_connected_pair(og, np.random.default_rng(11))
rng = np.random.default_rng(3)
samples = hostprep.draw_free_samples(rng, np.argwhere(og8 == 0), n)
heads = rng.integers(0, 64, size=n)
ctx = _ffi.Context(0); ctx.set_grid(og8)
serial = len(sys.argv) > 1 and sys.argv[1] == 'serial'
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # queries running side by side (the stamps are query 0's)
b = _ffi.Batch(ctx, Q, n, dubins=True, serial=serial)
q, keep = _ffi.make_query(_ffi.ALG_DUBINS_STAR, n, (int(xs[0]), int(xs[1]), 5), (int(xg[0]), int(xg[1]), 20), samples, r2_rewire=hostprep.radius_threshold(64), headings=heads, rho=8.0, nh=64)
for k in range(Q):
    b.set_query(k, q)
for rep in range(2):
    b.rearm(); b.launch(); b.sync()
ms = b.elapsed_ms()
r = b.get_result(0, arrays=False)
cyc = b.debug_cycles(0)
print("%d queries side by side" % Q)
print("kernel %.1f ms, j=%d, %.0f cyc/iter (at 2.4 GHz)" % (ms, r.c.j, ms * 2.4e6 / n))
print("kernel:", b.kernel_name(), " word evaluations per iteration: %.2f" % (r.c.n_words / n))
names = (["A scan", "pricing (wave 0's entries)", "wait for the nearest's word + sweep", "acceptance, test rounds", "D insert", "go2goal"] if serial else
         ["first record stream", "one word per lane", "sweeps (nearest + priced entries)", "pass 2 (stream, words, sweeps)", "wait for the slowest wave", "commit + publication"])
tot = sum(cyc[:6]) or 1
for nm, c in zip(names, cyc[:6]): print("  %-32s %12d  %5.1f%%  %8.1f cyc/iter" % (nm, c, 100 * c / tot, c / n))
print("near", r.c.sum_near / n, "los_cand", r.c.n_los_cand / n)
if not serial:
    print("samples resolved again (a younger vertex nearer than the snapshot nearest): %d (%.2f %%);  retirements that priced younger vertices: %d (%.2f %%), %d words" % (
        cyc[6], 100.0 * cyc[6] / n, cyc[7], 100.0 * cyc[7] / n, cyc[8]))
    print("lock held %.1f %% of the kernel's cycles: %d times, %.2f samples retired each, %.0f cycles each" % (
        100.0 * cyc[11] / (ms * 2.4e6), cyc[12], cyc[13] / max(cyc[12], 1), cyc[11] / max(cyc[12], 1)))
