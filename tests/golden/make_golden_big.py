#!/usr/bin/env python3
"""Bench-scale golden vectors from the REAL reference (policy A), plus the timing pin of the
"reference-like" numpy harness.

Runs only in the build container (reads /root/reference at run time; see make_golden.py for how
the reference is loaded -- identity `numba.njit`, source untouched).  Fixtures hold inputs and
outputs only.

  plans_big_A.npz
    bench1024__star_r64__s0__n20000   query 0 of BASELINE.json configs[3] (bench.py config 4): the bench's own
                                      1024x1024 noise grid, start/goal and planner seed, n = 20000, r_rewire = 64
    noise400__inf_r64_g12__s0__n6000  Informed RRT*, 400x400 noise grid, n = 6000, r_rewire = 64, r_goal = 12
    dtype__<name>                     the reference's 8 occupancy-grid dtypes (tests/test_rrt.py:8-17) on its own
                                      "square" fixture: RRTStar n = 100 -- recorded per dtype
  reference_like_pin.json             seconds of the real reference and of oracle/numpy_like.py on the same
                                      queries in this container (must agree within 20 %)

Usage:  python tests/golden/make_golden_big.py [--skip-timing]
"""
import json
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
from rrtplanner_amd import hostprep  # noqa: E402
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair  # noqa: E402

REF_DTYPES = ("int", "float", "uint32", "uint64", "int32", "int64", "float32", "float64")  # tests/test_rrt.py:8-17


def main():
    ref = mg.load_reference()
    cap = mg.Capture(ref)
    mg.set_policy(True)
    arrays, manifest, timing = {}, [], {}

    def one(cid, gname, og, alg, n, seed, xs, xg, rr, rg, full_graph=False, **extra):
        meta = dict(id=cid, grid=gname, alg=alg, r_rewire=rr, r_goal=rg, seed=seed, n=n,
                    xstart=[int(xs[0]), int(xs[1])], xgoal=[int(xg[0]), int(xg[1])], **extra)
        pl = mg.make_planner(ref, alg, og, n, seed, rr, rg)
        t0 = time.perf_counter()
        mg.record_plan(cap, pl, np.array(xs), np.array(xg), arrays, cid + "__", meta, full_graph=full_graph)
        meta["reference_seconds"] = time.perf_counter() - t0
        manifest.append(meta)
        print(cid, meta.get("vgoal"), f"{meta['reference_seconds']:.1f} s", flush=True)
        return meta

    # ---- the bench grid, the bench's query 0 of config 4 (bench.py: og seed 1, pairs from default_rng(7), planner seed 0)
    og = perlin_occupancygrid(1024, 1024, thresh=0.33, seed=1)
    arrays["grid__bench1024"] = (og != 0).astype(np.uint8)
    xs, xg = random_connected_pair(og, np.random.default_rng(7))
    m = one("bench1024__star_r64__s0__n20000", "bench1024", og, 1, 20000, 0, xs, xg, 64, None)
    timing["bench1024_star_n20000"] = {"reference_s": m["reference_seconds"]}

    og4 = perlin_occupancygrid(400, 400, thresh=0.33, seed=1)
    arrays["grid__noise400"] = (og4 != 0).astype(np.uint8)
    xs4, xg4 = random_connected_pair(og4, np.random.default_rng(7))
    one("noise400__inf_r64_g12__s0__n6000", "noise400", og4, 2, 6000, 0, xs4, xg4, 64, 12)

    # ---- the reference's 8 grid dtypes on its own square fixture (anything != 0 is an obstacle, rrt.py:191,218)
    sq = np.zeros((100, 100))
    sq[25:75, 25:75] = 1
    arrays["grid__square100"] = (sq != 0).astype(np.uint8)
    for name in REF_DTYPES:
        dt = {"int": int, "float": float}.get(name) or getattr(np, name)
        g = sq.astype(dt)
        if np.issubdtype(g.dtype, np.floating):
            g[30, 30] = 0.25  # a fractional obstacle value: != 0, so an obstacle
        one(f"dtype__{name}", "square100", g, 1, 100, 3, (10, 12), (90, 80), 20, None, full_graph=True, og_dtype=name,
            fractional=bool(np.issubdtype(g.dtype, np.floating)))

    arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "plans_big_A.npz"), **arrays)
    mg.set_policy(False)

    if "--skip-timing" in sys.argv:
        return
    # ---- pin of oracle/numpy_like.py (bench.py's "reference_like" comparator) against the real reference, same queries
    import oracle
    from oracle import numpy_like

    og8 = oracle.og_u8(og)
    free = np.argwhere(og == 0)
    samples = hostprep.draw_free_samples(np.random.default_rng(0), free, 20000)
    t0 = time.perf_counter()
    _, _, _, j, it, dl = numpy_like.rrtstar_like(og8, 20000, xs, xg, samples, 64)
    timing["bench1024_star_n20000"].update(numpy_like_s=dl, numpy_like_nodes=int(j - 1), iterations=int(it))
    # config-1 scale (200x200, n = 2000, RRT*, r = 32): reference vs harness
    og2 = perlin_occupancygrid(200, 200, thresh=0.33, seed=1)
    xs2, xg2 = random_connected_pair(og2, np.random.default_rng(7))
    mg.set_policy(True)
    pl = ref.RRTStar(og2, 2000, 32, pbar=False, seed=0)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pl.plan(xs2, xg2)
    tr = time.perf_counter() - t0
    mg.set_policy(False)
    s2 = hostprep.draw_free_samples(np.random.default_rng(0), np.argwhere(og2 == 0), 2000)
    _, _, _, j2, it2, dl2 = numpy_like.rrtstar_like(oracle.og_u8(og2), 2000, xs2, xg2, s2, 32)
    timing["noise200_star_r32_n2000"] = {"reference_s": tr, "numpy_like_s": dl2, "numpy_like_nodes": int(j2 - 1), "iterations": int(it2)}
    for k, v in timing.items():
        v["ratio_numpy_like_over_reference"] = v["numpy_like_s"] / v["reference_s"]
    timing["host"] = {"cpus": os.cpu_count(), "threads_used": 1, "numpy": np.__version__,
                      "note": "build container; numba absent, the reference's two njit functions ran as plain Python"}
    with open(os.path.join(HERE, "reference_like_pin.json"), "w") as f:
        json.dump(timing, f, indent=1)
    print(json.dumps(timing, indent=1))


if __name__ == "__main__":
    main()
