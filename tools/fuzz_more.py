#!/usr/bin/env python3
"""The fuzz of tests/test_gpu_parity.py::test_fuzz_small_queries_vs_oracle with other seeds (the one-CU pipeline and the small teams).

    python tools/fuzz_more.py [seed ...]"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle
from rrtplanner_amd import _ffi
from rrtplanner_amd.oggen import perlin_occupancygrid
import test_gpu_parity as T

seeds = [int(x) for x in sys.argv[1:]] or [1, 2, 3]
ctx = _ffi.Context(0)
total = 0
for seed in seeds:
    for kernel in ("block", "team2", "team3"):
        rng = np.random.default_rng(seed)
        for case in range(400):
            w, h = int(rng.integers(8, 90)), int(rng.integers(8, 90))
            dens = rng.choice([0.0, 0.1, 0.3, 0.5])
            og8 = (rng.uniform(size=(w, h)) < dens).astype(np.uint8)
            if case % 3 == 0:
                og8 = oracle.og_u8(perlin_occupancygrid(w, h, seed=case + seed))
            free = np.argwhere(og8 == 0)
            if free.shape[0] < 2:
                continue
            alg = int(rng.integers(0, 2))
            n = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 192, 193, 257, 700, 1500, 2500]))
            rr = float(rng.choice([0, 1, 3, 12, 16, 17, 25, 40, 64, 500]))
            xs = free[rng.integers(0, free.shape[0])] if case % 11 else np.array([int(rng.integers(0, w)), int(rng.integers(0, h))])
            xg = free[rng.integers(0, free.shape[0])]
            ctx.set_grid(og8)
            try:
                T._oracle_vs_device(ctx, og8, alg, n, case, xs, xg, rr if alg else None, None, kernel=kernel)
            except AssertionError as e:
                print(f"MISMATCH seed {seed} kernel {kernel} case {case}: grid {w}x{h} dens {dens} alg {alg} n {n} r {rr} xs {xs} xg {xg}")
                raise
            total += 1
    print(f"seed {seed}: ok ({total} cases so far)", flush=True)
print("fuzz ok:", total, "cases")
