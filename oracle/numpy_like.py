"""A numpy harness with the per-iteration operation mix of the reference's RRT* loop (TEST INFRASTRUCTURE: bench.py's
cpu_baseline leg only).

Written from scratch from the behaviour described in SURVEY.md 8(a) -- it is not the reference's code: per iteration the
reference (rrtplanner/rrt.py:498-548) does a full-array subtract + norm + argsort over all n rows (near, :150-155), a
full-array squared-distance mask (within, :176-181), a compiled line-of-sight walk (collisionfree, :202-229; here the C
oracle's), and a Python loop over the near set with a cost callback (:511-546).  The array sizes are the full capacity n
whatever the live node count, as there.  Only the time per iteration is of interest (the "reference-like" CPU number of
SURVEY.md 8(d)); the tree it builds follows the same rules (stable nearest, strict radius, first cheaper visible parent) and is
compared with the oracle's in tests/test_oracle_golden.py.
"""
import math
import time

import numpy as np

from . import collisionfree

_SENTINEL = np.iinfo(np.int64).min


def rrtstar_like(og8, n, xs, xg, samples, r_rewire, max_iters=None, time_limit=None):
    """Run up to `max_iters` iterations (or until `time_limit` seconds) of an n-capacity RRT* query.
    Returns (points, parents, vcosts, j, iterations done, seconds)."""
    points = np.full((n, 2), _SENTINEL, dtype=np.int64)
    vcosts = np.full((n,), np.inf)
    parents = {0: None}
    points[0] = xs
    vcosts[0] = 0.0
    sampled = set()
    r2 = r_rewire * r_rewire
    j, i = 1, 0
    iters = n if max_iters is None else min(n, max_iters)
    t0 = time.perf_counter()
    while i < iters:
        if time_limit is not None and (i & 63) == 0 and time.perf_counter() - t0 > time_limit:
            break
        x = samples[i].astype(np.int64)
        # nearest: all n rows, float norm, argsort (sentinel rows wrap to huge distances)
        with np.errstate(over="ignore"):
            diff = points - x
            order = np.argsort(np.linalg.norm(diff, axis=1), kind="stable")
            vnearest = int(order[0])
            # radius ball: all n rows, integer squared distance (sentinel rows wrap; their cost is inf)
            d2 = diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]
            vnear = np.atleast_1d(np.squeeze(np.argwhere(d2 < r2)))
        ok, _ = collisionfree(og8, points[vnearest], x)
        key = (int(x[0]), int(x[1]))
        if ok and key not in sampled and j != n:
            sampled.add(key)
            d = points[vnearest] - x
            vbest, cbest = vnearest, vcosts[vnearest] + math.sqrt(d[0] * d[0] + d[1] * d[1])
            for vn in vnear.tolist():  # choose parent: first strictly cheaper node with a free line of sight
                d = points[vn] - x
                cn = vcosts[vn] + math.sqrt(float(d[0]) * float(d[0]) + float(d[1]) * float(d[1])) if vcosts[vn] < np.inf else np.inf
                if cn < cbest and collisionfree(og8, points[vn], x)[0]:
                    vbest, cbest = vn, cn
            points[j] = x
            vcosts[j] = cbest
            parents[j] = vbest
            for vn in vnear.tolist():  # rewire scan: the predicate of the reference is never true with this cost
                d = points[vn] - x
                if vcosts[vn] < np.inf and cbest + math.sqrt(float(d[0]) * float(d[0]) + float(d[1]) * float(d[1])) < vcosts[vn]:
                    pass
            j += 1
        i += 1
    return points, parents, vcosts, j, i, time.perf_counter() - t0
