#!/usr/bin/env python3
"""Golden vectors for planners with a CUSTOM cost function, from the REAL reference (build container only; see
make_golden.py for how /root/reference/rrtplanner/rrt.py is loaded -- by path, `numba.njit` as the identity, source untouched).

The cost functions live in tests/costfns.py.  With `discount` the reference's rewire block (rrt.py:531-546, :731-742) really
fires; the fixtures hold what plan() handed to build_graph (points, parents in dict order, vcosts), the goal vertex, the
nearest-vertex log and the generator state, or the exception plan() raised (RRTStarInformed does not catch the ValueError of
a second rewire of one vertex, rrt.py:740).  Policy A (stable argsort) only.

Usage:  python tests/golden/make_golden_costfn.py        (writes tests/golden/plans_costfn_A.npz)
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
from costfns import COSTFNS  # noqa: E402


def main():
    ref = mg.load_reference()
    cap = mg.Capture(ref)
    grids = mg.make_grids()
    corner = np.zeros((40, 40), dtype=int)  # start next to the origin: |xnew| < r_rewire puts the unfilled rows into within()
    corner[15:25, 10:30] = 1
    grids["corner40"] = corner
    mg.set_policy(True)
    arrays, manifest = {}, []
    for gname in ("noise200", "maze64x96", "corner40"):
        arrays["grid__" + gname] = (grids[gname] != 0).astype(np.uint8)
    cases = []
    for fn in ("discount", "manhattan", "downhill"):
        for gname, n in (("noise200", 400), ("noise200", 2000), ("maze64x96", 600)):
            cases += [(gname, "std", 0, None, None, fn, 0, n), (gname, "star_r20", 1, 20, None, fn, 1, n), (gname, "star_r32p5", 1, 32.5, None, fn, 2, n),
                      (gname, "inf_r32_g12", 2, 32, 12, fn, 3, n)]
    for fn in ("discount", "manhattan"):
        cases += [("corner40", "star_r12", 1, 12, None, fn, 0, 300), ("corner40", "inf_r12_g6", 2, 12, 6, fn, 1, 300)]
    for gname, tag, alg, rr, rg, fn, seed, n in cases:
        og = grids[gname]
        xs, xg = (np.array((2, 3)), np.array((35, 36))) if gname == "corner40" else mg.pick_start_goal(og)
        cid = f"{gname}__{tag}__{fn}__s{seed}__n{n}"
        meta = dict(id=cid, grid=gname, alg=alg, r_rewire=rr, r_goal=rg, seed=seed, n=n, costfn=fn, xstart=[int(xs[0]), int(xs[1])], xgoal=[int(xg[0]), int(xg[1])])
        kw = dict(costfn=COSTFNS[fn], pbar=False, seed=seed)
        pl = ref.RRTStandard(og, n, **kw) if alg == 0 else ref.RRTStar(og, n, rr, **kw) if alg == 1 else ref.RRTStarInformed(og, n, rr, rg, **kw)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mg.record_plan(cap, pl, xs, xg, arrays, cid + "__", meta, full_graph=(n <= 400))
        except ValueError as e:  # rrt.py:740
            meta["raises"] = "ValueError"
            meta["raises_msg"] = str(e)
            meta["rng_state"] = mg.rng_state_tuple(pl.rand_gen)
            arrays[cid + "__nearest_log"] = np.asarray(cap.nearest, dtype=np.int32)
        if "vgoal" in meta:
            par = arrays[cid + "__parent"]
            meta["n_rewired_visible"] = int(np.sum(par[1:meta["rows"]] > np.arange(1, meta["rows"])))  # a parent younger than its child
            meta["n_selfloops"] = int(np.sum(par[:meta["rows"]] == np.arange(meta["rows"])))
        manifest.append(meta)
        print(cid, meta.get("vgoal"), meta.get("raises", ""), meta.get("n_rewired_visible", ""), flush=True)
    arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "plans_costfn_A.npz"), **arrays)
    mg.set_policy(False)


if __name__ == "__main__":
    main()
