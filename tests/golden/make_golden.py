#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (reads /root/reference at run time; nothing of the
reference is copied into the repository -- the fixtures hold inputs and outputs only).

How the reference is executed: /root/reference/rrtplanner/rrt.py is loaded by file path.  Its
only missing import is `numba` (not installed, no network); `numba.njit` is registered as the
identity decorator, i.e. the two jitted functions (r2norm, collisionfree -- integer-only on this
path) run as the plain Python they are written in.  The reference source is not modified.

Policy A (canonical): np.argsort is forced to kind="stable" inside this process, so ties in
rrt.py:154 and rrt.py:317 resolve to the lowest index.  Policy B: numpy's default argsort
(implementation-defined ties), recorded for prefix checks.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import importlib.util
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/rrtplanner/rrt.py"

from rrtplanner_amd.oggen import largest_free_component, perlin_occupancygrid  # noqa: E402  (our own grid generator)

INT64_MIN = np.iinfo(np.int64).min
INT32_MIN = np.iinfo(np.int32).min


def load_reference():
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    sys.modules["numba"] = nb
    spec = importlib.util.spec_from_file_location("ref_rrt", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


_orig_argsort = np.argsort


def set_policy(stable: bool):
    if stable:
        np.argsort = lambda a, *args, **kw: _orig_argsort(a, kind="stable")
    else:
        np.argsort = _orig_argsort


# --------------------------------------------------------------------------------------- grids
def make_grids():
    g = {}
    g["empty43x100"] = np.zeros((43, 100), dtype=int)
    sq = np.zeros((100, 100), dtype=int)
    sq[25:75, 25:75] = 1  # the reference's own "square" fixture (tests/test_rrt.py:34-37)
    g["square100"] = sq
    g["noise200"] = perlin_occupancygrid(200, 200, thresh=0.33, seed=1)
    mz = np.zeros((64, 96), dtype=int)  # walls with gaps: many blocked lines of sight
    for x in range(8, 64, 12):
        mz[x, :] = 1
        for gap in range(5 + (x % 7), 96, 23):
            mz[x, gap:gap + 4] = 0
    mz[:, 47] = 1
    mz[3:7, 47] = 0
    mz[40:45, 47] = 0
    g["maze64x96"] = mz
    return g


def pick_start_goal(og, seed=7):
    cells = np.argwhere(largest_free_component(og))
    r = np.random.default_rng(seed)
    return cells[r.integers(0, cells.shape[0])], cells[r.integers(0, cells.shape[0])]


# --------------------------------------------------------------------------------------- capture
class Capture:
    """Wraps build_graph / near of the loaded reference module to record what plan() computed."""

    def __init__(self, ref):
        self.ref = ref
        self.bg_args = None
        self.nearest = []
        orig_bg = ref.RRT.build_graph
        orig_near = ref.RRT.near
        cap = self

        def bg(self_, vgoal, points, parents, vcosts):
            cap.bg_args = (vgoal, points.copy(), dict(parents), vcosts.copy())
            return orig_bg(self_, vgoal, points, parents, vcosts)

        def near(points, x):
            s = orig_near(points, x)
            cap.nearest.append(int(s[0]))
            return s

        ref.RRT.build_graph = bg
        ref.RRT.near = staticmethod(near)

    def reset(self):
        self.bg_args = None
        self.nearest = []


def make_planner(ref, alg, og, n, seed, r_rewire, r_goal):
    if alg == 0:
        return ref.RRTStandard(og, n, pbar=False, seed=seed)
    if alg == 1:
        return ref.RRTStar(og, n, r_rewire, pbar=False, seed=seed)
    return ref.RRTStarInformed(og, n, r_rewire, r_goal, pbar=False, seed=seed)


def rng_state_tuple(gen):
    s = gen.bit_generator.state
    return [str(s["state"]["state"]), str(s["state"]["inc"]), int(s["has_uint32"]), int(s["uinteger"])]


def record_plan(cap, planner, xs, xg, arrays, prefix, meta, full_graph):
    cap.reset()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            T, gv = planner.plan(xs, xg)
    except IndexError as e:
        meta["raises"] = "IndexError"
        meta["raises_msg"] = str(e)
        meta["rng_state"] = rng_state_tuple(planner.rand_gen)
        arrays[prefix + "nearest_log"] = np.asarray(cap.nearest, dtype=np.int32)
        return
    vgoal, points, parents, vcosts = cap.bg_args
    rows = points.shape[0]
    pts32 = np.where(points == INT64_MIN, INT32_MIN, points).astype(np.int32)
    par = np.full(rows, -1, dtype=np.int32)
    for c, p in parents.items():
        if p is not None:
            par[c] = p
    arrays[prefix + "pts"] = pts32
    arrays[prefix + "vcost"] = vcosts.astype(np.float64)
    arrays[prefix + "parent"] = par
    arrays[prefix + "parent_order"] = np.asarray([c for c in parents.keys()], dtype=np.int32)
    arrays[prefix + "nearest_log"] = np.asarray(cap.nearest, dtype=np.int32)
    try:
        path = planner.route2gv(T, gv)
    except Exception as e:  # (custom cost functions only: the reference's rewire block can cut the goal off the root)
        meta["route_raises"] = type(e).__name__
        path = []
    arrays[prefix + "path"] = np.asarray([int(v) for v in path], dtype=np.int32)
    meta["vgoal"] = int(gv)
    meta["rows"] = int(rows)
    meta["n_nodes"] = T.number_of_nodes()
    meta["n_edges"] = T.number_of_edges()
    meta["rng_state"] = rng_state_tuple(planner.rand_gen)
    meta["gv_type"] = type(gv).__name__
    if full_graph:
        arrays[prefix + "node_order"] = np.asarray([int(v) for v in T.nodes], dtype=np.int32)
        ed = list(T.edges(data=True))
        arrays[prefix + "edge_uv"] = np.asarray([[int(u), int(v)] for u, v, _ in ed], dtype=np.int32).reshape(-1, 2)
        arrays[prefix + "edge_dist"] = np.asarray([d["dist"] for _, _, d in ed], dtype=np.float64)
        arrays[prefix + "edge_cost"] = np.asarray([d["cost"] for _, _, d in ed], dtype=np.float64)
        meta["edge_dist_type"] = type(ed[0][2]["dist"]).__name__ if ed else ""
        meta["edge_cost_type"] = type(ed[0][2]["cost"]).__name__ if ed else ""
        meta["pt_dtype"] = str(T.nodes[0]["pt"].dtype)
        arrays[prefix + "path_pts"] = planner.vertices_as_ndarray(T, path).astype(np.int64).reshape(-1, 2, 2)
    if hasattr(planner, "ellipses"):
        keys = list(planner.ellipses.keys())
        arrays[prefix + "ell_keys"] = np.asarray(keys, dtype=np.int32)
        vals = [planner.ellipses[k] for k in keys]
        arrays[prefix + "ell_vals"] = np.asarray([[v[0][0], v[0][1], v[1], v[2], v[3]] for v in vals], dtype=np.float64).reshape(-1, 5)


ALGS = [
    # (tag, alg, r_rewire, r_goal)
    ("std", 0, None, None),
    ("star_r20", 1, 20, None),
    ("star_r32", 1, 32, None),
    ("star_r64p5", 1, 64.5, None),
    ("inf_r32_g12", 2, 32, 12),
    ("inf_r20_g5", 2, 20, 5.0),
]


def gen_plans(ref, cap, grids, stable, seeds, ns, out_name):
    set_policy(stable)
    arrays, manifest = {}, []
    for gname, og in grids.items():
        arrays["grid__" + gname] = (og != 0).astype(np.uint8)
        xs, xg = pick_start_goal(og)
        for tag, alg, rr, rg in ALGS:
            for seed in seeds:
                for n in ns:
                    cid = f"{gname}__{tag}__s{seed}__n{n}"
                    meta = dict(id=cid, grid=gname, alg=alg, r_rewire=rr, r_goal=rg, seed=seed, n=n,
                                xstart=[int(xs[0]), int(xs[1])], xgoal=[int(xg[0]), int(xg[1])])
                    pl = make_planner(ref, alg, og, n, seed, rr, rg)
                    record_plan(cap, pl, xs, xg, arrays, cid + "__", meta, full_graph=(n <= 100))
                    manifest.append(meta)
                    print(cid, meta.get("vgoal"), meta.get("raises", ""), flush=True)
    arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, out_name), **arrays)
    set_policy(False)


def gen_special(ref, cap, grids):
    """Edge cases of the path: unreachable goal (both branches of rrt.py:328-331 / the fault at
    :318), start == goal, n == 1, a sample equal to xstart, consecutive plan() calls on one planner
    (RNG continues, rrt.py:85) with set_og in between (anim.py:92-93 caller pattern)."""
    set_policy(True)
    arrays, manifest = {}, []
    wall = np.zeros((20, 20), dtype=int)
    wall[10, :] = 1
    arrays["grid__wall20"] = (wall != 0).astype(np.uint8)
    for gname, og in grids.items():
        arrays["grid__" + gname] = (og != 0).astype(np.uint8)

    def one(cid, og_name, og, alg, n, seed, xs, xg, rr=None, rg=None, **extra):
        meta = dict(id=cid, grid=og_name, alg=alg, r_rewire=rr, r_goal=rg, seed=seed, n=n,
                    xstart=[int(xs[0]), int(xs[1])], xgoal=[int(xg[0]), int(xg[1])], **extra)
        pl = make_planner(ref, alg, og, n, seed, rr, rg)
        record_plan(cap, pl, np.array(xs), np.array(xg), arrays, cid + "__", meta, full_graph=True)
        manifest.append(meta)
        print(cid, meta.get("vgoal"), meta.get("raises", ""), flush=True)
        return pl, meta

    # goal behind a wall, j < n  -> the reference faults (IndexError in pure Python)
    one("wall_unreachable_fault__std", "wall20", wall, 0, 30, 0, (3, 3), (15, 15))
    one("wall_unreachable_fault__star", "wall20", wall, 1, 30, 1, (3, 3), (15, 15), rr=6)
    # goal behind a wall, tree full (j == n) -> vgoal = 0: need every one of the first n-1 samples accepted
    for seed in range(200):
        pl = make_planner(ref, 0, wall, 3, seed, None, None)
        cap.reset()
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                T, gv = pl.plan(np.array((3, 3)), np.array((15, 15)))
        except IndexError:
            continue
        one("wall_unreachable_full__std", "wall20", wall, 0, 3, seed, (3, 3), (15, 15))
        break
    og = grids["square100"]
    one("start_eq_goal__std", "square100", og, 0, 50, 0, (10, 10), (10, 10))
    one("start_eq_goal__star", "square100", og, 1, 50, 0, (10, 10), (10, 10), rr=30)
    one("n1__std", "square100", og, 0, 1, 0, (10, 10), (12, 15))
    one("n1__star", "square100", og, 1, 1, 0, (10, 10), (12, 15), rr=10)
    one("n2__inf", "square100", og, 2, 2, 0, (10, 10), (12, 15), rr=10, rg=50)
    # informed where the very first insert already lies in the goal region, huge radii
    one("inf_bigradius", "empty43x100", grids["empty43x100"], 2, 300, 3, (5, 5), (30, 80), rr=500, rg=40)
    one("star_bigradius", "noise200", grids["noise200"], 1, 600, 4, tuple(pick_start_goal(grids["noise200"])[0]),
        tuple(pick_start_goal(grids["noise200"])[1]), rr=1000)
    one("star_zero_radius", "noise200", grids["noise200"], 1, 300, 4, tuple(pick_start_goal(grids["noise200"])[0]),
        tuple(pick_start_goal(grids["noise200"])[1]), rr=0)
    # start on an obstacle cell: every line of sight from the root is blocked at its first cell
    one("start_on_obstacle__std", "square100", og, 0, 40, 0, (50, 50), (10, 10), expect="fault")
    # tiny grid, many duplicate samples (rrt.py:425 `not in sampled`), sample == xstart allowed once
    tiny = np.zeros((6, 5), dtype=int)
    tiny[2, 2] = 1
    arrays["grid__tiny6x5"] = (tiny != 0).astype(np.uint8)
    one("tiny_dups__std", "tiny6x5", tiny, 0, 80, 0, (0, 0), (5, 4))
    one("tiny_dups__star", "tiny6x5", tiny, 1, 80, 1, (0, 0), (5, 4), rr=3)
    one("tiny_dups__inf", "tiny6x5", tiny, 2, 80, 2, (0, 0), (5, 4), rr=3, rg=2)

    # consecutive plans on one planner with set_og between (RNG stream continues)
    noise = grids["noise200"]
    noise2 = perlin_occupancygrid(200, 200, thresh=0.33, seed=2)
    arrays["grid__noise200b"] = (noise2 != 0).astype(np.uint8)
    for tag, alg, rr, rg in (("std", 0, None, None), ("star", 1, 32, None), ("inf", 2, 32, 12)):
        pl = make_planner(ref, alg, noise, 300, 5, rr, rg)
        xs, xg = pick_start_goal(noise)
        for step in range(3):
            if step == 2:
                pl.set_og(noise2)
                xs, xg = pick_start_goal(noise2)
            cid = f"replan__{tag}__step{step}"
            meta = dict(id=cid, grid="noise200b" if step == 2 else "noise200", alg=alg, r_rewire=rr, r_goal=rg, seed=5, n=300,
                        xstart=[int(xs[0]), int(xs[1])], xgoal=[int(xg[0]), int(xg[1])], chain=f"replan__{tag}", step=step)
            record_plan(cap, pl, xs, xg, arrays, cid + "__", meta, full_graph=False)
            manifest.append(meta)
            print(cid, meta.get("vgoal"), flush=True)
    arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "special_A.npz"), **arrays)
    set_policy(False)


class CellLogger:
    """Stands in for `og` inside the reference's collisionfree to record the cells it reads."""

    def __init__(self):
        self.cells = []

    def __getitem__(self, xy):
        self.cells.append((int(xy[0]), int(xy[1])))
        return 0


def gen_primitives(ref, grids):
    set_policy(False)
    rng = np.random.default_rng(123)
    out = {}
    # r2norm (the reference's own test: tests/test_rrt.py:68-71)
    v = rng.integers(-3000, 3000, size=(500, 2))
    out["r2norm_in"] = v
    out["r2norm_out"] = np.asarray([ref.r2norm(p) for p in v], dtype=np.float64)
    # collisionfree: every ordered pair of a 12x12 grid
    g12 = (rng.uniform(size=(12, 12)) < 0.18).astype(int)
    out["cf12_grid"] = g12.astype(np.uint8)
    pairs = [(a, b, c, d) for a in range(12) for b in range(12) for c in range(12) for d in range(12)]
    out["cf12_free"] = np.asarray([ref.RRT.collisionfree(g12, np.array((a, b)), np.array((c, d))) for a, b, c, d in pairs], dtype=np.uint8)
    # collisionfree on the noise grid: random segments
    ng = grids["noise200"]
    seg = rng.integers(0, 200, size=(4000, 4))
    out["cf200_seg"] = seg.astype(np.int32)
    out["cf200_free"] = np.asarray([ref.RRT.collisionfree(ng, s[:2], s[2:]) for s in seg], dtype=np.uint8)
    # the literal cell sequence of the walk (incl. long segments up to 2048)
    segs = np.concatenate([rng.integers(0, 64, size=(300, 4)), rng.integers(0, 2048, size=(60, 4)),
                           np.array([[0, 0, 2047, 2047], [2047, 0, 0, 2047], [5, 5, 5, 5], [0, 7, 2047, 8], [9, 2047, 8, 0],
                                     [100, 100, 101, 1100], [100, 100, 1100, 101], [3, 3, 3, 40], [40, 3, 3, 3]])])
    cells, offs = [], [0]
    for s in segs:
        lg = CellLogger()
        ok = ref.RRT.collisionfree(lg, np.array(s[:2]), np.array(s[2:]))
        assert ok
        cells.extend(lg.cells)
        offs.append(len(cells))
    out["walk_seg"] = segs.astype(np.int32)
    out["walk_cells"] = np.asarray(cells, dtype=np.int32)
    out["walk_offs"] = np.asarray(offs, dtype=np.int64)
    # within: strict <, int and float radius, plus the reference's own 4-corner case (tests/test_rrt.py:116-119)
    pts = rng.integers(0, 200, size=(3000, 2))
    xq = rng.integers(0, 200, size=(40, 2))
    for k, r in enumerate([0, 1, 5, 7.5, 10, 32, 64, 64.5, 1000]):
        cnt, sm = [], []
        for x in xq:
            w = ref.RRT.within(pts, x, r)
            cnt.append(len(w))
            sm.append(int(np.sum(w)))
        out[f"within_cnt_{k}"] = np.asarray(cnt, dtype=np.int32)
        out[f"within_sum_{k}"] = np.asarray(sm, dtype=np.int64)
    out["within_r"] = np.asarray([0, 1, 5, 7.5, 10, 32, 64, 64.5, 1000], dtype=np.float64)
    out["within_pts"] = pts.astype(np.int32)
    out["within_xq"] = xq.astype(np.int32)
    out["within_corner_count"] = np.asarray([ref.RRT.within(np.array([[0, 0], [1, 0], [1, 1], [0, 1]]), np.array([0.5, 0.5]), 1.0).shape[0]])
    # near()[0] under both policies on tie-rich integer data
    near_raw, near_stable = [], []
    for x in xq:
        near_raw.append(int(ref.RRT.near(pts, x)[0]))
    set_policy(True)
    for x in xq:
        near_stable.append(int(ref.RRT.near(pts, x)[0]))
    set_policy(False)
    out["near_raw"] = np.asarray(near_raw, dtype=np.int32)
    out["near_stable"] = np.asarray(near_stable, dtype=np.int32)
    # PCG64 interleave vector (free-space draw = bounded 32-bit, uniform = 64-bit)
    g = np.random.default_rng(0)
    F = 700001
    out["pcg_interleave"] = np.asarray([g.choice(F), g.uniform(0, 1), g.choice(F), g.choice(F), g.uniform(0, 1)], dtype=np.float64)
    # Informed sampler: rotation matrix and sample_ellipse table
    og = np.zeros((300, 200), dtype=int)
    inf = ref.RRTStarInformed(og, 10, 10, 5, pbar=False, seed=0)
    sg = np.array([[10, 10, 250, 150], [250, 150, 10, 10], [40, 180, 40, 20], [5, 100, 290, 100], [100, 5, 101, 190], [17, 23, 18, 24]])
    out["rot_sg"] = sg.astype(np.int32)
    out["rot_C"] = np.asarray([inf.rotation_to_world_frame(s[:2], s[2:]) for s in sg], dtype=np.float64)
    tab_in, tab_out = [], []
    fma_ok = True
    for s in sg:
        xs, xg = s[:2], s[2:]
        dmin = float(np.linalg.norm(xg - xs))
        for _ in range(400):
            c = dmin * rng.uniform(1.0, 2.5)
            u = rng.uniform(-1, 1, size=2)
            while u[0] * u[0] + u[1] * u[1] >= 1:
                u = rng.uniform(-1, 1, size=2)
            inf.unitball = lambda u=u: u
            xn = inf.sample_ellipse(xs, xg, c)
            tab_in.append([xs[0], xs[1], xg[0], xg[1], c, u[0], u[1]])
            tab_out.append([int(xn[0]), int(xn[1])])
    out["ell_in"] = np.asarray(tab_in, dtype=np.float64)
    out["ell_out"] = np.asarray(tab_out, dtype=np.int32)
    out["ell_WH"] = np.asarray([300, 200], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **out)


def main():
    ref = load_reference()
    grids = make_grids()
    gen_primitives(ref, grids)
    cap = Capture(ref)
    gen_special(ref, cap, grids)
    gen_plans(ref, cap, grids, stable=True, seeds=(0, 1, 2), ns=(25, 100, 400, 2000), out_name="plans_A.npz")
    gen_plans(ref, cap, {k: grids[k] for k in ("square100", "noise200")}, stable=False, seeds=(0,), ns=(100, 400),
              out_name="plans_B.npz")


if __name__ == "__main__":
    main()
