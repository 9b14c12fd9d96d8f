"""Test helpers: golden-fixture access and an oracle-backed stand-in for the device context.

`OracleCtx` implements the two methods of rrtplanner_amd._ffi.Context that RRT._run uses
(plan / plan_resume) on top of the CPU oracle, so the host logic of the planner classes (RNG
stream handling, thresholds, DiGraph materialisation) can be tested without a GPU.  It lives
under tests/ -- the product never sees it.
"""
import ctypes as C
import json
import os

import numpy as np

import oracle
from rrtplanner_amd import _ffi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INT32_MIN = np.iinfo(np.int32).min
INT64_MIN = np.iinfo(np.int64).min


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLD, name))
        self.manifest = json.loads(bytes(self.z["manifest"]).decode()) if "manifest" in self.z else []
        self.by_id = {m["id"]: m for m in self.manifest}

    def grid(self, name):
        return self.z["grid__" + name]

    def arr(self, cid, key):
        return self.z[f"{cid}__{key}"]

    def has(self, cid, key):
        return f"{cid}__{key}" in self.z


_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


def query_fields(query):
    n = query.n
    if query.samples_packed:
        pk = np.ctypeslib.as_array(C.cast(query.samples_packed, C.POINTER(C.c_uint32)), shape=(n,))
        samples = np.stack([pk & 0xffff, pk >> 16], axis=1).astype(np.int32)
    else:
        samples = np.ctypeslib.as_array(C.cast(query.samples, C.POINTER(C.c_int32)), shape=(n, 2)).copy()
    return dict(alg=query.alg, n=n, xs=(query.xs[0], query.xs[1]), xg=(query.xg[0], query.xg[1]),
                r2_rewire=query.r2_rewire, goal_d2=query.goal_d2, samples=samples,
                Cmat=np.array([query.C[k] for k in range(4)]))


def goal_d2_to_r(goal_d2):
    """The oracle compares sqrt(d2) < r_goal; any r with ceil-threshold goal_d2 is equivalent:
    sqrt(goal_d2 - 1) < r <= sqrt(goal_d2).  Use r = sqrt(goal_d2) (0 -> 0)."""
    return float(np.sqrt(float(goal_d2))) if goal_d2 > 0 else 0.0


class OracleCtx:
    """Context stand-in backed by oracle.plan (CPU)."""

    def __init__(self, og8):
        self.og8 = np.ascontiguousarray(og8, dtype=np.uint8)
        self._q = None

    def set_grid(self, og8):
        self.og8 = np.ascontiguousarray(og8, dtype=np.uint8)

    def _fill(self, res, st, r):
        n = res.n
        live = r.j + (1 if r.found else 0)
        res.pts[:live] = r.pts[:live]
        res.vcost[:live] = r.vcost[:live]
        res.parent[:live] = r.parent[:live]
        for k in ("j", "vgoal", "found", "i_switch", "rows", "sum_j", "sum_cells_nn", "sum_near", "sum_cells_cand", "n_rewired",
                  "n_propagated"):
            setattr(res.c, k, int(getattr(r, k)))
        res.c.status = st
        if hasattr(res, "nearest_log"):
            res.nearest_log[:] = r.nearest_log
            res.accept_log[:] = r.accept_log
            res.cbest_log[:] = r.cbest_log
            res.j_log[:] = r.jlog
        assert n == r.pts.shape[0] - 1

    def plan(self, query, n, logs=False, rewire=False):
        if query.alg >= _ffi.ALG_DUBINS:
            return self._plan_dubins(query, n, logs)
        f = query_fields(query)
        f["rewire"] = rewire
        self._q = f
        st, r = oracle.plan(self.og8, n, f["alg"], f["xs"], f["xg"], f["samples"], r2_rewire=f["r2_rewire"],
                            r_goal=goal_d2_to_r(f["goal_d2"]), Cmat=f["Cmat"], rewire=rewire)
        res = _ffi.ResultArrays(n, logs)
        self._fill(res, st, r)
        return st, res

    def _plan_dubins(self, query, n, logs):
        samples = np.ctypeslib.as_array(C.cast(query.samples, C.POINTER(C.c_int32)), shape=(n, 2)).copy()
        heads = np.ctypeslib.as_array(C.cast(query.headings, C.POINTER(C.c_int32)), shape=(n,)).copy()
        st, r = oracle.dubins_plan(self.og8, n, query.alg == _ffi.ALG_DUBINS_STAR, (query.xs[0], query.xs[1], query.hs),
                                   (query.xg[0], query.xg[1], query.hg), samples, heads, r2_rewire=query.r2_rewire, rho=query.rho, nh=query.nh)
        res = _ffi.ResultArrays(n, logs, headings=True)
        live = r.j + (1 if r.found else 0)
        res.pts[:live], res.vcost[:live], res.parent[:live], res.head[:live] = r.pts[:live], r.vcost[:live], r.parent[:live], r.head[:live]
        for k in ("j", "vgoal", "found", "rows", "sum_j", "sum_cells_nn", "sum_near", "sum_cells_cand"):
            setattr(res.c, k, int(getattr(r, k)))
        res.c.status = st
        if logs:
            res.nearest_log[:], res.accept_log[:] = r.nearest_log, r.accept_log
        return st, res

    def plan_resume(self, unitball, res):
        f = self._q
        st, r = oracle.plan(self.og8, f["n"], f["alg"], f["xs"], f["xg"], f["samples"], r2_rewire=f["r2_rewire"],
                            r_goal=goal_d2_to_r(f["goal_d2"]), unitball=unitball, ub_offset=res.i_switch, Cmat=f["Cmat"],
                            rewire=f["rewire"])
        self._fill(res, st, r)
        return st


def make_planner(mod, meta, og, device_ctx=None, costfn=None):
    """Planner of `mod` (rrtplanner_amd.rrt) for a golden manifest entry."""
    alg, n, seed = meta["alg"], meta["n"], meta["seed"]
    if alg == 0:
        p = mod.RRTStandard(og, n, costfn=costfn, pbar=False, seed=seed)
    elif alg == 1:
        p = mod.RRTStar(og, n, meta["r_rewire"], costfn=costfn, pbar=False, seed=seed)
    else:
        p = mod.RRTStarInformed(og, n, meta["r_rewire"], meta["r_goal"], costfn=costfn, pbar=False, seed=seed)
    return p


class NumpyProvider:
    """Stand-in for the device primitives of a host-driven planner (rrtplanner_amd/hostloop.py: rrt_tree_query /
    rrt_prim_collisionfree), so that the loop itself can be checked against the reference's goldens without a GPU.  numpy for
    near()[0] / within(), the C oracle's line walk for collisionfree."""

    def __init__(self, og8):
        self.og8 = np.ascontiguousarray(og8, dtype=np.uint8)
        self.pts = np.zeros((0, 2), dtype=np.int64)
        self.j = 0
        self.queries = 0

    def reset(self):
        self.pts = np.zeros((1024, 2), dtype=np.int64)
        self.j = 0

    def append(self, x, y):
        if self.j == self.pts.shape[0]:
            self.pts = np.concatenate([self.pts, np.zeros_like(self.pts)])
        self.pts[self.j] = (x, y)
        self.j += 1
        return self.j - 1

    def query(self, x, y, r2):
        self.queries += 1
        p = self.pts[:self.j]
        d = p - np.array((x, y))
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        nearest = int(np.argmin(d2))  # first minimum: lowest index among equal distance
        idx = np.flatnonzero(d2 < r2).astype(np.int32)
        los = np.array([oracle.collisionfree(self.og8, p[k], (x, y))[0] for k in idx], dtype=bool)
        return nearest, idx, bool(oracle.collisionfree(self.og8, p[nearest], (x, y))[0]), los

    def collisionfree_many(self, ab):
        return np.array([oracle.collisionfree(self.og8, s[:2], s[2:])[0] for s in np.asarray(ab).reshape(-1, 4)], dtype=bool)

    def close(self):
        pass


def use_oracle(planner):
    """Route planner._device() to the oracle stand-in (tests only)."""
    ctx = OracleCtx(oracle.og_u8(planner.og))

    def _device():
        if planner._grid_dirty:
            ctx.set_grid(oracle.og_u8(planner.og))
            planner._grid_dirty = False
        return ctx

    planner._device = _device
    return planner


def rng_state_tuple(gen):
    s = gen.bit_generator.state
    return [str(s["state"]["state"]), str(s["state"]["inc"]), int(s["has_uint32"]), int(s["uinteger"])]


def check_plan_against_golden(G, meta, planner, T, gv, check_graph=True):
    """Compare a (T, gv) returned by a planner with the reference's golden record."""
    cid = meta["id"]
    pts = G.arr(cid, "pts").astype(np.int64)
    pts[pts == INT32_MIN] = INT64_MIN
    vcost = G.arr(cid, "vcost")
    parent = G.arr(cid, "parent")
    rows = meta["rows"]
    assert int(gv) == meta["vgoal"]
    assert T.number_of_nodes() == meta["n_nodes"]
    assert T.number_of_edges() == meta["n_edges"]
    # node table
    got_pts = np.array([T.nodes[i]["pt"] for i in range(rows)], dtype=np.int64)
    assert np.array_equal(got_pts, pts), "tree node coordinates differ from the reference"
    # topology: bit-exact parents; edge costs within 1e-6 (they are in fact identical)
    got_parent = np.full(rows, -1, dtype=np.int64)
    got_cost = np.full(rows, np.inf)
    got_cost[0] = 0.0
    for u, v, d in T.edges(data=True):
        got_parent[v] = u
        got_cost[v] = d["cost"]
    assert np.array_equal(got_parent, parent.astype(np.int64)), "tree topology differs from the reference"
    live = got_parent >= 0
    assert np.allclose(got_cost[live], vcost[live], rtol=0, atol=1e-6)
    assert np.array_equal(got_cost[live], vcost[live]), "edge costs are not bit-identical"
    if meta.get("route_raises"):  # (custom cost functions: the reference's rewire block can cut the goal off the root)
        import networkx as nx
        import pytest

        with pytest.raises(getattr(nx, meta["route_raises"])):
            planner.route2gv(T, gv)
        path = []
    else:
        path = planner.route2gv(T, gv)
    assert [int(v) for v in path] == G.arr(cid, "path").tolist()
    assert rng_state_tuple(planner.rand_gen) == meta["rng_state"], "generator state after plan() differs"
    if check_graph and G.has(cid, "node_order"):
        assert [int(v) for v in T.nodes] == G.arr(cid, "node_order").tolist()
        ed = list(T.edges(data=True))
        assert [[int(u), int(v)] for u, v, _ in ed] == G.arr(cid, "edge_uv").tolist()
        assert np.array_equal(np.array([d["dist"] for _, _, d in ed]), G.arr(cid, "edge_dist"))
        assert np.array_equal(np.array([d["cost"] for _, _, d in ed]), G.arr(cid, "edge_cost"))
        if ed:
            assert type(ed[0][2]["dist"]).__name__ == meta["edge_dist_type"]
            assert type(ed[0][2]["cost"]).__name__ == meta["edge_cost_type"]
        assert str(T.nodes[0]["pt"].dtype) == meta["pt_dtype"]
        assert np.array_equal(planner.vertices_as_ndarray(T, path).reshape(-1, 2, 2), G.arr(cid, "path_pts"))
    if G.has(cid, "ell_keys"):
        keys = list(planner.ellipses.keys())
        assert keys == G.arr(cid, "ell_keys").tolist()
        vals = np.array([[planner.ellipses[k][0][0], planner.ellipses[k][0][1], planner.ellipses[k][1],
                          planner.ellipses[k][2], planner.ellipses[k][3]] for k in keys]).reshape(-1, 5)
        assert np.allclose(vals, G.arr(cid, "ell_vals"), rtol=1e-12, atol=1e-9)
