"""Opt-in true RRT* rewire (`rewire="correct"`, RRT_FLAG_REWIRE).  NOT the reference's behaviour --
its rewire never fires (rrt.py:532-536) -- so there is no reference parity here: the oracle's restatement
(oracle/rrt_oracle.c, "opt-in correct rewire") is checked for the properties that define a correct rewire, and the HIP
kernel is checked bit for bit against that oracle.  The default path must stay bit-identical to the reference goldens
(tests/test_gpu_parity.py, tests/test_oracle_golden.py run it unchanged)."""
import numpy as np
import pytest

import oracle
import orchelp
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd import rrt as amd
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair


def _query(grid, n, seed, gseed=1):
    og = perlin_occupancygrid(grid, grid, seed=gseed)
    og8 = oracle.og_u8(og)
    xs, xg = random_connected_pair(og, np.random.default_rng(7))
    rng = np.random.default_rng(seed)
    samples = hostprep.draw_free_samples(rng, np.argwhere(og8 == 0), n)
    return og, og8, xs, xg, samples, rng


def _check_tree(og8, r, xs):
    """Every property a rewired tree must have: rooted and acyclic, costs exactly parent cost + edge length, every edge free
    in the direction the rewire / choose-parent tested it (child-side endpoint last), costs non-decreasing along root paths."""
    live = r.j + (1 if r.found else 0)
    par = r.parent[:live].astype(np.int64)
    pts = r.pts[:live].astype(np.int64)
    assert par[0] == -1 and np.all(par[1:] >= 0) and np.all(par[1:] < r.j) and pts[0].tolist() == [int(xs[0]), int(xs[1])]
    depth = np.zeros(live, dtype=np.int64)
    order = np.argsort(r.vcost[:live], kind="stable")
    hops = np.full(live, -1)
    hops[0] = 0
    for _ in range(live):  # relax until every vertex has its hop count: terminates iff the parent map is a tree
        todo = hops < 0
        if not todo.any():
            break
        ok = todo & (hops[np.where(par >= 0, par, 0)] >= 0)
        assert ok.any(), "cycle or orphan in the parent map"
        hops[ok] = hops[par[ok]] + 1
    d = pts[1:] - pts[par[1:]]
    dist = np.sqrt((d * d).sum(1).astype(np.float64))
    assert np.array_equal(r.vcost[1:live], r.vcost[par[1:]] + dist), "a cost is not its parent's cost + the edge length"
    assert np.all(r.vcost[1:live] >= r.vcost[par[1:]])
    for c in range(1, live, max(1, live // 400)):
        assert oracle.collisionfree(og8, pts[par[c]], pts[c])[0] or oracle.collisionfree(og8, pts[c], pts[par[c]])[0]
    del depth, order


@pytest.mark.parametrize("alg,rr,rg,grid,n", [(1, 24, None, 160, 1500), (1, 64, None, 512, 6000), (2, 32, 10, 200, 2500)])
def test_oracle_rewire_is_a_correct_rewire(alg, rr, rg, grid, n):
    og, og8, xs, xg, samples, rng = _query(grid, n, 3)
    r2 = hostprep.radius_threshold(rr)
    kw = dict(r2_rewire=r2, r_goal=rg or 0.0, logs=False)
    if alg == 2:
        kw["Cmat"] = hostprep.rotation_to_world_frame(xs, xg)
    st0, r0 = oracle.plan(og8, n, alg, xs, xg, samples, **kw)
    st1, r1 = oracle.plan(og8, n, alg, xs, xg, samples, rewire=True, **kw)
    if st1 == oracle.ORC_NEED_UNITBALL:
        assert st0 == st1 and r0.i_switch == r1.i_switch  # up to the switch both modes sample the same stream
        ub = hostprep.draw_unitball(rng, n - r1.i_switch)
        st0, r0 = oracle.plan(og8, n, alg, xs, xg, samples, unitball=ub, ub_offset=r1.i_switch, **kw)
        st1, r1 = oracle.plan(og8, n, alg, xs, xg, samples, unitball=ub, ub_offset=r1.i_switch, rewire=True, **kw)
    assert st0 == 0 and st1 == 0 and r0.n_rewired == 0 and r1.n_rewired > 0 and r1.n_propagated > 0
    _check_tree(og8, r1, xs)
    if alg == 1:  # same sample stream, same acceptance: the vertex set is the reference's, only parents and costs differ
        assert r1.j == r0.j and np.array_equal(r1.pts[:r1.j], r0.pts[:r0.j])
        assert np.all(r1.vcost[:r1.j] <= r0.vcost[:r0.j] + 1e-9), "a rewire made some vertex more expensive"
        assert r1.vcost[:r1.j].mean() < r0.vcost[:r0.j].mean()
        # local optimality at the end is not implied, but no single rewire may be left over for the LAST inserted vertex
        v = r1.j - 1
        dd = r1.pts[:v].astype(np.int64) - r1.pts[v]
        d2 = (dd * dd).sum(1)
        for u in np.flatnonzero(d2 < r2):
            c = r1.vcost[v] + np.sqrt(float(d2[u]))
            assert not (c < r1.vcost[u] and oracle.collisionfree(og8, r1.pts[u], r1.pts[v])[0])


def test_planner_classes_take_the_rewire_option():
    og, og8, xs, xg, samples, _ = _query(120, 600, 0)
    with pytest.raises(ValueError):
        amd.RRTStar(og, 10, 5, rewire="yes")
    assert amd.RRTStar(og, 10, 5).rewire == "reference" and amd.RRTStarInformed(og, 10, 5, 3).rewire == "reference"
    a = orchelp.use_oracle(amd.RRTStar(og, 600, 20, pbar=False, seed=0))
    b = orchelp.use_oracle(amd.RRTStar(og, 600, 20, pbar=False, seed=0, rewire="correct"))
    Ta, ga = a.plan(xs, xg)
    Tb, gb = b.plan(xs, xg)
    assert a.last_stats["n_rewired"] == 0 and b.last_stats["n_rewired"] > 0
    assert Ta.number_of_nodes() == Tb.number_of_nodes() and Ta.number_of_edges() == Tb.number_of_edges()
    # the returned graph keeps the build_graph contract; root paths are still read off the parent pointers
    pa, pb = a.route2gv(Ta, ga), b.route2gv(Tb, gb)
    assert pa[0] == 0 and pb[0] == 0 and pa[-1] == ga and pb[-1] == gb
    cost = lambda T, path: sum(T.edges[u, v]["dist"] for u, v in zip(path[:-1], path[1:]))
    assert cost(Tb, pb) <= cost(Ta, pa) + 1e-9
    assert np.isclose(cost(Tb, pb), Tb.edges[pb[-2], pb[-1]]["cost"])


# ------------------------------------------------------------------------------------------------------------------ GPU
def _device_vs_oracle_rewire(ctx, og8, alg, n, seed_rng, xs, xg, samples, rr, rg):
    r2 = hostprep.radius_threshold(rr)
    gd2 = hostprep.goal_threshold(rg) if rg is not None else 0
    Cm = hostprep.rotation_to_world_frame(np.asarray(xs, dtype=np.int64), np.asarray(xg, dtype=np.int64)) if alg == 2 else None
    q, keep = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2, goal_d2=gd2, Cmat=Cm)
    rc, res = ctx.plan(q, n, logs=True, rewire=True)
    kw = dict(r2_rewire=r2, r_goal=rg or 0.0, Cmat=Cm, rewire=True)
    st, ro = oracle.plan(og8, n, alg, xs, xg, samples, **kw)
    if rc == _ffi.RRT_NEED_UNITBALL:
        assert st == oracle.ORC_NEED_UNITBALL and res.i_switch == ro.i_switch
        ub = hostprep.draw_unitball(seed_rng, n - res.i_switch)
        rc = ctx.plan_resume(ub, res)
        st, ro = oracle.plan(og8, n, alg, xs, xg, samples, unitball=ub, ub_offset=res.i_switch, **kw)
    assert rc == st
    live = ro.j + (1 if ro.found else 0)
    assert (res.j, res.found, res.vgoal, res.i_switch) == (ro.j, ro.found, ro.vgoal, ro.i_switch)
    assert (res.n_rewired, res.n_propagated) == (ro.n_rewired, ro.n_propagated)
    assert np.array_equal(res.nearest_log, ro.nearest_log) and np.array_equal(res.accept_log, ro.accept_log)
    assert np.array_equal(res.pts[:live], ro.pts[:live])
    assert np.array_equal(res.parent[:live], ro.parent[:live])
    assert np.array_equal(res.vcost[:live], ro.vcost[:live])
    assert np.array_equal(res.cbest_log, ro.cbest_log, equal_nan=True)
    return res, ro


@pytest.mark.gpu
@pytest.mark.parametrize("alg,rr,rg,grid,n,seed", [
    (1, 24, None, 160, 1500, 1), (1, 64, None, 1024, 20000, 0), (2, 64, 12, 1024, 20000, 0), (2, 30, 8, 96, 2500, 2),
    (1, 9, None, 64, 3000, 1), (1, 1e6, None, 200, 3000, 3), (1, 64, None, 2048, 12000, 5)])
def test_device_rewire_equals_the_oracle(gpu_ctx, alg, rr, rg, grid, n, seed):
    og, og8, xs, xg, samples, rng = _query(grid, n, seed, gseed=1 if grid != 2048 else 3)
    gpu_ctx.set_grid(og8)
    res, ro = _device_vs_oracle_rewire(gpu_ctx, og8, alg, n, rng, xs, xg, samples, rr, rg)
    assert res.n_rewired > 0
    _check_tree(og8, res, xs)


@pytest.mark.gpu
def test_device_rewire_config2_full_size_and_planner_class(gpu_ctx):
    """BASELINE config 2's query with the opt-in rewire: whole tree equal to the oracle's, through the planner class."""
    og, og8, xs, xg, samples, _ = _query(1024, 50000, 0)
    p = amd.RRTStar(og, 50000, 64, pbar=False, seed=0, rewire="correct")
    res = p._run(_ffi.ALG_STAR, xs, xg, r_rewire=64, rewire=True)
    st, ro = oracle.plan(og8, 50000, 1, xs, xg, samples, r2_rewire=hostprep.radius_threshold(64), logs=False, rewire=True)
    live = ro.j + 1
    assert st == 0 and res.j == ro.j and res.vgoal == ro.vgoal and res.n_rewired == ro.n_rewired and res.n_propagated == ro.n_propagated
    assert np.array_equal(res.parent[:live], ro.parent[:live]) and np.array_equal(res.vcost[:live], ro.vcost[:live])
    # and the default mode on the same planner class is untouched by the option's existence
    q = amd.RRTStar(og, 3000, 64, pbar=False, seed=0)
    r0 = q._run(_ffi.ALG_STAR, xs, xg, r_rewire=64)
    st, o0 = oracle.plan(og8, 3000, 1, xs, xg, samples[:3000], r2_rewire=hostprep.radius_threshold(64), logs=False)
    assert r0.n_rewired == 0 and np.array_equal(r0.parent[:o0.j + 1], o0.parent[:o0.j + 1])


@pytest.mark.gpu
def test_device_rewire_fuzz_small(gpu_ctx):
    rng = np.random.default_rng(77)
    for case in range(150):
        w, h = int(rng.integers(8, 60)), int(rng.integers(8, 60))
        og8 = (rng.uniform(size=(w, h)) < rng.choice([0.0, 0.15, 0.35])).astype(np.uint8)
        free = np.argwhere(og8 == 0)
        if free.shape[0] < 2:
            continue
        alg = int(rng.integers(1, 3))
        n = int(rng.choice([1, 2, 17, 64, 100, 300, 900]))
        rr = float(rng.choice([1.5, 3, 8, 20, 500]))
        rg = float(rng.choice([1, 3, 10]))
        xs, xg = free[rng.integers(0, free.shape[0])], free[rng.integers(0, free.shape[0])]
        if alg == 2 and (xs == xg).all():
            continue
        gpu_ctx.set_grid(og8)
        srng = np.random.default_rng(case)
        samples = hostprep.draw_free_samples(srng, free, n)
        try:
            _device_vs_oracle_rewire(gpu_ctx, og8, alg, n, srng, xs, xg, samples, rr, rg if alg == 2 else None)
        except AssertionError as e:
            raise AssertionError(f"rewire fuzz case {case}: {w}x{h} alg {alg} n {n} r {rr} rg {rg} xs {xs} xg {xg}") from e
