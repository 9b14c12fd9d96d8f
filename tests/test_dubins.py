"""Dubins-vehicle RRT / RRT* (BASELINE.json configs[4]).  NO REFERENCE PARITY: the reference only advertises these planners
(README.md:12,18-19), so three things are checked instead --
  1. the geometry of include/rrt_dubins.h (fixed-order arithmetic shared by oracle and kernel) against numpy / libm: every
     reported word, integrated forward, ends in the goal pose; lengths equal an independent numpy implementation
     (rrtplanner_amd/dubins.py) and obey the symmetries of shortest Dubins paths;
  2. the oracle's sequential loop (oracle/dubins_oracle.c) for the properties that define the planner;
  3. the HIP kernel bit for bit against that oracle (GPU)."""
import math
import os

import numpy as np
import pytest

import oracle
import orchelp
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd import dubins as dub
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

KIND = {0: (1, 0, 1), 1: (1, 0, -1), 2: (-1, 0, 1), 3: (-1, 0, -1), 4: (-1, 1, -1), 5: (1, -1, 1)}


def _integrate(x, y, th, word, t, p, q, rho):
    for kind, tau in zip(KIND[word], (t, p, q)):
        if kind == 0:
            x, y = x + rho * tau * math.cos(th), y + rho * tau * math.sin(th)
        else:
            x += kind * rho * (math.sin(th + kind * tau) - math.sin(th))
            y -= kind * rho * (math.cos(th + kind * tau) - math.cos(th))
            th += kind * tau
    return x, y, th


def test_elementary_functions_match_libm():
    rng = np.random.default_rng(0)
    for a in np.concatenate([rng.uniform(-60, 60, 4000), np.linspace(-7, 7, 701)]):
        s, c = oracle.dub_sincos(a)
        assert abs(s - math.sin(a)) < 4e-16 and abs(c - math.cos(a)) < 4e-16
    for _ in range(4000):
        y, x = rng.normal(size=2) * rng.choice([1e-3, 1, 1e3])
        assert abs(oracle.dub_atan2(y, x) - math.atan2(y, x)) < 1e-15
    for y, x in [(0, 1), (0, -1), (1, 0), (-1, 0), (0, 0), (1, 1), (-1, -1), (1e-300, 1), (-2.0, 0.0)]:
        assert abs(oracle.dub_atan2(y, x) - math.atan2(y, x)) < 1e-15


def test_every_word_reaches_the_goal_pose_and_lengths_are_symmetric():
    rng = np.random.default_rng(1)
    seen = set()
    for _ in range(20000):
        rho = float(rng.choice([1.0, 4.0, 8.0, 13.7]))
        sc = float(rng.choice([0.5, 2, 6, 40])) * rho
        x0, y0, x1, y1 = rng.uniform(-sc, sc, 4)
        th0, th1 = rng.uniform(0, 2 * math.pi, 2)
        t, p, q, L, w = oracle.dub_shortest(x0, y0, th0, x1, y1, th1, rho)
        assert w < 6
        seen.add(w)
        xe, ye, the = _integrate(x0, y0, th0, w, t, p, q, rho)
        assert max(abs(xe - x1), abs(ye - y1), abs(math.remainder(the - th1, 2 * math.pi))) < 1e-9
        assert L >= math.hypot(x1 - x0, y1 - y0) - 1e-9 and abs(L - rho * (t + p + q)) < 1e-12 * max(1, L)
        assert abs(L - oracle.dub_shortest(x1, y1, th1 + math.pi, x0, y0, th0 + math.pi, rho)[3]) < 1e-7 * max(1, L)  # driven backwards
        assert abs(L - oracle.dub_shortest(x0, -y0, -th0, x1, -y1, -th1, rho)[3]) < 1e-7 * max(1, L)               # mirrored
    assert seen == {0, 1, 2, 3, 4, 5}


def test_header_agrees_with_the_independent_numpy_implementation():
    rng = np.random.default_rng(2)
    m = 5000
    rho = 7.5
    P = rng.uniform(-60, 60, size=(m, 4))
    th = rng.integers(0, 64, size=(m, 2)) * (2 * math.pi / 64)
    t, p, q, L, w = dub.dubins_shortest(P[:, 0], P[:, 1], th[:, 0], P[:, 2], P[:, 3], th[:, 1], rho)
    for k in range(m):
        t2, p2, q2, L2, w2 = oracle.dub_shortest(P[k, 0], P[k, 1], th[k, 0], P[k, 2], P[k, 3], th[k, 1], rho)
        assert abs(L[k] - L2) < 1e-9 * max(1.0, L2)
        if w[k] != w2:  # only when two words tie to rounding
            assert abs(L[k] - L2) < 1e-9


def test_sweep_cells_lie_on_the_numpy_polyline():
    rng = np.random.default_rng(3)
    for _ in range(300):
        rho, nh = float(rng.choice([3.0, 8.0])), 64
        a = (int(rng.integers(0, 200)), int(rng.integers(0, 200)), int(rng.integers(0, nh)))
        b = (int(rng.integers(0, 200)), int(rng.integers(0, 200)), int(rng.integers(0, nh)))
        cells = oracle.dub_sweep_cells(a[0], a[1], 2 * math.pi * a[2] / nh, b[0], b[1], 2 * math.pi * b[2] / nh, rho)
        poly = dub.dubins_polyline(a, b, rho, nh, ds=0.5)
        assert len(cells) == len(poly) - 1 or len(cells) == len(poly)  # the polyline appends the end point
        pts = poly[:len(cells)]
        off = np.abs(pts - cells)  # a sample lies in the cell it rounds to: within half a cell (+ rounding noise at the boundary)
        assert np.all(off <= 0.5 + 1e-6)
        assert cells[0].tolist() == [a[0], a[1]] and np.allclose(poly[-1], b[:2], atol=1e-8)


def _dub_query(grid, n, seed, gseed=1):
    og = perlin_occupancygrid(grid, grid, seed=gseed)
    og8 = oracle.og_u8(og)
    xs, xg = random_connected_pair(og, np.random.default_rng(11))
    rng = np.random.default_rng(seed)
    samples = hostprep.draw_free_samples(rng, np.argwhere(og8 == 0), n)
    heads = rng.integers(0, 64, size=n)
    return og, og8, (int(xs[0]), int(xs[1]), 5), (int(xg[0]), int(xg[1]), 20), samples, heads


def _independent_sweep(og8, a, b, rho, nh):
    """The collision sweep of the edge a -> b WITHOUT include/rrt_dubins.h: the polyline of rrtplanner_amd/dubins.py (numpy /
    libm, its own word construction) sampled every DUB_DS = 0.5 cells of arc length, a sample occupying the cell round-half-up
    of its coordinates.  Returns (blocked, ambiguous) per sample: blocked = outside the grid or on an obstacle; ambiguous = a
    coordinate within 1e-7 of a cell boundary (libm and the fixed-order arithmetic may then name neighbouring cells)."""
    f = dub.dubins_polyline(a, b, rho, nh, ds=0.5) + 0.5
    cells = np.floor(f).astype(np.int64)
    amb = np.abs(f - np.round(f)).min(axis=1) < 1e-7
    inside = (cells[:, 0] >= 0) & (cells[:, 0] < og8.shape[0]) & (cells[:, 1] >= 0) & (cells[:, 1] < og8.shape[1])
    blocked = ~inside
    blocked[inside] = og8[cells[inside, 0], cells[inside, 1]] != 0
    return blocked, amb


def _independent_collision_witness(og8, res, samples, heads, rho, nh, count=300):
    """The device's collision DECISIONS against the independent sweep above (the oracle shares the kernel's geometry header, so
    HIP == oracle says nothing about a wrong formula in it): every edge of the device's tree is free, and every iteration the
    device rejected although its cell was new has a blocked (or boundary-ambiguous) sample on the word nearest -> sample."""
    live = res.j + (1 if res.found else 0)
    par, pts, hd = res.parent[:live].astype(np.int64), res.pts[:live].astype(np.int64), res.head[:live].astype(np.int64)
    for c in range(1, live, max(1, live // count)):
        p = par[c]
        blocked, amb = _independent_sweep(og8, (pts[p, 0], pts[p, 1], hd[p]), (pts[c, 0], pts[c, 1], hd[c]), rho, nh)
        assert not np.any(blocked & ~amb), f"edge {p} -> {c} of the device's tree crosses an obstacle"
    first_index = {}
    for k in range(res.j - 1, 0, -1):
        first_index[(int(pts[k, 0]), int(pts[k, 1]))] = k
    rej = np.flatnonzero(res.accept_log == 0)
    seen = 0
    for i in rej[:: max(1, rej.size // count)]:
        x, y, h, jtop = int(samples[i, 0]), int(samples[i, 1]), int(heads[i]), int(res.j_log[i])
        if first_index.get((x, y), jtop) < jtop or jtop >= res.n:
            continue  # rejected as a duplicate of an earlier vertex (rrt.py:425) or because the tree is full
        v = int(res.nearest_log[i])
        blocked, amb = _independent_sweep(og8, (pts[v, 0], pts[v, 1], hd[v]), (x, y, h), rho, nh)
        assert np.any(blocked | amb), f"iteration {i}: rejected, but the word from vertex {v} to the sample is free"
        seen += 1
    return seen


def _check_dubins_tree(og8, r, rho, nh, xs):
    live = r.j + (1 if r.found else 0)
    par, pts, hd = r.parent[:live].astype(np.int64), r.pts[:live].astype(np.int64), r.head[:live].astype(np.int64)
    assert par[0] == -1 and np.all(par[1:] >= 0) and np.all(par[1:] < np.arange(1, live)) and (pts[0, 0], pts[0, 1], hd[0]) == tuple(xs)
    assert np.all(og8[pts[:, 0], pts[:, 1]] == 0) and np.all((0 <= hd) & (hd < nh))
    assert len({(a, b) for a, b in pts[:r.j].tolist()}) >= r.j - 1  # one node per cell (xstart may repeat once, rrt.py:425)
    L = dub.dubins_shortest(pts[par[1:], 0], pts[par[1:], 1], 2 * math.pi * hd[par[1:]] / nh, pts[1:, 0], pts[1:, 1], 2 * math.pi * hd[1:] / nh, rho)[3]
    assert np.allclose(r.vcost[1:live], r.vcost[par[1:]] + L, rtol=0, atol=1e-8)
    for c in range(1, live, max(1, live // 300)):  # every edge's sweep is free
        p = par[c]
        cells = oracle.dub_sweep_cells(pts[p, 0], pts[p, 1], 2 * math.pi * hd[p] / nh, pts[c, 0], pts[c, 1], 2 * math.pi * hd[c] / nh, rho)
        assert np.all((cells >= 0) & (cells < og8.shape[0])) and np.all(og8[cells[:, 0], cells[:, 1]] == 0)


@pytest.mark.parametrize("star", [0, 1])
def test_oracle_dubins_plan_properties(star):
    og, og8, xs, xg, samples, heads = _dub_query(300, 4000, 0)
    st, r = oracle.dubins_plan(og8, 4000, star, xs, xg, samples, heads, r2_rewire=hostprep.radius_threshold(40), rho=6.0, nh=64)
    assert st == 0 and r.found and r.j > 1000
    _check_dubins_tree(og8, r, 6.0, 64, xs)
    st0, r0 = oracle.dubins_plan(og8, 4000, 0, xs, xg, samples, heads, rho=6.0, nh=64)
    assert r.j == r0.j and np.array_equal(r.pts[:r.j], r0.pts[:r0.j])  # acceptance does not depend on choose-parent
    if star:
        assert np.all(r.vcost[:r.j] <= r0.vcost[:r0.j] + 1e-9) and r.vcost[:r.j].mean() < r0.vcost[:r0.j].mean()


def test_planner_classes_on_the_oracle_stand_in():
    og, og8, xs, xg, _, _ = _dub_query(160, 700, 0)
    with pytest.raises(ValueError):
        dub.RRTStarDubins(og, 10, 5, rho=0.0)
    p = orchelp.use_oracle(dub.RRTStarDubins(og, 700, 30, rho=5.0, n_headings=32, pbar=False, seed=4))
    with pytest.raises(ValueError):
        p.plan(np.array([1, 2]), np.array(xg))  # not a pose
    with pytest.raises(ValueError):
        p.plan(np.array(xs), np.array([xg[0], xg[1], 32]))  # heading index outside [0, 32)
    T, gv = p.plan(np.array(xs), np.array(xg))
    assert T.number_of_nodes() == 701 and gv == p.last_stats["j"]
    path = p.route2gv(T, gv)
    assert path[0] == 0 and path[-1] == gv and T.nodes[gv]["heading"] == xg[2] and T.nodes[gv]["pt"].tolist() == list(xg[:2])
    d = [T.edges[u, v]["dist"] for u, v in zip(path[:-1], path[1:])]
    assert np.isclose(sum(d), T.edges[path[-2], path[-1]]["cost"], atol=1e-8)
    poly = p.path_points(T, path, ds=0.5)
    assert np.allclose(poly[0], xs[:2]) and np.allclose(poly[-1], xg[:2], atol=1e-7)
    seg = np.hypot(*np.diff(poly, axis=0).T)
    assert np.all(seg <= 0.5 + 1e-9) and np.isclose(seg.sum(), sum(d), rtol=2e-2)  # chords of arcs: slightly shorter than the arcs
    cells = np.floor(poly + 0.5).astype(int)
    assert np.all(og8[cells[:, 0], cells[:, 1]] == 0)
    q = orchelp.use_oracle(dub.RRTDubins(og, 300, rho=5.0, n_headings=32, pbar=False, seed=4))
    T2, g2 = q.plan(np.array(xs), np.array(xg))
    assert T2.number_of_nodes() == 301
    s = q.sample_all_free()
    assert s.shape == (3,) and og[s[0], s[1]] == 0 and 0 <= s[2] < 32


def _assert_audit_clean(a, star):
    """What oracle/dubins_ref.c's second opinion must say about a tree (its head comment): every accept / reject and every
    chosen parent either equal to its own, or within the stated tolerance / ambiguity -- never plainly different."""
    assert a["nearest_mismatch"] == 0 and a["accept_mismatch"] == 0, a
    assert a["parent_wrong"] == 0 and a["parent_blocked"] == 0 and a["cost_mismatch"] == 0, a
    assert a["max_cost_err"] < 1e-8, a
    assert a["parent_is_argmin"] + a["parent_within_tol"] == a["n_accepted"], a
    # the tolerance classes are the exception, not the rule
    assert a["parent_within_tol"] <= max(3, a["n_accepted"] // 1000) and a["accept_ambiguous"] + a["parent_blocked_ambiguous"] <= max(3, a["n_accepted"] // 1000), a


def test_independent_reference_words_agree_with_the_shared_header():
    """dubins_ref.c (libm, textbook closed forms, no shared header) against include/rrt_dubins.h through the oracle, on random
    pose pairs: same word, lengths and segments to 1e-9."""
    rng = np.random.default_rng(3)
    for _ in range(4000):
        x0, y0, x1, y1 = rng.uniform(0, 200, 4)
        th0, th1 = 2 * math.pi * rng.integers(0, 64, 2) / 64
        rho = float(rng.choice([2.0, 6.0, 8.0, 25.0]))
        a, b = oracle.dubref_shortest(x0, y0, th0, x1, y1, th1, rho), oracle.dub_shortest(x0, y0, th0, x1, y1, th1, rho)
        assert abs(a[3] - b[3]) < 1e-9 * (1 + b[3])
        if a[4] == b[4]:
            assert np.allclose(a[:3], b[:3], atol=1e-8)
        else:  # two words of (numerically) the same length
            assert abs(a[3] - b[3]) < 1e-9


@pytest.mark.parametrize("star,grid,n,rr,rho,seed", [(0, 300, 4000, None, 6.0, 0), (1, 300, 4000, 40, 6.0, 0), (1, 128, 1500, 20, 25.0, 6), (1, 64, 2500, 12, 2.0, 2)])
def test_independent_audit_of_the_oracles_tree(star, grid, n, rr, rho, seed):
    """the oracle and the kernel share the geometry header, so their equality cannot see a wrong formula in it.
    oracle/dubins_ref.c does not include it; its audit recomputes every accept / reject and every choose-parent decision behind
    the oracle's tree with libm arithmetic."""
    og, og8, xs, xg, samples, heads = _dub_query(grid, n, seed)
    r2 = hostprep.radius_threshold(rr) if star else 0
    st, r = oracle.dubins_plan(og8, n, star, xs, xg, samples, heads, r2_rewire=r2, rho=rho, nh=64)
    a = oracle.dubins_audit(og8, n, star, samples, heads, r.pts, r.head, r.vcost, r.parent, r.j, r2_rewire=r2, rho=rho, nh=64)
    assert a["n_accepted"] == r.j - 1
    _assert_audit_clean(a, star)
    # and the audit is not blind: a tree with one wrong parent / one wrong cost is caught
    if star and r.j > 50:
        bad = r.parent.copy()
        k = int(np.argmax(r.parent[1:r.j] != np.asarray(r.nearest_log)[np.flatnonzero(r.accept_log)][: r.j - 1])) + 1  # a vertex whose parent is not its nearest
        bad[k] = 0 if r.parent[k] != 0 else 1
        b = oracle.dubins_audit(og8, n, star, samples, heads, r.pts, r.head, r.vcost, bad, r.j, r2_rewire=r2, rho=rho, nh=64)
        assert b["parent_wrong"] + b["cost_mismatch"] + b["parent_blocked"] + b["nearest_mismatch"] >= 1
        cost = r.vcost.copy()
        cost[r.j // 2] += 1e-6
        c = oracle.dubins_audit(og8, n, star, samples, heads, r.pts, r.head, cost, r.parent, r.j, r2_rewire=r2, rho=rho, nh=64)
        assert c["cost_mismatch"] >= 1


# ------------------------------------------------------------------------------------------------------------------ GPU
def _device_vs_oracle_dubins(ctx, og8, star, n, xs, xg, samples, heads, rr, rho, nh=64, serial=False, counters=False):
    """serial = False: the 16-samples-per-round kernel (rrt_dubins_block.h, the default); True: the one-sample-per-iteration
    kernel (RRT_FLAG_SERIAL), kept as a second implementation of the same semantics."""
    r2 = hostprep.radius_threshold(rr) if star else 0
    q, keep = _ffi.make_query(_ffi.ALG_DUBINS_STAR if star else _ffi.ALG_DUBINS, n, xs, xg, samples, r2_rewire=r2, headings=heads, rho=rho, nh=nh)
    rc, res = ctx.plan(q, n, logs=True, serial=serial)
    st, ro = oracle.dubins_plan(og8, n, star, xs, xg, samples, heads, r2_rewire=r2, rho=rho, nh=nh, counters=counters)
    assert rc == st
    live = ro.j + (1 if ro.found else 0)
    assert (res.j, res.found, res.vgoal) == (ro.j, ro.found, ro.vgoal)
    assert np.array_equal(res.nearest_log, ro.nearest_log) and np.array_equal(res.accept_log, ro.accept_log)
    assert np.array_equal(res.pts[:live], ro.pts[:live]) and np.array_equal(res.head[:live], ro.head[:live])
    assert np.array_equal(res.parent[:live], ro.parent[:live])
    assert np.array_equal(res.vcost[:live], ro.vcost[:live])  # bit-exact: one arithmetic on both sides
    assert res.sum_j == ro.sum_j and res.sum_cells_nn == ro.sum_cells_nn and res.sum_near == ro.sum_near
    return res, ro


@pytest.mark.gpu
@pytest.mark.parametrize("star,grid,n,rr,rho,seed", [
    (0, 300, 4000, None, 6.0, 0), (1, 300, 4000, 40, 6.0, 0), (1, 1024, 20000, 64, 8.0, 1), (1, 64, 2500, 12, 2.0, 2),
    (1, 200, 3000, 1e6, 4.0, 3), (0, 2048, 30000, None, 8.0, 5), (1, 2048, 30000, 64, 8.0, 5), (1, 128, 1500, 20, 25.0, 6)])
@pytest.mark.parametrize("serial", [False, True])
def test_device_dubins_equals_the_oracle(gpu_ctx, star, grid, n, rr, rho, seed, serial):
    og, og8, xs, xg, samples, heads = _dub_query(grid, n, seed, gseed=3 if grid == 2048 else 1)
    gpu_ctx.set_grid(og8)
    res, ro = _device_vs_oracle_dubins(gpu_ctx, og8, star, n, xs, xg, samples, heads, rr, rho, serial=serial)
    _check_dubins_tree(og8, res, rho, 64, xs)
    if not serial:
        _independent_collision_witness(og8, res, samples, heads, rho, 64)
    assert res.n_words > 0 or serial  # (the one-sample-per-iteration kernel does not count its words)


@pytest.mark.gpu
def test_config5_full_size_equals_the_oracle(gpu_ctx):
    """BASELINE configs[4] at its own size -- Dubins-RRT*, 2048 x 2048, n = 100000, r_rewire = 64, rho = 8, 64 headings --
    exactly as `bench.py --config 5` poses query 0: the whole tree (vertices, headings, parents, costs bit for bit, per-iteration
    nearest / accept logs, statistics) equals the oracle's.  The oracle runs in its study mode next to it: no word it evaluates
    is shorter than the slackened chord the kernel prunes with, and a search pruned by that bound ends at the parent the
    index-order walk of rrt.py:515-521 chooses, for every accepted sample."""
    import bench

    cfg = bench.CONFIGS[5]
    n = cfg["n"]
    og = perlin_occupancygrid(cfg["grid"], cfg["grid"], thresh=0.33, seed=cfg["grid_seed"])
    og8 = oracle.og_u8(og)
    free = np.argwhere(og == 0)
    from rrtplanner_amd.oggen import random_connected_pairs

    a, b = random_connected_pairs(og, np.random.default_rng(7), 1)[0]
    rng = np.random.default_rng(0)
    samples = hostprep.draw_free_samples(rng, free, n)
    heads = rng.integers(0, cfg["nh"], size=n)
    xs, xg = (int(a[0]), int(a[1]), 0), (int(b[0]), int(b[1]), 5 % cfg["nh"])
    gpu_ctx.set_grid(og8)
    res, ro = _device_vs_oracle_dubins(gpu_ctx, og8, 1, n, xs, xg, samples, heads, cfg["r_rewire"], cfg["rho"], cfg["nh"], counters=True)
    assert ro.j > 80000 and ro.lb_violations == 0 and ro.lb_mismatch == 0
    assert 0 < res.n_words < ro.n_dubins // 2  # the oracle's walk prices every near-set entry; the kernel one per lane of a pass, up to 64
    _check_dubins_tree(og8, res, cfg["rho"], cfg["nh"], xs)
    assert _independent_collision_witness(og8, res, samples, heads, cfg["rho"], cfg["nh"], count=400) > 50
    # every accept / reject and every chosen parent of the DEVICE's tree recomputed by oracle/dubins_ref.c (libm,
    # no shared header) at the config's full size; the counts go to profiles/ via tools/gpu_round.sh
    a = oracle.dubins_audit(og8, n, 1, samples, heads, res.pts, res.head, res.vcost, res.parent, res.j, r2_rewire=hostprep.radius_threshold(cfg["r_rewire"]),
                            rho=cfg["rho"], nh=cfg["nh"])
    print("config 5 independent audit:", a)
    out = os.environ.get("RRT_AUDIT_OUT")
    if out:
        import json

        with open(out, "w") as f:
            json.dump({"workload": "BASELINE.json configs[4], query 0 (bench.py --config 5), the DEVICE's tree", "audit": a,
                       "by": "oracle/dubins_ref.c (libm, textbook words, own sweep; does not include include/rrt_dubins.h)"}, f, indent=1)
    assert a["n_accepted"] == res.j - 1
    _assert_audit_clean(a, 1)


@pytest.mark.gpu
def test_device_dubins_fuzz_small(gpu_ctx):
    rng = np.random.default_rng(99)
    for case in range(150):
        w, h = int(rng.integers(8, 70)), int(rng.integers(8, 70))
        og8 = (rng.uniform(size=(w, h)) < rng.choice([0.0, 0.1, 0.3])).astype(np.uint8)
        free = np.argwhere(og8 == 0)
        if free.shape[0] < 2:
            continue
        star = int(rng.integers(0, 2))
        n = int(rng.choice([1, 2, 17, 64, 100, 300, 800]))
        nh = int(rng.choice([1, 8, 64, 256]))
        rho = float(rng.choice([0.5, 1.5, 4.0, 12.0]))
        rr = float(rng.choice([2, 8, 20, 500]))
        a, b = free[rng.integers(0, free.shape[0])], free[rng.integers(0, free.shape[0])]
        xs, xg = (int(a[0]), int(a[1]), int(rng.integers(0, nh))), (int(b[0]), int(b[1]), int(rng.integers(0, nh)))
        srng = np.random.default_rng(case)
        samples = hostprep.draw_free_samples(srng, free, n)
        heads = srng.integers(0, nh, size=n)
        gpu_ctx.set_grid(og8)
        try:
            _device_vs_oracle_dubins(gpu_ctx, og8, star, n, xs, xg, samples, heads, rr, rho, nh, serial=bool(case % 3 == 2))
        except AssertionError as e:
            raise AssertionError(f"dubins fuzz case {case}: {w}x{h} star {star} n {n} nh {nh} rho {rho} r {rr} xs {xs} xg {xg}") from e


@pytest.mark.gpu
def test_dubins_planner_class_and_batch_on_device(gpu_ctx):
    og, og8, xs, xg, _, _ = _dub_query(400, 5000, 0)
    p = dub.RRTStarDubins(og, 5000, 48, rho=6.0, pbar=False, seed=2)
    try:
        T, gv = p.plan(np.array(xs), np.array(xg))
    except IndexError:
        pytest.skip("goal pose unreachable in this draw")
    o = orchelp.use_oracle(dub.RRTStarDubins(og, 5000, 48, rho=6.0, pbar=False, seed=2))
    To, go = o.plan(np.array(xs), np.array(xg))
    assert gv == go and list(T.edges) == list(To.edges)
    assert [T.nodes[v]["heading"] for v in T.nodes] == [To.nodes[v]["heading"] for v in To.nodes]
    assert [d["cost"] for _, _, d in T.edges(data=True)] == [d["cost"] for _, _, d in To.edges(data=True)]
    # a batch of Dubins queries (one CU each), and the flag / algorithm mismatch is refused
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    Q, n = 6, 2500
    b = _ffi.Batch(gpu_ctx, Q, n, dubins=True)
    refs, keep = [], []
    sg = np.random.default_rng(5)
    for q in range(Q):
        a, c = random_connected_pair(og, sg)
        qs, qg = (int(a[0]), int(a[1]), q), (int(c[0]), int(c[1]), 3 * q)
        rng = np.random.default_rng(q)
        s = hostprep.draw_free_samples(rng, free, n)
        hd = rng.integers(0, 64, size=n)
        qu, k = _ffi.make_query(_ffi.ALG_DUBINS_STAR if q % 2 else _ffi.ALG_DUBINS, n, qs, qg, s, r2_rewire=hostprep.radius_threshold(30), headings=hd, rho=5.0, nh=64)
        keep.append(k)
        b.set_query(q, qu)
        refs.append(oracle.dubins_plan(og8, n, q % 2, qs, qg, s, hd, r2_rewire=hostprep.radius_threshold(30), rho=5.0, nh=64, logs=False))
    b.launch()
    b.sync()
    for q, (st, ro) in enumerate(refs):
        res = b.get_result(q)
        live = ro.j + (1 if ro.found else 0)
        assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal
        assert np.array_equal(res.parent[:live], ro.parent[:live]) and np.array_equal(res.vcost[:live], ro.vcost[:live])
        assert np.array_equal(res.head[:live], ro.head[:live])
    b.close()
    plain = _ffi.Batch(gpu_ctx, 1, 100)
    with pytest.raises(_ffi.RRTError):
        plain.set_query(0, qu)
    plain.close()


@pytest.mark.gpu
def test_plan_batch_with_dubins_queries(gpu_ctx):
    """rrt_plan_batch picks the Dubins kernel from its first query; a batch that mixes Dubins and straight-line planners is refused."""
    og, og8, xs, xg, samples, heads = _dub_query(200, 1500, 4)
    gpu_ctx.set_grid(og8)
    r2 = hostprep.radius_threshold(24)
    qs, keep, refs = [], [], []
    for k, star in enumerate((1, 0, 1)):
        q, kp = _ffi.make_query(_ffi.ALG_DUBINS_STAR if star else _ffi.ALG_DUBINS, 1500, xs, xg, samples, r2_rewire=r2 if star else 0, headings=heads,
                                rho=4.0 + k, nh=64)
        qs.append(q)
        keep.append(kp)
        refs.append(oracle.dubins_plan(og8, 1500, star, xs, xg, samples, heads, r2_rewire=r2 if star else 0, rho=4.0 + k, nh=64, logs=False))
    rc, res = gpu_ctx.plan_batch(qs, [1500] * 3)
    for r, (st, ro) in zip(res, refs):
        live = ro.j + (1 if ro.found else 0)
        assert r.status == st and r.j == ro.j and r.vgoal == ro.vgoal
        assert np.array_equal(r.parent[:live], ro.parent[:live]) and np.array_equal(r.vcost[:live], ro.vcost[:live])
    plain, kp = _ffi.make_query(1, 1500, xs[:2], xg[:2], samples, r2_rewire=r2)
    with pytest.raises(_ffi.RRTError):
        gpu_ctx.plan_batch([qs[0], plain], [1500, 1500])
    with pytest.raises(_ffi.RRTError):
        _ffi.Batch(gpu_ctx, 1, 100, dubins=True, rewire=True)  # the opt-in rewire is not built for the Dubins planners
