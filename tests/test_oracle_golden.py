"""CPU: the oracle (oracle/rrt_oracle.c) and the host logic of the planner classes against the golden
vectors captured from the real reference (tests/golden/make_golden.py).  Bit-exact for tree
topology, node coordinates, collision decisions and (in fact) edge costs; plot metadata
(`ellipses`) to 1e-12 relative."""
import math

import numpy as np
import pytest

import oracle
import orchelp
from rrtplanner_amd import hostprep
from rrtplanner_amd import rrt as amd

P = orchelp.golden("primitives.npz").z


# ------------------------------------------------------------------------------- primitives
def test_r2norm_matches_reference():
    out = np.array([amd.r2norm(p) for p in P["r2norm_in"]])
    assert np.array_equal(out, P["r2norm_out"])
    # the reference's own assertion (tests/test_rrt.py:68-71)
    assert np.allclose(out, np.linalg.norm(P["r2norm_in"], axis=1))


def test_collisionfree_all_pairs_12x12():
    g = P["cf12_grid"]
    want = P["cf12_free"].astype(bool)
    k = 0
    for a in range(12):
        for b in range(12):
            for c in range(12):
                for d in range(12):
                    got, _ = oracle.collisionfree(g, (a, b), (c, d))
                    assert got == want[k], (a, b, c, d)
                    k += 1


def test_collisionfree_static_helper_all_pairs_12x12():
    g = P["cf12_grid"]
    want = P["cf12_free"].astype(bool)
    pairs = [(a, b, c, d) for a in range(12) for b in range(12) for c in range(12) for d in range(12)]
    for k in range(0, len(pairs), 7):
        a, b, c, d = pairs[k]
        assert amd.RRT.collisionfree(g, (a, b), (c, d)) == want[k]


def test_collisionfree_noise_segments():
    g = orchelp.golden("plans_A.npz").grid("noise200")
    for s, w in zip(P["cf200_seg"], P["cf200_free"]):
        got, _ = oracle.collisionfree(g, s[:2], s[2:])
        assert got == bool(w)


def test_walk_cells_literal_and_cell_count():
    offs = P["walk_offs"]
    free = np.zeros((2048, 2048), dtype=np.uint8)
    for k, s in enumerate(P["walk_seg"]):
        want = P["walk_cells"][offs[k]:offs[k + 1]]
        got = oracle.bresenham_cells(s[:2], s[2:])
        assert np.array_equal(got, want), s
        ok, cells = oracle.collisionfree(free, s[:2], s[2:])
        assert ok and cells == len(want)


def test_within_counts_and_reference_corner_case():
    pts, xq = P["within_pts"], P["within_xq"]
    for k, r in enumerate(P["within_r"]):
        R = hostprep.radius_threshold(int(r) if float(r).is_integer() else float(r))
        for x, c, s in zip(xq, P[f"within_cnt_{k}"], P[f"within_sum_{k}"]):
            w = oracle.within(pts, x, R)
            assert len(w) == c and int(w.sum()) == s
            w2 = amd.RRT.within(pts.astype(np.int64), x.astype(np.int64), r)
            assert len(w2) == c and int(np.sum(w2)) == s
    # tests/test_rrt.py:116-119 of the reference
    assert P["within_corner_count"][0] == 4
    assert amd.RRT.within(np.array([[0, 0], [1, 0], [1, 1], [0, 1]]), np.array([0.5, 0.5]), 1.0).shape[0] == 4


def test_nearest_canonical_policy():
    pts, xq = P["within_pts"], P["within_xq"]
    got = [oracle.nearest(pts, x) for x in xq]
    assert got == P["near_stable"].tolist()
    assert [int(amd.RRT.near(pts.astype(np.int64), x.astype(np.int64))[0]) for x in xq] == P["near_stable"].tolist()
    # raw numpy argsort agrees wherever the minimum is unique
    d2 = ((pts[None, :, :].astype(np.int64) - xq[:, None, :]) ** 2).sum(-1)
    uniq = (d2 == d2.min(axis=1, keepdims=True)).sum(axis=1) == 1
    assert np.array_equal(np.array(got)[uniq], P["near_raw"][uniq])
    assert (~uniq).sum() > 0, "fixture should contain ties"


def test_pcg64_interleave_vector():
    g = np.random.default_rng(0)
    F = 700001
    v = [g.choice(F), g.uniform(0, 1), g.choice(F), g.choice(F), g.uniform(0, 1)]
    assert np.array_equal(np.array(v, dtype=np.float64), P["pcg_interleave"])


def test_rotation_matrix_and_sample_ellipse_table():
    sg, Cs = P["rot_sg"], P["rot_C"]
    W, H = P["ell_WH"]
    for s, Cm in zip(sg, Cs):
        got = hostprep.rotation_to_world_frame(s[:2].astype(np.int64), s[2:].astype(np.int64))
        assert np.array_equal(got, Cm)
    Cmap = {tuple(s.tolist()): Cm for s, Cm in zip(sg, Cs)}
    for row, want in zip(P["ell_in"], P["ell_out"]):
        xs, xg = row[:2].astype(np.int32), row[2:4].astype(np.int32)
        Cm = Cmap[(int(xs[0]), int(xs[1]), int(xg[0]), int(xg[1]))]
        got = oracle.sample_ellipse(Cm, xs, xg, int(W), int(H), row[4], row[5:7])
        assert got.tolist() == want.tolist()


# ------------------------------------------------------------------------------- full plans
GA = orchelp.golden("plans_A.npz")
GS = orchelp.golden("special_A.npz")


def _run_case(G, meta, planner=None):
    og = G.grid(meta["grid"]).astype(np.int64)
    p = planner or orchelp.use_oracle(orchelp.make_planner(amd, meta, og))
    xs, xg = np.array(meta["xstart"]), np.array(meta["xgoal"])
    if meta.get("raises") == "IndexError":
        with pytest.raises(IndexError):
            p.plan(xs, xg)
        assert orchelp.rng_state_tuple(p.rand_gen) == meta["rng_state"]
        return p
    T, gv = p.plan(xs, xg)
    orchelp.check_plan_against_golden(G, meta, p, T, gv)
    return p


@pytest.mark.parametrize("meta", GA.manifest, ids=[m["id"] for m in GA.manifest])
def test_plan_policy_A(meta):
    _run_case(GA, meta)


@pytest.mark.parametrize("meta", [m for m in GS.manifest if "chain" not in m], ids=lambda m: m["id"])
def test_plan_special_cases(meta):
    _run_case(GS, meta)


@pytest.mark.parametrize("tag", ["std", "star", "inf"])
def test_replan_chain_rng_continues_and_set_og(tag):
    chain = [m for m in GS.manifest if m.get("chain") == f"replan__{tag}"]
    assert len(chain) == 3
    p = None
    for m in sorted(chain, key=lambda m: m["step"]):
        og = GS.grid(m["grid"]).astype(np.int64)
        if p is None:
            p = orchelp.use_oracle(orchelp.make_planner(amd, m, og))
        elif m["step"] == 2:
            p.set_og(og)
        _run_case(GS, m, planner=p)


def test_nearest_log_matches_reference_every_iteration():
    """Per-iteration vnearest of the oracle == the reference's near()[0] (policy A)."""
    for meta in GA.manifest:
        if meta["n"] > 400 or meta["alg"] == 2:
            continue
        og8 = GA.grid(meta["grid"])
        rng = np.random.default_rng(meta["seed"])
        free = np.argwhere(og8 == 0)
        samples = hostprep.draw_free_samples(rng, free, meta["n"])
        r2 = hostprep.radius_threshold(meta["r_rewire"]) if meta["r_rewire"] is not None else 0
        st, r = oracle.plan(og8, meta["n"], meta["alg"], meta["xstart"], meta["xgoal"], samples, r2_rewire=r2)
        assert r.nearest_log.tolist() == GA.arr(meta["id"], "nearest_log").tolist(), meta["id"]
        assert r.n_rewired == 0  # rrt.py:536 is never true with the default cost


def test_policy_B_prefix():
    """Raw numpy argsort (implementation-defined ties): our canonical run agrees with it on every
    iteration up to the first one whose minimum distance is tied."""
    GB = orchelp.golden("plans_B.npz")
    checked = 0
    for meta in GB.manifest:
        if meta["alg"] == 2 or meta.get("raises"):
            continue
        og8 = GB.grid(meta["grid"])
        rng = np.random.default_rng(meta["seed"])
        samples = hostprep.draw_free_samples(rng, np.argwhere(og8 == 0), meta["n"])
        r2 = hostprep.radius_threshold(meta["r_rewire"]) if meta["r_rewire"] is not None else 0
        st, r = oracle.plan(og8, meta["n"], meta["alg"], meta["xstart"], meta["xgoal"], samples, r2_rewire=r2)
        raw = GB.arr(meta["id"], "nearest_log")
        diff = np.flatnonzero(raw != r.nearest_log)
        first = int(diff[0]) if diff.size else meta["n"]
        if diff.size:
            # at the first divergence both candidates are equidistant from the sample
            i = first
            jlive = r.jlog[i]
            x = samples[i]
            d2 = ((r.pts[:jlive].astype(np.int64) - x) ** 2).sum(1)
            assert d2[raw[i]] == d2[r.nearest_log[i]] == d2.min(), meta["id"]
        checked += 1
    assert checked >= 6


def test_numpy_reference_like_harness_builds_the_oracles_tree():
    """oracle/numpy_like.py (the "reference-like" CPU timing harness of bench.py) follows the same rules as the oracle."""
    from oracle import numpy_like
    from rrtplanner_amd import hostprep
    from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

    og = perlin_occupancygrid(160, 120, seed=3)
    og8 = oracle.og_u8(og)
    xs, xg = random_connected_pair(og, np.random.default_rng(5))
    free = np.argwhere(og8 == 0)
    n = 900
    samples = hostprep.draw_free_samples(np.random.default_rng(1), free, n)
    pts, par, vc, j, it, _ = numpy_like.rrtstar_like(og8, n, xs, xg, samples, 18)
    st, ro = oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=hostprep.radius_threshold(18))
    assert it == n and j == ro.j
    assert np.array_equal(pts[:j], ro.pts[:j]) and np.array_equal(vc[:j], ro.vcost[:j])
    assert all(par[c] == ro.parent[c] for c in range(1, j))


# ------------------------------------------------------------------------------- bench-scale goldens from the real reference
GBIG = orchelp.golden("plans_big_A.npz")


@pytest.mark.parametrize("meta", GBIG.manifest, ids=[m["id"] for m in GBIG.manifest])
def test_plan_bench_scale_and_grid_dtypes(meta):
    """The oracle (and the host logic around it) against the reference at bench scale: query 0 of BASELINE config 4 on the
    bench's own 1024x1024 grid (n = 20000), Informed RRT* on 400x400 (n = 6000), and the reference's 8 grid dtypes
    (tests/test_rrt.py:8-17; a fractional value is an obstacle like any non-zero)."""
    og = GBIG.grid(meta["grid"]).astype(np.int64)
    if "og_dtype" in meta:
        dt = {"int": int, "float": float}.get(meta["og_dtype"]) or getattr(np, meta["og_dtype"])
        og = og.astype(dt)
        if meta["fractional"]:
            og[30, 30] = 0.25
    p = orchelp.use_oracle(orchelp.make_planner(amd, meta, og))
    T, gv = p.plan(np.array(meta["xstart"]), np.array(meta["xgoal"]))
    orchelp.check_plan_against_golden(GBIG, meta, p, T, gv)


def test_bench_grid_generator_has_not_drifted():
    """bench.py regenerates its grid from the seed; the golden holds the grid the reference ran on."""
    from rrtplanner_amd.oggen import perlin_occupancygrid

    assert np.array_equal((perlin_occupancygrid(1024, 1024, thresh=0.33, seed=1) != 0).astype(np.uint8), GBIG.grid("bench1024"))
