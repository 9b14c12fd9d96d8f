"""GPU: the HIP path through the C ABI against the CPU oracle and the reference's golden vectors.

Bit-exact for tree topology, node coordinates, collision decisions and edge costs (f64).  Everything
here calls librrt_hip.so; the oracle is only the checker."""
import math
import os

import numpy as np
import pytest

import oracle
import orchelp
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd import rrt as amd
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

pytestmark = pytest.mark.gpu

P = orchelp.golden("primitives.npz").z
GA = orchelp.golden("plans_A.npz")
GS = orchelp.golden("special_A.npz")
GBIG = orchelp.golden("plans_big_A.npz")


# ------------------------------------------------------------------------------- primitives
def test_sqrt_of_every_squared_distance_is_correctly_rounded(gpu_ctx):
    """r2norm radicands on a 2048x2048 grid are the integers below 2^23: check them ALL against the
    host's correctly rounded sqrt (the reference's math.sqrt)."""
    step = 1 << 22
    for lo in range(0, 1 << 23, step):
        got = gpu_ctx.prim_sqrt_u32(lo, step)
        want = np.sqrt(np.arange(lo, lo + step, dtype=np.float64))
        assert np.array_equal(got, want)
    got = gpu_ctx.prim_sqrt_u32((1 << 24) - 4096, 4096)
    assert np.array_equal(got, np.sqrt(np.arange((1 << 24) - 4096, 1 << 24, dtype=np.float64)))


def test_block_kernel_short_sqrt_is_exact_for_every_radicand(gpu_ctx):
    step = 1 << 22
    for lo in range(0, 1 << 24, step):
        assert np.array_equal(gpu_ctx.prim_sqrt_u24(lo, step), np.sqrt(np.arange(lo, lo + step, dtype=np.float64)))


def test_sqrt_f64_random(gpu_ctx):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 1e7, 2_000_000), rng.uniform(0, 1, 500_000) ** 8 * 1e12,
                        np.array([0.0, 1.0, 2.0, 0.25, 1e-300, 1e300])])
    assert np.array_equal(gpu_ctx.prim_sqrt_f64(x), np.sqrt(x))


def test_collisionfree_all_pairs_12x12(gpu_ctx):
    g = P["cf12_grid"]
    gpu_ctx.set_grid(g)
    ab = np.array([(a, b, c, d) for a in range(12) for b in range(12) for c in range(12) for d in range(12)], dtype=np.int32)
    free, cells = gpu_ctx.prim_collisionfree(ab)
    assert np.array_equal(free, P["cf12_free"].astype(bool))
    for k in range(0, len(ab), 97):
        ok, c = oracle.collisionfree(g, ab[k, :2], ab[k, 2:])
        assert ok == free[k] and c == cells[k]


def test_collisionfree_every_short_segment(gpu_ctx):
    """Every segment of fewer than 64 steps (the device evaluates their cells without the integer fix-up of the closed form:
    rrt_device.h short_line_cell), from two start cells to every cell within 63 of them, on a sparse random grid: decision and cell
    count against the oracle's serial walk (rrt.py:202-229)."""
    rng = np.random.default_rng(11)
    g = (rng.uniform(size=(160, 150)) < 0.01).astype(np.uint8)
    gpu_ctx.set_grid(g)
    for (cx, cy) in ((80, 75), (66, 70)):
        g[cx, cy] = 0
    gpu_ctx.set_grid(g)
    segs = []
    for (cx, cy) in ((80, 75), (66, 70)):
        for dx in range(-63, 64):
            for dy in range(-63, 64):
                segs.append((cx, cy, cx + dx, cy + dy))
                if (dx + dy) % 7 == 0:
                    segs.append((cx + dx, cy + dy, cx, cy))
    seg = np.array(segs, dtype=np.int32)
    free, cells = gpu_ctx.prim_collisionfree(seg)
    for k in range(len(seg)):
        ok, c = oracle.collisionfree(g, seg[k, :2], seg[k, 2:])
        assert ok == free[k] and c == cells[k], seg[k]


def test_collisionfree_noise_and_long_segments(gpu_ctx):
    g = GA.grid("noise200")
    gpu_ctx.set_grid(g)
    free, cells = gpu_ctx.prim_collisionfree(P["cf200_seg"])
    assert np.array_equal(free, P["cf200_free"].astype(bool))
    # 2048 x 2048 noise grid, long random segments, against the oracle (decision and cell count)
    big = (perlin_occupancygrid(2048, 2048, seed=3) != 0).astype(np.uint8)
    gpu_ctx.set_grid(big)
    rng = np.random.default_rng(5)
    seg = rng.integers(0, 2048, size=(20000, 4)).astype(np.int32)
    seg[:2000, 2:] = np.clip(seg[:2000, :2] + rng.integers(-70, 70, size=(2000, 2)), 0, 2047)
    free, cells = gpu_ctx.prim_collisionfree(seg)
    for k in range(len(seg)):
        ok, c = oracle.collisionfree(big, seg[k, :2], seg[k, 2:])
        assert ok == free[k] and c == cells[k], seg[k]
    # empty grid: cells == walk length (closed form covers the whole walk)
    gpu_ctx.set_grid(np.zeros((2048, 2048), dtype=np.uint8))
    free, cells = gpu_ctx.prim_collisionfree(P["walk_seg"])
    assert free.all()
    assert np.array_equal(cells, np.diff(P["walk_offs"]).astype(np.int32))


def test_nearest_and_within_fixture(gpu_ctx):
    pts, xq = P["within_pts"], P["within_xq"]
    for k, r in enumerate(P["within_r"]):
        R = hostprep.radius_threshold(int(r) if float(r).is_integer() else float(r))
        nn, cnt, isum = gpu_ctx.prim_nearest_within(pts, xq, R)
        assert nn.tolist() == P["near_stable"].tolist()
        assert cnt.tolist() == P[f"within_cnt_{k}"].tolist()
        assert isum.tolist() == P[f"within_sum_{k}"].tolist()


@pytest.mark.parametrize("j", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8192, 50000, 100000])
def test_nearest_and_within_sizes_with_ties(gpu_ctx, j):
    rng = np.random.default_rng(j)
    side = 64 if j < 5000 else 2048  # small side => many equidistant nodes
    pts = rng.integers(0, side, size=(j, 2)).astype(np.int32)
    xq = rng.integers(0, side, size=(24, 2)).astype(np.int32)
    xq[:4] = pts[rng.integers(0, j, size=4)]  # distance 0
    for R in (0, 1, 64 * 64, 1 << 24):
        nn, cnt, isum = gpu_ctx.prim_nearest_within(pts, xq, R)
        for k, x in enumerate(xq):
            assert nn[k] == oracle.nearest(pts, x)
            w = oracle.within(pts, x, R)
            assert cnt[k] == len(w) and isum[k] == int(w.astype(np.int64).sum())


# ------------------------------------------------------------------------------- full plans vs golden
def _run_case(G, meta, planner=None):
    og = G.grid(meta["grid"]).astype(np.int64)
    p = planner or orchelp.make_planner(amd, meta, og)
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


@pytest.mark.parametrize("meta", GBIG.manifest, ids=[m["id"] for m in GBIG.manifest])
def test_plan_bench_scale_and_grid_dtypes(meta):
    """The device against the REAL reference at bench scale: query 0 of BASELINE config 4 on the bench's own 1024x1024 grid
    (n = 20000, r_rewire = 64), Informed RRT* on 400x400 (n = 6000), and the reference's 8 occupancy-grid dtypes
    (tests/test_rrt.py:8-17; a fractional cell value is an obstacle like any non-zero)."""
    og = GBIG.grid(meta["grid"]).astype(np.int64)
    if "og_dtype" in meta:
        dt = {"int": int, "float": float}.get(meta["og_dtype"]) or getattr(np, meta["og_dtype"])
        og = og.astype(dt)
        if meta["fractional"]:
            og[30, 30] = 0.25
    p = orchelp.make_planner(amd, meta, og)
    T, gv = p.plan(np.array(meta["xstart"]), np.array(meta["xgoal"]))
    orchelp.check_plan_against_golden(GBIG, meta, p, T, gv)


@pytest.mark.parametrize("tag", ["std", "star", "inf"])
def test_replan_chain_rng_continues_and_set_og(tag):
    chain = sorted([m for m in GS.manifest if m.get("chain") == f"replan__{tag}"], key=lambda m: m["step"])
    p = None
    for m in chain:
        og = GS.grid(m["grid"]).astype(np.int64)
        if p is None:
            p = orchelp.make_planner(amd, m, og)
        elif m["step"] == 2:
            p.set_og(og)
        _run_case(GS, m, planner=p)


# ------------------------------------------------------------------------------- device vs oracle, larger
# teams of up to 64 workers per query (default: pipelined, one more CU that only commits), the same without the pipeline, capped
# teams (2, 3 and 4 workers: 16 samples per member, pipelined and -- 4 -- not; 16: 4 samples per member, 4 waves per sample), one CU,
# one sample per iteration, and a team that loses a member (must finish on one CU per query)
KERNELS = ["team", "teamnp", "team2", "team3", "team4", "team4np", "team16", "block", "block16", "serial", "teamfault"]
_KERNEL_ARGS = {"team": {}, "teamnp": {"pipe": False}, "team2": {"team": 2}, "team3": {"team": 3}, "team4": {"team": 4},
                "team4np": {"team": 4, "pipe": False}, "team16": {"team": 16},
                # one CU per query: the barrier-free pipeline (rrt_pipe.h; Informed queries run the block kernel), and the block kernel
                "block": {"team": 1}, "block16": {"team": 1, "pipe1": False},
                "serial": {"serial": True}, "teamfault": {"team": 8, "team_fault": True}}
KERNELS_NOFAULT = [k for k in KERNELS if k != "teamfault"]


def _oracle_vs_device(ctx, og8, alg, n, seed, xs, xg, r_rewire=None, r_goal=None, kernel="team"):
    rng = np.random.default_rng(seed)
    free = np.argwhere(og8 == 0)
    samples = hostprep.draw_free_samples(rng, free, n)
    r2 = hostprep.radius_threshold(r_rewire) if r_rewire is not None else 0
    gd2 = hostprep.goal_threshold(r_goal) if r_goal is not None else 0
    Cm = hostprep.rotation_to_world_frame(np.asarray(xs, dtype=np.int64), np.asarray(xg, dtype=np.int64)) if alg == 2 else None
    q, keep = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2, goal_d2=gd2, Cmat=Cm)
    rc, res = ctx.plan(q, n, logs=True, **_KERNEL_ARGS[kernel])
    st, ro = oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, r_goal=r_goal or 0.0, Cmat=Cm)
    ub = None
    if rc == _ffi.RRT_NEED_UNITBALL:
        assert st == oracle.ORC_NEED_UNITBALL and res.i_switch == ro.i_switch
        ub = hostprep.draw_unitball(rng, n - res.i_switch)
        rc = ctx.plan_resume(ub, res)
        st, ro = oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, r_goal=r_goal, unitball=ub, ub_offset=res.i_switch, Cmat=Cm)
    assert rc == st
    assert res.j == ro.j and res.found == ro.found and res.vgoal == ro.vgoal and res.i_switch == ro.i_switch
    live = ro.j + (1 if ro.found else 0)
    assert np.array_equal(res.nearest_log, ro.nearest_log)
    assert np.array_equal(res.accept_log, ro.accept_log)
    assert np.array_equal(res.j_log, ro.jlog)
    assert np.array_equal(res.pts[:live], ro.pts[:live])
    assert np.array_equal(res.parent[:live], ro.parent[:live])
    assert np.array_equal(res.vcost[:live], ro.vcost[:live])  # bit-exact f64 (tolerance stated by the north star: 1e-6)
    assert np.array_equal(res.cbest_log, ro.cbest_log, equal_nan=True)
    assert res.sum_j == ro.sum_j and res.sum_cells_nn == ro.sum_cells_nn and res.sum_near == ro.sum_near
    return res, ro


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("alg,rr,rg", [(0, None, None), (1, 64, None), (2, 64, 12)])
def test_device_vs_oracle_1024_n6000(gpu_ctx, alg, rr, rg, kernel):
    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(7))
    _oracle_vs_device(gpu_ctx, og8, alg, 6000, 0, xs, xg, rr, rg, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS_NOFAULT)
def test_device_vs_oracle_beyond_lds_capacity(gpu_ctx, kernel):
    """n = 40000 exceeds the LDS-resident node chunks: the scan crosses from LDS chunks into HBM chunks."""
    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(7))
    _oracle_vs_device(gpu_ctx, og8, 1, 40000, 0, xs, xg, 64, None, kernel=kernel)


@pytest.mark.parametrize("kernel", ["team", "block"])
def test_device_vs_oracle_informed_1024_n25000(gpu_ctx, kernel):
    """Informed RRT* well into the ellipse phase (dense near sets, blocks cut where the ellipse shrinks), BASELINE config 3's grid."""
    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(7))
    res, ro = _oracle_vs_device(gpu_ctx, og8, 2, 25000, 0, xs, xg, 64, 12, kernel=kernel)
    assert res.i_switch < 25000  # the ellipse phase was reached


@pytest.mark.parametrize("kernel", ["team", "block"])
def test_device_vs_oracle_informed_config3_full_size(gpu_ctx, kernel):
    """BASELINE config 3 at its full size: Informed RRT*, 1024x1024, n = 50000, r_rewire = 64, r_goal = 12 (the pipelined team's
    RESTART / void blocks and every ellipse move of the bench run), every array and per-iteration log against the oracle."""
    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(7))
    res, ro = _oracle_vs_device(gpu_ctx, og8, 2, 50000, 0, xs, xg, 64, 12, kernel=kernel)
    assert res.i_switch < 50000


@pytest.mark.parametrize("kernel", KERNELS_NOFAULT)
def test_device_vs_oracle_near_set_spills(gpu_ctx, kernel):
    """r_rewire far beyond the grid: the near set is the whole tree and overflows the LDS lists."""
    og = perlin_occupancygrid(256, 256, seed=4)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(1))
    _oracle_vs_device(gpu_ctx, og8, 1, 5000, 3, xs, xg, 1e6, None, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("alg,rr,rg,grid,n,seed", [
    (1, 9, None, 64, 3000, 1),      # tiny grid: most samples are duplicates or interact inside a block
    (2, 30, 8, 96, 2500, 2),        # informed on a small grid: many ellipse changes cut blocks
    (1, 200, None, 300, 4000, 3),   # radius comparable to the grid
    (0, None, None, 40, 1500, 4),   # n close to the number of free cells
])
def test_device_vs_oracle_dense_interactions(gpu_ctx, alg, rr, rg, grid, n, seed, kernel):
    og = perlin_occupancygrid(grid, grid, seed=seed)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(seed))
    _oracle_vs_device(gpu_ctx, og8, alg, n, seed, xs, xg, rr, rg, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS_NOFAULT)
def test_device_vs_oracle_2048_grid(gpu_ctx, kernel):
    """The largest supported grid (BASELINE config 5's size): squared distances use all 23 key bits."""
    og = perlin_occupancygrid(2048, 2048, seed=3)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(11))
    _oracle_vs_device(gpu_ctx, og8, 1, 30000, 5, xs, xg, 64, None, kernel=kernel)
    _oracle_vs_device(gpu_ctx, og8, 2, 12000, 6, (3, 2044), (2040, 5), 200.5, 40, kernel=kernel) if og8[3, 2044] == 0 and og8[2040, 5] == 0 else None


@pytest.mark.parametrize("kernel", KERNELS_NOFAULT)
def test_fuzz_small_queries_vs_oracle(gpu_ctx, kernel):
    """600 random small queries (grids 8..70 cells wide, n up to 700, radii from 0 to beyond the grid, all three planners, starts
    on obstacles, n larger than the free space) -- every block-cut / capacity / duplicate / interaction path of the kernels."""
    rng = np.random.default_rng(20260101)
    for case in range(600):
        w, h = int(rng.integers(8, 70)), int(rng.integers(8, 70))
        dens = rng.choice([0.0, 0.1, 0.3, 0.5])
        og8 = (rng.uniform(size=(w, h)) < dens).astype(np.uint8)
        if case % 3 == 0:
            og8 = oracle.og_u8(perlin_occupancygrid(w, h, seed=case))
        free = np.argwhere(og8 == 0)
        if free.shape[0] < 2:
            continue
        alg = int(rng.integers(0, 3))
        n = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 192, 193, 257, 700, 1500]))  # (64-sample blocks, up to two in flight)
        rr = float(rng.choice([0, 1, 1.5, 3, 7.9, 12, 25, 64, 500]))
        rg = float(rng.choice([0, 1, 2.5, 6, 15, 100]))
        xs = free[rng.integers(0, free.shape[0])] if case % 11 else np.array([int(rng.integers(0, w)), int(rng.integers(0, h))])
        xg = free[rng.integers(0, free.shape[0])]
        if alg == 2 and (xs == xg).all():
            continue
        gpu_ctx.set_grid(og8)
        try:
            _oracle_vs_device(gpu_ctx, og8, alg, n, case, xs, xg, rr if alg else None, rg if alg == 2 else None, kernel=kernel)
        except AssertionError as e:
            raise AssertionError(f"fuzz case {case}: grid {w}x{h} dens {dens} alg {alg} n {n} r {rr} rg {rg} xs {xs} xg {xg}") from e


def test_batch_of_queries_matches_single_queries(gpu_ctx):
    og = perlin_occupancygrid(512, 512, seed=2)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    sg = np.random.default_rng(7)
    Q, n = 12, 3000
    b = _ffi.Batch(gpu_ctx, Q, n)
    keep, refs = [], []
    for q in range(Q):
        xs, xg = random_connected_pair(og, sg)
        samples = hostprep.draw_free_samples(np.random.default_rng(q), free, n)
        alg = q % 2
        qu, k = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=hostprep.radius_threshold(48))
        keep.append(k)
        b.set_query(q, qu)
        refs.append(oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=hostprep.radius_threshold(48)))
    for rep in range(2):  # second pass: rearm keeps the inputs resident
        b.launch()
        b.sync()
        assert b.elapsed_ms() > 0
        for q in range(Q):
            res = b.get_result(q)
            st, ro = refs[q]
            live = ro.j + (1 if ro.found else 0)
            assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal
            assert np.array_equal(res.pts[:live], ro.pts[:live])
            assert np.array_equal(res.parent[:live], ro.parent[:live])
            assert np.array_equal(res.vcost[:live], ro.vcost[:live])
        b.rearm()
    b.close()


@pytest.mark.parametrize("with_informed", [False, True])
def test_two_cus_per_query_without_a_committer_runs_the_pipeline(gpu_ctx, with_informed):
    """100 queries: two CUs per query fit, a third (a pipelined team's committer) does not.  Without Informed queries the batch runs the
    one-CU pipeline (faster than the unpipelined team of two); with one, the team of two; a caller's cap of 2 keeps the team.  Results
    equal the oracle's either way."""
    og = perlin_occupancygrid(300, 260, seed=6)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    sg = np.random.default_rng(12)
    Q, n = 100, 500
    r2 = hostprep.radius_threshold(24)
    for cap in (None, 2):
        b = _ffi.Batch(gpu_ctx, Q, n, team=cap)
        keep, refs, algs = [], [], []
        for q in range(Q):
            xs, xg = random_connected_pair(og, sg)
            samples = hostprep.draw_free_samples(np.random.default_rng(2000 + q), free, n)
            alg = 2 if (with_informed and q == 7) else q % 2
            kw = dict(goal_d2=hostprep.goal_threshold(5), Cmat=hostprep.rotation_to_world_frame(xs, xg)) if alg == 2 else {}
            qu, k = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2, **kw)
            keep.append(k)
            b.set_query(q, qu)
            algs.append(alg)
            refs.append(None if alg == 2 else oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2))
        b.launch()
        b.sync()
        if cap is None and not with_informed:
            assert b.kernel_name() == "rrt_pipe_kernel"
        else:
            assert b.kernel_name().startswith("rrt_expand_block_kernel<2, 16, false")
        for q in range(Q):
            if algs[q] == 2:
                continue  # (needs the host's unit-ball stream to go on: not this test's subject)
            res = b.get_result(q)
            st, ro = refs[q]
            live = ro.j + (1 if ro.found else 0)
            assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal, q
            assert np.array_equal(res.pts[:live], ro.pts[:live]), q
            assert np.array_equal(res.parent[:live], ro.parent[:live]), q
            assert np.array_equal(res.vcost[:live], ro.vcost[:live]), q
        b.close()


@pytest.mark.parametrize("pipe1", [True, False])
def test_more_queries_than_compute_units_one_cu_each(gpu_ctx, pipe1):
    """300 queries on one CU each (more workgroups than the device has CUs: the last ones start when the first have finished), the
    barrier-free pipeline and the block kernel; RRTStandard and RRT* mixed, two radii."""
    og = perlin_occupancygrid(300, 260, seed=5)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    sg = np.random.default_rng(11)
    Q, n = 300, 400
    b = _ffi.Batch(gpu_ctx, Q, n, pipe1=pipe1)
    keep, refs = [], []
    for q in range(Q):
        xs, xg = random_connected_pair(og, sg)
        samples = hostprep.draw_free_samples(np.random.default_rng(1000 + q), free, n)
        alg, r2 = q % 2, hostprep.radius_threshold(20 if q % 3 else 70)
        qu, k = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2)
        keep.append(k)
        b.set_query(q, qu)
        refs.append(oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2))
    for rep in range(2):
        b.launch()
        b.sync()
        assert b.kernel_name() == ("rrt_pipe_kernel" if pipe1 else "rrt_expand_block_kernel<1, 16, false, false>")
        for q in range(Q):
            res = b.get_result(q)
            st, ro = refs[q]
            live = ro.j + (1 if ro.found else 0)
            assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal, q
            assert np.array_equal(res.pts[:live], ro.pts[:live]), q
            assert np.array_equal(res.parent[:live], ro.parent[:live]), q
            assert np.array_equal(res.vcost[:live], ro.vcost[:live]), q
            assert res.sum_j == ro.sum_j and res.sum_cells_nn == ro.sum_cells_nn and res.sum_near == ro.sum_near, q
        b.rearm()
    b.close()


def _bench_config4_queries(og, free, Q, n, first=0, stride=1):
    """bench.py's config-4 queries first, first + stride, ...: start/goal from default_rng(7), planner seed = query index."""
    from rrtplanner_amd.oggen import random_connected_pairs

    pairs = random_connected_pairs(og, np.random.default_rng(7), first + stride * Q)
    out = []
    for k in range(Q):
        g = first + stride * k
        xs, xg = pairs[g]
        samples = hostprep.draw_free_samples(np.random.default_rng(g), free, n)
        out.append((xs, xg, samples))
    return out


def test_config4_share_of_one_gpu_equals_the_oracle(gpu_ctx):
    """BASELINE configs[3]'s per-GPU share exactly as bench.py's `batched` leg runs it: 64 independent RRT* queries, n = 20000,
    r_rewire = 64, the 1024x1024 bench grid, default teams (3 workers + 1 committer per query, all 256 CUs busy, 64 teams
    contending for L2).
    Every query's nodes, parents and costs must equal the oracle's, no hand-off may time out, and a second launch after
    rearm must reproduce them."""
    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    Q, n = 64, 20000
    r2 = hostprep.radius_threshold(64)
    qs = _bench_config4_queries(og, free, Q, n)
    b = _ffi.Batch(gpu_ctx, Q, n)
    keep = []
    for q, (xs, xg, samples) in enumerate(qs):
        qu, k = _ffi.make_query(1, n, xs, xg, samples, r2_rewire=r2)
        keep.append(k)
        b.set_query(q, qu)
    refs = [oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=r2, logs=False) for xs, xg, samples in qs]
    for rep in range(2):
        b.launch()
        b.sync()
        assert b.team() == (3, 0) and b.pipelined()  # 64 x (3 workers + 1 committer) = all 256 CUs
        for q in range(Q):
            res = b.get_result(q)
            st, ro = refs[q]
            live = ro.j + (1 if ro.found else 0)
            assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal and res.found == ro.found, q
            assert np.array_equal(res.pts[:live], ro.pts[:live]), q
            assert np.array_equal(res.parent[:live], ro.parent[:live]), q
            assert np.array_equal(res.vcost[:live], ro.vcost[:live]), q
            assert res.sum_j == ro.sum_j and res.sum_cells_nn == ro.sum_cells_nn and res.sum_near == ro.sum_near, q
        b.rearm()
    b.close()
    # query 0 of this batch is also pinned to the real reference: tests/golden/plans_big_A.npz (bench1024__star_r64__s0__n20000)
    m = GBIG.by_id["bench1024__star_r64__s0__n20000"]
    assert m["xstart"] == [int(v) for v in qs[0][0]] and m["xgoal"] == [int(v) for v in qs[0][1]]
    live = refs[0][1].j + 1
    assert np.array_equal(GBIG.arr(m["id"], "parent")[:live], refs[0][1].parent[:live])


def test_config4_shard_of_rank_3_of_8_equals_the_oracle(gpu_ctx):
    """What rank 3 of an 8-GPU run of BASELINE configs[3] computes (bench.py: query g = rank + world * slot, i.e. queries 3, 11,
    ... 507 of the 512): its 64 queries as one batch on this GPU, every tree equal to the oracle's.  One GPU cannot run the
    other seven ranks, but it can run any rank's share."""
    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    Q, n, rank, world = 64, 20000, 3, 8
    r2 = hostprep.radius_threshold(64)
    from rrtplanner_amd import multi

    assert multi.shard_queries(Q * world, world, rank) == [rank + world * k for k in range(Q)]
    qs = _bench_config4_queries(og, free, Q, n, first=rank, stride=world)
    b = _ffi.Batch(gpu_ctx, Q, n)
    keep = []
    for q, (xs, xg, samples) in enumerate(qs):
        qu, k = _ffi.make_query(1, n, xs, xg, samples, r2_rewire=r2)
        keep.append(k)
        b.set_query(q, qu)
    b.launch()
    b.sync()
    assert b.team() == (3, 0) and b.pipelined()
    for q, (xs, xg, samples) in enumerate(qs):
        st, ro = oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=r2, logs=False)
        res = b.get_result(q)
        live = ro.j + (1 if ro.found else 0)
        assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal and res.found == ro.found, q
        assert np.array_equal(res.pts[:live], ro.pts[:live]) and np.array_equal(res.parent[:live], ro.parent[:live]), q
        assert np.array_equal(res.vcost[:live], ro.vcost[:live]), q
    b.close()


def _run_bench(extra_env, *args):
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_as_a_rank_and_as_its_own_launcher():
    """bench.py in the three ways it can be started on a one-GPU box: plainly; as rank 0 of 1 the way a launcher starts it
    (communicator, gather and max-over-ranks timing in the loop); and the JSON contract of both.  The rate with the collective
    in the loop stays within a few per cent of the plain run, and no roofline fraction above 1 is printed anywhere."""
    args = ("--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-batched")
    plain = _run_bench({}, *args)
    ranked = _run_bench({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29731"}, *args)
    for d in (plain, ranked):
        assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["unit"] == "nodes/s" and d["scaling"] == "weak"
        r = d["roofline"]
        assert r["frac"] is None or 0 < r["frac"] <= 1
        assert r["kernel"].startswith("rrt_expand_block_kernel<64, 1, true")
        assert d["host_gap_ms"] == pytest.approx(d["ms_per_step"] - r["kernel_ms"]) and d["config"]["team_fallbacks"] == 0
    assert plain["config"]["collective"] is None and "ncclAllGather" in ranked["config"]["collective"]
    assert ranked["value"] == pytest.approx(plain["value"], rel=0.06)
    assert plain["host_gap_ms"] < 0.25, plain  # the step is the kernel: what the host adds stays below a quarter millisecond


def test_two_contexts_share_the_device_without_timeouts():
    """two planners in one process (two contexts, two host threads) used to size their teams as if each had the whole
    device, and each then waited 0.5 s for members that could not be resident.  Now a launch claims its compute units in the
    library's per-device registry and takes the largest team that fits next to the launches in flight: both batches finish with
    the oracle's trees, no hand-off times out, the later launch of a pair runs a smaller team, and alone again a batch gets its
    full team back."""
    import threading
    import time

    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    free = np.argwhere(og8 == 0)
    Q, n = 8, 12000
    r2 = hostprep.radius_threshold(64)
    qs = _bench_config4_queries(og, free, 2 * Q, n)
    refs = [oracle.plan(og8, n, 1, xs, xg, s, r2_rewire=r2, logs=False) for xs, xg, s in qs]
    ctxs = [_ffi.Context(0), _ffi.Context(0)]
    batches, keep = [], []
    for c, ctx in enumerate(ctxs):
        ctx.set_grid(og8)
        b = _ffi.Batch(ctx, Q, n)
        for q in range(Q):
            xs, xg, s = qs[c * Q + q]
            qu, k = _ffi.make_query(1, n, xs, xg, s, r2_rewire=r2)
            keep.append(k)
            b.set_query(q, qu)
        batches.append(b)
    assert batches[0].team_info()["created"] == 16 and batches[1].team_info()["created"] == 16  # 8 x (16 + 1) = 136 of 256 CUs each
    gate = threading.Barrier(2)
    errors, wall = [], [0.0, 0.0]

    def run(c):
        try:
            b = batches[c]
            for rep in range(4):
                b.rearm()
                gate.wait()
                t0 = time.perf_counter()
                b.launch()
                b.sync()
                wall[c] = max(wall[c], time.perf_counter() - t0)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            gate.abort()

    th = [threading.Thread(target=run, args=(c,)) for c in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    info = [b.team_info() for b in batches]
    assert info[0]["timeouts"] == 0 and info[1]["timeouts"] == 0, info
    assert info[0]["shrunk"] + info[1]["shrunk"] >= 1, info  # the two 136-CU shapes do not fit 256 CUs together
    assert max(wall) < 0.3, wall  # nowhere near a 0.5 s hand-off time-out
    for c, b in enumerate(batches):
        for q in range(Q):
            st, ro = refs[c * Q + q]
            res = b.get_result(q)
            live = ro.j + (1 if ro.found else 0)
            assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal, (c, q)
            assert np.array_equal(res.parent[:live], ro.parent[:live]) and np.array_equal(res.vcost[:live], ro.vcost[:live]), (c, q)
    before = batches[0].team_info()["shrunk"]
    batches[0].rearm()
    batches[0].launch()
    batches[0].sync()  # alone on the device: the full team again
    assert batches[0].team_info() == dict(created=16, last=16, timeouts=0, shrunk=before)
    for b in batches:
        b.close()
    for ctx in ctxs:
        ctx.close()


def test_no_size_at_which_the_class_refuses():
    """VERDICT r2, missing 2: the reference's arrays are unbounded, the expansion kernels take grids up to 2048 x 2048 and n up to
    262143.  Beyond that the planner classes run the host-driven loop over the device primitives (rrt_tree_query with the wide
    line walk) -- slower, same results: a 2600 x 2200 grid and n = 262200 against the oracle; the batch API still says
    RRT_E_UNSUPPORTED for such a grid, and a grid beyond 32767 cells per axis is refused by rrt_set_grid."""
    og = np.zeros((2600, 2200), dtype=np.int64)
    og[1000:1010, :1800] = 1
    og[1700:1705, 400:] = 1
    og8 = oracle.og_u8(og)
    xs, xg = np.array((5, 5)), np.array((2500, 2100))
    for cls, alg, kw in ((amd.RRTStar, 1, dict(r_rewire=300)), (amd.RRTStandard, 0, {})):
        p = cls(og, 3000, pbar=False, seed=0, **kw)
        assert p._beyond_the_kernels()
        T, gv = p.plan(xs, xg)
        samples = hostprep.draw_free_samples(np.random.default_rng(0), np.argwhere(og == 0), 3000)
        st, ro = oracle.plan(og8, 3000, alg, xs, xg, samples, r2_rewire=hostprep.radius_threshold(kw.get("r_rewire", 0)), logs=False)
        live = ro.j + 1
        assert st == 0 and ro.found and gv == ro.vgoal
        par = np.full(live, -1, dtype=np.int64)
        cost = np.zeros(live)
        for u, v, d in T.edges(data=True):
            par[v], cost[v] = u, d["cost"]
        assert np.array_equal(par, ro.parent[:live]) and np.array_equal(cost[1:], ro.vcost[1:live])
        assert np.array_equal(np.array([T.nodes[v]["pt"] for v in range(live)]), ro.pts[:live])
    ctx = _ffi.Context(0)
    ctx.set_grid(og8)
    with pytest.raises(_ffi.RRTError) as e:
        _ffi.Batch(ctx, 1, 100)
    assert e.value.code == _ffi.RRT_E_UNSUPPORTED
    seg = np.array([[5, 5, 2599, 2199], [2599, 0, 0, 2199], [0, 1500, 2599, 1501], [1200, 100, 1201, 2100]], dtype=np.int32)
    free, cells = ctx.prim_collisionfree(seg)  # the wide line walk (64-bit quotient) against the oracle's literal walk
    for k in range(len(seg)):
        ok, c = oracle.collisionfree(og8, seg[k, :2], seg[k, 2:])
        assert ok == free[k] and c == cells[k]
    with pytest.raises(_ffi.RRTError):
        _ffi.lib()  # (keep the library loaded)
        ctx.set_grid(np.zeros((32768, 1), dtype=np.uint8))
    ctx.close()
    # n beyond 262143 (on a small grid, small radius): 262200 iterations of one device round trip each
    og2 = perlin_occupancygrid(700, 700, seed=2)
    xs2, xg2 = random_connected_pair(og2, np.random.default_rng(3))
    n = 262200
    p = amd.RRTStar(og2, n, 8, pbar=False, seed=0)
    assert p._beyond_the_kernels()
    T, gv = p.plan(xs2, xg2)
    samples = hostprep.draw_free_samples(np.random.default_rng(0), np.argwhere(og2 == 0), n)
    st, ro = oracle.plan(oracle.og_u8(og2), n, 1, xs2, xg2, samples, r2_rewire=hostprep.radius_threshold(8), logs=False)
    assert st == 0 and gv == ro.vgoal and T.number_of_nodes() == (n + 1 if ro.found else n)
    live = ro.j + (1 if ro.found else 0)
    par = np.full(live, -1, dtype=np.int64)
    for u, v in T.edges():
        par[v] = u
    assert np.array_equal(par, ro.parent[:live])
    assert np.array_equal(np.array([T.edges[par[v], v]["cost"] for v in range(1, live, 97)]), ro.vcost[1:live:97])


def test_integration_stub_runs():
    """INTEGRATION.md shows the ctypes stub a maintainer of the reference would add to rrtplanner/rrt.py.  Execute that very
    block: bind its _device_plan onto a minimal class that has what the reference's planner has (og, free, n, rand_gen,
    r_rewire, build_graph) and compare with rrtplanner_amd.RRTStar / RRTStandard -- same graph, same goal vertex."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    assert "_device_plan" in code and 'C.CDLL("librrt_hip.so")' in code
    ns = {}
    exec(compile(code.replace('C.CDLL("librrt_hip.so")', f"C.CDLL({_ffi.LIB_PATH!r})"), "INTEGRATION.md", "exec"), ns)

    class Minimal:  # the state the reference's RRT.__init__ leaves behind (rrt.py:59-85)
        def __init__(self, og, n, r_rewire, seed):
            self.og, self.n, self.r_rewire = og, n, r_rewire
            self.free = np.argwhere(og == 0)
            self.rand_gen = np.random.default_rng(seed)

        build_graph = amd.RRT.build_graph

    og = perlin_occupancygrid(300, 300, seed=5)
    xs, xg = random_connected_pair(og, np.random.default_rng(3))
    for alg, cls, kw in ((1, amd.RRTStar, dict(r_rewire=40.5)), (0, amd.RRTStandard, {})):
        m = Minimal(og, 4000, kw.get("r_rewire", 0), seed=9)
        Ts, gs = ns["_device_plan"](m, alg, xs, xg)
        p = cls(og, 4000, pbar=False, seed=9, **kw)
        T, g = p.plan(xs, xg)
        assert gs == g and list(Ts.nodes) == list(T.nodes) and list(Ts.edges) == list(T.edges)
        assert [d["cost"] for _, _, d in Ts.edges(data=True)] == [d["cost"] for _, _, d in T.edges(data=True)]
        assert all(np.array_equal(Ts.nodes[v]["pt"], T.nodes[v]["pt"]) for v in list(T.nodes)[:500])
        assert m.rand_gen.bit_generator.state == p.rand_gen.bit_generator.state
        ns["_lib"].rrt_ctx_destroy(m._ctx)


def test_plan_batch_one_shot(gpu_ctx):
    """rrt_plan_batch (the one-call form of the batch API): 6 mixed RRTStandard / RRTStar queries of different n."""
    og = perlin_occupancygrid(512, 512, seed=2)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    sg = np.random.default_rng(3)
    ns = [2500, 4000, 1, 3999, 64, 4000]
    queries, keep, refs = [], [], []
    for q, n in enumerate(ns):
        xs, xg = random_connected_pair(og, sg)
        samples = hostprep.draw_free_samples(np.random.default_rng(50 + q), free, n)
        alg = q % 2
        r2 = hostprep.radius_threshold(40) if alg else 0
        qu, k = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2)
        queries.append(qu)
        keep.append(k)
        refs.append(oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, logs=False))
    rc, res = gpu_ctx.plan_batch(queries, ns)
    assert rc == next((st for st, _ in refs if st != 0), 0)
    for q, (st, ro) in enumerate(refs):
        live = ro.j + (1 if ro.found else 0)
        assert res[q].status == st and res[q].j == ro.j and res[q].vgoal == ro.vgoal and res[q].rows == ro.rows
        assert np.array_equal(res[q].pts[:live], ro.pts[:live])
        assert np.array_equal(res[q].parent[:live], ro.parent[:live])
        assert np.array_equal(res[q].vcost[:live], ro.vcost[:live])


def test_gather_single_rank_self_test():
    """rrt_comm_init / rrt_gather / rrt_gather_fetch / rrt_comm_allreduce_f64 on a communicator of one rank (all a one-GPU box
    can hold): the gathered slab is the batch's own, self-describing ({status, j, vgoal, found} per query), and fetch returns
    the oracle's trees.  The N > 1 shard / layout / id hand-over logic is covered on CPU (tests/test_dist_gloo.py)."""
    from rrtplanner_amd import multi

    ctx = _ffi.Context(0)
    og = perlin_occupancygrid(256, 256, seed=4)
    og8 = oracle.og_u8(og)
    ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    multi.init_comm(ctx, 0, 1)
    assert ctx.allreduce([1.5, -2.0], "max").tolist() == [1.5, -2.0] and ctx.allreduce([3.0]).tolist() == [3.0]
    ctx.barrier()
    Q, n = 5, 3000
    b = _ffi.Batch(ctx, Q, n)
    sg = np.random.default_rng(9)
    refs, keep = [], []
    for q in range(Q):
        xs, xg = random_connected_pair(og, sg)
        samples = hostprep.draw_free_samples(np.random.default_rng(q), free, n)
        qu, k = _ffi.make_query(1, n, xs, xg, samples, r2_rewire=hostprep.radius_threshold(30))
        keep.append(k)
        b.set_query(q, qu)
        refs.append(oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=hostprep.radius_threshold(30), logs=False))
    b.launch()
    b.sync()
    ptr, nbytes = b.gather()
    assert ptr and nbytes == b.result_block()[1] and nbytes == multi.slab_bytes(Q, ((n + 1 + 4095) // 4096) * 4096)
    for q in range(Q):
        st, ro = refs[q]
        res = b.gather_fetch(0, q)
        live = ro.j + (1 if ro.found else 0)
        assert (res.status, res.j, res.vgoal, res.found) == (st, ro.j, ro.vgoal, ro.found)
        assert np.array_equal(res.pts[:live], ro.pts[:live]) and np.array_equal(res.parent[:live], ro.parent[:live])
        assert np.array_equal(res.vcost[:live], ro.vcost[:live])
    with pytest.raises(_ffi.RRTError):
        b.gather_fetch(1, 0)  # no such rank
    # a fetch never writes more rows than the caller's arrays hold ...
    import ctypes as C

    short = _ffi.ResultArrays(10)
    short.c.rows = 11
    rc = _ffi.lib().rrt_gather_fetch(b._h, 0, 0, C.byref(short.c))
    assert rc == _ffi.RRT_E_ARG and not short.pts.any() and not short.vcost.any()
    # ... and a batch of the same size whose slabs were NOT the ones gathered is refused instead of served another batch's trees
    b2 = _ffi.Batch(ctx, Q, n)
    with pytest.raises(_ffi.RRTError) as e2:
        b2.gather_fetch(0, 0)
    assert e2.value.code == _ffi.RRT_E_COMM
    b2.close()
    assert b.gather_fetch(0, 1).j == refs[1][1].j  # b's own slabs are still served
    b.close()
    ctx.comm_destroy()
    with pytest.raises(_ffi.RRTError) as e:
        ctx.allreduce([1.0])
    assert e.value.code == _ffi.RRT_E_COMM
    ctx.close()


def test_pipelined_informed_batch_has_no_timeouts(gpu_ctx):
    """Informed queries on pipelined teams (blocks in flight are voided when a commit moves the ellipse): the staged API with the
    host's unit-ball hand-over, no hand-off may time out, trees equal the oracle's."""
    og = perlin_occupancygrid(512, 512, seed=2)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    sg = np.random.default_rng(21)
    algs = [2, 1, 2, 0, 2, 1]  # a mixed batch: the Informed-capable kernel also runs the RRTStandard / RRTStar queries
    Q, n, rr, rg = len(algs), 7000, 48, 10
    b = _ffi.Batch(gpu_ctx, Q, n)
    qs = []
    for q in range(Q):
        xs, xg = random_connected_pair(og, sg)
        rng = np.random.default_rng(300 + q)
        st0 = rng.bit_generator.state
        samples = hostprep.draw_free_samples(rng, free, n)
        Cm = hostprep.rotation_to_world_frame(xs, xg)
        qu, keep = _ffi.make_query(algs[q], n, xs, xg, samples, r2_rewire=hostprep.radius_threshold(rr) if algs[q] else 0,
                                   goal_d2=hostprep.goal_threshold(rg), Cmat=Cm)
        b.set_query(q, qu)
        qs.append((xs, xg, samples, Cm, rng, st0, keep))
    b.launch()
    b.sync()
    assert b.pipelined()
    ubs = {}
    for q in range(Q):
        r = b.get_result(q, arrays=False)
        if r.c.status == _ffi.RRT_NEED_UNITBALL:
            xs, xg, samples, Cm, rng, st0, keep = qs[q]
            rng.bit_generator.state = st0
            hostprep.draw_free_samples(rng, free, r.c.i_switch)
            ubs[q] = (hostprep.draw_unitball(rng, n - r.c.i_switch), r.c.i_switch)
            b.set_unitball(q, ubs[q][0], ubs[q][1])
    assert ubs  # at least one query reached its goal region
    b.launch()
    b.sync()
    assert b.team()[1] == 0
    for q in range(Q):
        xs, xg, samples, Cm, rng, st0, keep = qs[q]
        res = b.get_result(q)
        kw = dict(unitball=ubs[q][0], ub_offset=ubs[q][1]) if q in ubs else {}
        st, ro = oracle.plan(og8, n, algs[q], xs, xg, samples, r2_rewire=hostprep.radius_threshold(rr) if algs[q] else 0, r_goal=rg, Cmat=Cm, **kw)
        live = ro.j + (1 if ro.found else 0)
        assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal and res.i_switch == ro.i_switch
        assert np.array_equal(res.pts[:live], ro.pts[:live])
        assert np.array_equal(res.parent[:live], ro.parent[:live])
        assert np.array_equal(res.vcost[:live], ro.vcost[:live])
    b.close()


def test_team_that_loses_a_member_finishes_on_one_cu_per_query(gpu_ctx):
    """RRT_FLAG_TEAM_FAULT: member 1 of every team leaves at once, so a hand-off of the others times out (bounded wait); the
    batch must notice, continue from the consistent block boundary with one CU per query, and give the oracle's trees."""
    og = perlin_occupancygrid(512, 512, seed=4)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    sg = np.random.default_rng(11)
    Q, n = 3, 2500
    b = _ffi.Batch(gpu_ctx, Q, n, team=8, team_fault=True)
    assert b.team() == (8, 0)
    keep, refs = [], []
    for q in range(Q):
        xs, xg = random_connected_pair(og, sg)
        samples = hostprep.draw_free_samples(np.random.default_rng(100 + q), free, n)
        qu, k = _ffi.make_query(1, n, xs, xg, samples, r2_rewire=hostprep.radius_threshold(40))
        keep.append(k)
        b.set_query(q, qu)
        refs.append(oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=hostprep.radius_threshold(40)))
    for rep in range(2):  # the team size stays: the next launch tries the team again (and, with the fault flag, falls back again)
        b.launch()
        b.sync()
        assert b.team() == (8, rep + 1)
        assert b.elapsed_ms() > 400.0  # the launch that timed out (0.5 s bounded wait) is part of the reported time
        for q in range(Q):
            res = b.get_result(q)
            st, ro = refs[q]
            live = ro.j + (1 if ro.found else 0)
            assert res.status == st and res.j == ro.j and res.vgoal == ro.vgoal
            assert np.array_equal(res.pts[:live], ro.pts[:live])
            assert np.array_equal(res.parent[:live], ro.parent[:live])
            assert np.array_equal(res.vcost[:live], ro.vcost[:live])
        b.rearm()
    b.close()


def test_every_team_size_gives_the_same_trees_every_time(gpu_ctx):
    """The same 3-query batch on 1, 2, 4, 8, 16, 32 and as many CUs per query as fit, three launches each: every array of every
    result must be identical (placement, timing and team size must not show; tools/stress_team.py is the long version)."""
    og = perlin_occupancygrid(1024, 1024, seed=1)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    sg = np.random.default_rng(7)
    Q, n = 3, 9000
    qs = []
    for q in range(Q):
        xs, xg = random_connected_pair(og, sg)
        samples = hostprep.draw_free_samples(np.random.default_rng(q), free, n)
        qs.append(_ffi.make_query(1, n, xs, xg, samples, r2_rewire=hostprep.radius_threshold(64)))
    ref = None
    for team, pipe in ((None, True), (None, False), (32, True), (16, True), (16, False), (8, True), (4, True), (4, False), (3, True), (2, True),
                       (2, False), (1, True)):
        b = _ffi.Batch(gpu_ctx, Q, n, team=team, pipe=pipe)
        for q, (qu, keep) in enumerate(qs):
            b.set_query(q, qu)
        for rep in range(3):
            b.rearm()
            b.launch()
            b.sync()
            out = []
            for q in range(Q):
                r = b.get_result(q)
                live = r.j + (1 if r.found else 0)
                out.append((r.status, r.j, r.vgoal, r.pts[:live].copy(), r.parent[:live].copy(), r.vcost[:live].copy()))
            if ref is None:
                ref = out
            for q in range(Q):
                assert out[q][:3] == ref[q][:3], (team, pipe, rep, q)
                assert all(np.array_equal(x, y) for x, y in zip(out[q][3:], ref[q][3:])), (team, pipe, rep, q)
        assert b.team()[1] == 0
        b.close()


def test_maximum_capacity_on_the_largest_grid(gpu_ctx):
    """The limits of this path: n = 262143 samples on a 2048 x 2048 grid, RRT*, default team -- node indices use
    all 18 bits, the scan key all 64 chunk tags, squared distances all 23 bits.  Whole tree equal to the oracle's; one sample
    more is refused.  (tools/archive/big_case.py runs the other planners and one CU at this size.)"""
    og = perlin_occupancygrid(2048, 2048, seed=3)
    og8 = oracle.og_u8(og)
    gpu_ctx.set_grid(og8)
    xs, xg = random_connected_pair(og, np.random.default_rng(11))
    n = 262143
    samples = hostprep.draw_free_samples(np.random.default_rng(5), np.argwhere(og8 == 0), n)
    r2 = hostprep.radius_threshold(64)
    q, keep = _ffi.make_query(1, n, xs, xg, samples, r2_rewire=r2)
    rc, res = gpu_ctx.plan(q, n)
    st, ro = oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=r2, logs=False)
    live = ro.j + (1 if ro.found else 0)
    assert rc == st and res.j == ro.j and res.vgoal == ro.vgoal and ro.j > 250000
    assert np.array_equal(res.pts[:live], ro.pts[:live]) and np.array_equal(res.parent[:live], ro.parent[:live])
    assert np.array_equal(res.vcost[:live], ro.vcost[:live])
    assert res.sum_j == ro.sum_j and res.sum_near == ro.sum_near
    with pytest.raises(_ffi.RRTError) as e:
        _ffi.Batch(gpu_ctx, 1, n + 1)
    assert e.value.code == _ffi.RRT_E_UNSUPPORTED


# ------------------------------------------------------------------------------- properties at full size
def test_full_size_properties_rrtstar_1024_n50000():
    """BASELINE config 2 (RRT*, 1024x1024, n=50000): size-independent properties of the result --
    every edge has a clear line of sight, cost[child] == cost[parent] + dist (no rewire ever fires),
    no duplicate nodes, every node is a free cell, parents precede children, the path reaches the goal."""
    og = perlin_occupancygrid(1024, 1024, seed=1)
    xs, xg = random_connected_pair(og, np.random.default_rng(7))
    n = 50000
    p = amd.RRTStar(og, n, 64, pbar=False, seed=0)
    res = p._run(_ffi.ALG_STAR, xs, xg, r_rewire=64)
    j, live = res.j, res.j + 1
    assert res.found and res.vgoal == j
    pts, par, vc = res.pts[:live].astype(np.int64), res.parent[:live].astype(np.int64), res.vcost[:live]
    assert par[0] == -1 and np.all(par[1:] >= 0) and np.all(par[1:] < np.arange(1, live))
    assert np.all(og[pts[:, 0], pts[:, 1]] == 0)
    assert len({(a, b) for a, b in pts[:j].tolist()}) >= j - 1  # only xstart may repeat once (rrt.py:425)
    d = pts[1:] - pts[par[1:]]
    dist = np.sqrt((d * d).sum(1).astype(np.float64))
    assert np.array_equal(vc[1:], vc[par[1:]] + dist)
    og8 = oracle.og_u8(og)
    ctx = p._device()
    free, _ = ctx.prim_collisionfree(np.concatenate([pts[par[1:]], pts[1:]], axis=1).astype(np.int32))
    assert free.all()
    # accepted nodes are exactly a subsequence of the sample stream
    samples = hostprep.draw_free_samples(np.random.default_rng(0), np.argwhere(og == 0), n)
    it = iter(map(tuple, samples.tolist()))
    assert all(any(s == tuple(v) for s in it) for v in pts[1:j].tolist())
    T, gv = p._materialise(res)
    path = p.route2gv(T, gv)
    assert path[0] == 0 and path[-1] == gv and np.array_equal(T.nodes[gv]["pt"], xg)
    # and the oracle agrees on the whole tree
    st, ro = oracle.plan(og8, n, 1, xs, xg, samples, r2_rewire=hostprep.radius_threshold(64), logs=False)
    assert ro.j == j and np.array_equal(ro.pts[:live], res.pts[:live]) and np.array_equal(ro.parent[:live], res.parent[:live])
    assert np.array_equal(ro.vcost[:live], vc)


@pytest.mark.parametrize("w,h,frames,seed", [(200, 200, 1, 1), (1024, 1024, 1, 1), (96, 160, 5, 3), (2048, 2048, 1, 3), (333, 77, 3, 9)])
def test_device_noise_grids_equal_host_generator(w, h, frames, seed):
    """Grids generated on the device are bit-identical to the host generator."""
    from rrtplanner_amd.oggen import DeviceGrids

    ctx = _ffi.Context(0)
    g = DeviceGrids(ctx, w, h, thresh=0.33, frames=frames, seed=seed)
    want = perlin_occupancygrid(w, h, thresh=0.33, frames=frames if frames > 1 else None, seed=seed)
    assert np.array_equal(g.host if frames > 1 else g.host[0], want)
    ctx.close()


def test_replanning_on_resident_frames_equals_uploaded_grids():
    """anim.py:92-93 pattern: set_og per frame, plan; with DeviceGrids the frames never leave the device."""
    from rrtplanner_amd.oggen import DeviceGrids

    frames = perlin_occupancygrid(160, 160, thresh=0.33, frames=4, seed=5)
    xs, xg = random_connected_pair(frames[0], np.random.default_rng(2))
    a = amd.RRTStar(frames[0], 800, 24, pbar=False, seed=3)
    b = amd.RRTStar(frames[0], 800, 24, pbar=False, seed=3)
    grids = DeviceGrids(b.device_context(), 160, 160, thresh=0.33, frames=4, seed=5)
    for k in range(4):
        if frames[k][xs[0], xs[1]] or frames[k][xg[0], xg[1]]:
            continue
        a.set_og(frames[k])
        b.set_og_resident(grids, k)
        try:
            Ta, ga = a.plan(xs, xg)
        except IndexError:
            with pytest.raises(IndexError):
                b.plan(xs, xg)
            continue
        Tb, gb = b.plan(xs, xg)
        assert ga == gb and list(Ta.edges) == list(Tb.edges)
        assert all(np.array_equal(Ta.nodes[v]["pt"], Tb.nodes[v]["pt"]) for v in Ta.nodes)
        assert [d["cost"] for _, _, d in Ta.edges(data=True)] == [d["cost"] for _, _, d in Tb.edges(data=True)]


def test_resident_frames_are_invalidated_by_any_other_upload():
    """set_og_resident -> set_og + plan -> set_og_resident must not silently plan on the other grid."""
    from rrtplanner_amd.oggen import DeviceGrids

    frames = perlin_occupancygrid(96, 96, thresh=0.33, frames=3, seed=5)
    other = perlin_occupancygrid(96, 96, thresh=0.33, seed=8)
    xs, xg = random_connected_pair(frames[1], np.random.default_rng(2))
    p = amd.RRTStar(frames[0], 300, 16, pbar=False, seed=1)
    grids = DeviceGrids(p.device_context(), 96, 96, thresh=0.33, frames=3, seed=5)
    p.set_og_resident(grids, 1)
    assert grids.valid()
    p.plan(xs, xg)
    p.set_og(other)  # an upload replaces the resident frames at the next plan()
    try:
        p.plan(xs, xg)
    except IndexError:
        pass
    assert not grids.valid()
    for k in (0, 1, 2):
        with pytest.raises(RuntimeError):
            p.set_og_resident(grids, k)
    assert p.og is other  # the failed switch left the planner on the uploaded grid
    # a second DeviceGrids on the same context evicts the first one as well
    g2 = DeviceGrids(p.device_context(), 96, 96, thresh=0.33, frames=2, seed=6)
    g3 = DeviceGrids(p.device_context(), 96, 96, thresh=0.33, frames=2, seed=7)
    assert g3.valid() and not g2.valid()
    with pytest.raises(RuntimeError):
        p.set_og_resident(g2, 0)
    p.set_og_resident(g3, 1)
    assert np.array_equal(p.og, g3.host[1])


def test_missing_grid_and_bad_arguments_fail_loudly():
    ctx = _ffi.Context(0)
    with pytest.raises(_ffi.RRTError):
        _ffi.Batch(ctx, 1, 10)  # no grid yet
    ctx.set_grid(np.zeros((8, 8), dtype=np.uint8))
    q, keep = _ffi.make_query(0, 4, (0, 0), (9, 9), np.zeros((4, 2), dtype=np.int32))
    with pytest.raises(_ffi.RRTError):
        ctx.plan(q, 4)  # goal outside the grid
    ctx.set_grid(np.zeros((4096, 8), dtype=np.uint8))  # fine for the primitives / the host-driven path (up to 32767 cells per axis) ...
    with pytest.raises(_ffi.RRTError) as e:
        _ffi.Batch(ctx, 1, 10)  # ... but beyond what the expansion kernels take
    assert e.value.code == _ffi.RRT_E_UNSUPPORTED
    with pytest.raises(_ffi.RRTError):
        ctx.set_grid(np.zeros((32768, 2), dtype=np.uint8))
    ctx.close()


def test_custom_costfn_runs_on_the_device_primitives():
    """A custom cost function no longer raises: the loop stays in Python, the device answers its questions (rrt_tree_query);
    tests/test_costfn.py compares 40 such plans with the real reference.  Here: the callable really is called, per candidate."""
    og = np.zeros((20, 20), dtype=int)
    calls = []

    def cost(vc, pts, v, x):
        calls.append(int(v))
        return vc[v] + 1.0  # every edge costs 1: the cost of a vertex is its depth

    p = amd.RRTStar(og, 60, 6, costfn=cost, pbar=False)
    T, gv = p.plan(np.array([1, 1]), np.array([15, 15]))
    assert len(calls) > 200 and T.number_of_nodes() == 61
    depth = {0: 0}
    for u, v, d in sorted(T.edges(data=True), key=lambda e: e[1]):
        assert d["cost"] == depth.setdefault(u, 0) + 1.0 or v == gv
        depth[v] = d["cost"]


def test_device_tree_vertices_are_checked_against_the_grid():
    """every vertex of a host-driven planner's tree later starts a line-of-sight walk over the context's grid, so
    rrt_tree_append refuses a vertex outside THAT grid (not just outside the 15-bit range), and a query after the grid changed
    shape is refused until the tree is reset."""
    ctx = _ffi.Context(0)
    tr = _ffi.DeviceTree(ctx, 16)
    with pytest.raises(_ffi.RRTError) as e:
        tr.append(1, 1)  # no grid yet
    assert e.value.code == _ffi.RRT_E_NOGRID
    ctx.set_grid(np.zeros((40, 30), dtype=np.uint8))
    assert tr.append(5, 5) == 0 and tr.append(39, 29) == 1
    for bad in ((40, 5), (5, 30), (-1, 0), (1000, 1000)):
        with pytest.raises(_ffi.RRTError) as e:
            tr.append(*bad)
        assert e.value.code == _ffi.RRT_E_ARG
    nn, idx, los, _ = tr.query(6, 6, 10 ** 6)
    assert nn == 0 and idx.tolist() == [0, 1] and los
    ctx.set_grid(np.zeros((20, 20), dtype=np.uint8))  # vertex 1 = (39, 29) now lies outside
    with pytest.raises(_ffi.RRTError) as e:
        tr.query(6, 6, 100)
    assert e.value.code == _ffi.RRT_E_ARG
    with pytest.raises(_ffi.RRTError):
        tr.append(3, 3)
    tr.reset()
    assert tr.append(3, 3) == 0 and tr.query(4, 4, 100)[0] == 0
    tr.close()
    ctx.close()


def test_context_close_takes_its_batches_along():
    """Closing a context first closes the batches created on it (they hold a pointer to it); closing them again is harmless."""
    ctx = _ffi.Context(0)
    ctx.set_grid(np.zeros((32, 32), dtype=np.uint8))
    b1, b2 = _ffi.Batch(ctx, 2, 100), _ffi.Batch(ctx, 1, 50, team=1)
    ctx.close()
    b1.close()
    b2.close()
    del b1, b2, ctx
