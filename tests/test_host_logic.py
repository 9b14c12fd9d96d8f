"""CPU: host-side logic of the drop-in (RNG stream handling, thresholds, grid generator, statics)."""
import math

import numpy as np
import pytest

from rrtplanner_amd import hostprep, multi, oggen
from rrtplanner_amd import rrt as amd


def test_rng_block_draws_equal_scalar_draws_and_state():
    F = 700001
    free = np.stack([np.arange(F) % 1000, np.arange(F) // 1000], axis=1)
    g1, g2 = np.random.default_rng(3), np.random.default_rng(3)
    a = np.array([free[g1.choice(F)] for _ in range(1237)])
    b = hostprep.draw_free_samples(g2, free, 1237)
    assert np.array_equal(a, b) and g1.bit_generator.state == g2.bit_generator.state
    # unit ball: scalar sequence of the reference (rrt.py:582-586) vs one block
    u1 = []
    for _ in range(501):
        r = g1.uniform(0, 1)
        th = 2 * np.pi * g1.uniform(0, 1)
        u1.append([np.sqrt(r) * np.cos(th), np.sqrt(r) * np.sin(th)])
    u2 = hostprep.draw_unitball(g2, 501)
    assert np.array_equal(np.array(u1), u2) and g1.bit_generator.state == g2.bit_generator.state
    # rewind to a prefix: same state as drawing only the prefix
    g3 = np.random.default_rng(9)
    s0 = g3.bit_generator.state
    full = hostprep.draw_free_samples(g3, free, 4000)
    g3.bit_generator.state = s0
    part = hostprep.draw_free_samples(g3, free, 1501)
    g4 = np.random.default_rng(9)
    [g4.choice(F) for _ in range(1501)]
    assert np.array_equal(full[:1501], part) and g3.bit_generator.state == g4.bit_generator.state


@pytest.mark.parametrize("r", [0, 1, 2, 5, 7.5, 10, 31.999, 32, 64, 64.5, 1e9, float("inf"), float("nan"), -3, np.float32(12.5), np.int64(40)])
def test_radius_threshold_matches_numpy_comparison(r):
    R = hostprep.radius_threshold(r)
    d2 = np.arange(0, 5000, dtype=np.int64)
    with np.errstate(all="ignore"):
        want = d2 < r * r
    assert np.array_equal(d2 < R, want)


@pytest.mark.parametrize("rg", [0, -1, 0.5, 1, 1.0000001, 5, 5.0, 12, 12.5, math.sqrt(50), math.nextafter(math.sqrt(50), 0), 2900.0, 1e9, float("nan")])
def test_goal_threshold_matches_r2norm_comparison(rg):
    G = hostprep.goal_threshold(rg)
    for d2 in list(range(0, 400)) + [2499, 2500, 2501, 8410000 - 1, 8410000, 8410001]:
        assert (math.sqrt(d2) < rg) == (d2 < G), (rg, d2, G)


def test_noise_grid_contract_and_determinism():
    a = oggen.perlin_occupancygrid(120, 90, seed=5)
    b = oggen.perlin_occupancygrid(120, 90, seed=5)
    c = oggen.perlin_occupancygrid(120, 90, seed=6)
    assert a.shape == (120, 90) and a.dtype.kind == "i" and set(np.unique(a)) <= {0, 1}
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    f = oggen.perlin_occupancygrid(40, 30, frames=3, seed=1)
    assert f.shape == (3, 40, 30)
    assert 0.5 < (a == 0).mean() < 0.95
    m = oggen.largest_free_component(a)
    xs, xg = oggen.random_connected_pair(a, np.random.default_rng(0))
    assert m[xs[0], xs[1]] and m[xg[0], xg[1]]


def test_class_surface_matches_reference():
    og = np.zeros((30, 20), dtype=int)
    og[10:12, 5:15] = 1
    p = amd.RRT(og, 17)
    assert p.n == 17 and p.pbar is True and p.not_a_point == [np.inf, np.inf] and p.not_a_dist == np.inf
    assert np.array_equal(p.free, np.argwhere(og == 0)) and p.og is og
    with pytest.raises(NotImplementedError):
        p.plan(np.array([0, 0]), np.array([1, 1]))
    p.set_n(5)
    assert p.n == 5
    og2 = np.zeros((8, 8), dtype=int)
    p.set_og(og2)
    assert p.og is og2 and p.free.shape[0] == 64 and p._grid_dirty
    s = p.sample_all_free()
    assert s.shape == (2,) and og2[s[0], s[1]] == 0
    assert amd.RRTStar(og, 5, 3.5).r_rewire == 3.5
    q = amd.RRTStarInformed(og, 5, 3, 2)
    assert q.r_goal == 2 and q.ellipses == {}
    # default cost closure (rrt.py:72-78)
    vc = np.array([0.0, 2.5])
    pts = np.array([[0, 0], [3, 4]])
    assert q.cost(vc, pts, 1, np.array([0, 0])) == 7.5
    assert amd.r2norm(np.array([3, 4])) == 5.0 and isinstance(amd.r2norm(np.array([3, 4])), float)
    rp = amd.random_point_og(og, np.random.default_rng(1))
    assert og[rp[0], rp[1]] == 0
    # same seed -> same stream as the reference's self.rand_gen (rrt.py:85)
    assert amd.RRTStandard(og, 5, seed=4).rand_gen.integers(0, 1 << 30) == np.random.default_rng(4).integers(0, 1 << 30)


def test_plan_argument_validation_without_device():
    og = np.zeros((10, 10), dtype=int)
    p = amd.RRTStandard(og, 5, pbar=False)
    with pytest.raises(ValueError):
        p._run(0, np.array([0, 0]), np.array([10, 3]))
    with pytest.raises(ValueError):
        p._run(0, np.array([0.5, 0]), np.array([1, 3]))
    # a custom cost function keeps the loop on the host (hostloop.py) but validates the same way, and never runs without a device
    q = amd.RRTStandard(og, 5, costfn=lambda *a: 0.0, pbar=False)
    with pytest.raises(ValueError):
        q.plan(np.array([0, 0]), np.array([10, 3]))
    with pytest.raises(amd._ffi.RRTError):  # no GPU in this test run: no silent CPU path
        q.plan(np.array([0, 0]), np.array([1, 1]))


def test_sharding_helpers():
    for total, ws in [(512, 8), (7, 3), (1, 4), (64, 1)]:
        seen = []
        for r in range(ws):
            mine = multi.shard_queries(total, ws, r)
            assert all(multi.owner_of(q, ws) == r for q in mine)
            assert [multi.local_slot(q, ws) for q in mine] == list(range(len(mine)))
            seen += mine
        assert sorted(seen) == list(range(total))
    with pytest.raises(ValueError):
        multi.shard_queries(4, 2, 2)
    Q, stride = 3, 8
    slab = np.zeros(multi.slab_bytes(Q, stride), dtype=np.uint8)
    slab[-16:-12] = np.array([77], dtype=np.int32).view(np.uint8)
    v, nd, pa, meta = multi.unpack_slab(slab, Q, stride)
    assert v.shape == (Q, stride) and nd.dtype == np.uint32 and pa.dtype == np.int32 and meta.shape == (Q, 4) and meta[2, 0] == 77
    with pytest.raises(ValueError):
        multi.unpack_slab(slab[:-1], Q, stride)


def test_build_graph_equals_call_by_call_construction():
    """The direct adjacency fill of build_graph == the reference's add_node / add_edge sequence (rrt.py:357-369)."""
    import networkx as nx

    rng = np.random.default_rng(0)
    n, live = 60, 41
    points = np.full((n + 1, 2), hostprep.INT64_MIN, dtype=np.int64)
    points[:live] = rng.integers(0, 100, size=(live, 2))
    points[n] = points[live - 1]
    vcosts = np.full(n + 1, np.inf)
    vcosts[:live] = rng.uniform(0, 50, size=live)
    parents = {0: None}
    for c in range(1, live):
        parents[c] = int(rng.integers(0, c))
    vgoal = live - 1
    p = amd.RRT(np.zeros((100, 100), dtype=int), n)
    T = p.build_graph(vgoal, points, parents, vcosts)
    R = nx.DiGraph()
    R.add_node(vgoal, pt=points[vgoal])
    for i, q in enumerate(points):
        R.add_node(i, pt=q)
    for child, parent in parents.items():
        if parent is not None:
            d = points[child] - points[parent]
            R.add_edge(parent, child, dist=math.sqrt(d[0] * d[0] + d[1] * d[1]), cost=vcosts[child])
    assert list(T.nodes) == list(R.nodes) and list(T.edges) == list(R.edges)
    for v in R.nodes:
        assert np.array_equal(T.nodes[v]["pt"], R.nodes[v]["pt"])
        assert list(T.pred[v]) == list(R.pred[v]) and list(T.succ[v]) == list(R.succ[v])
    for u, v, d in R.edges(data=True):
        assert T.edges[u, v] == d and type(T.edges[u, v]["dist"]) is float and isinstance(T.edges[u, v]["cost"], np.float64)
        assert T.pred[v][u] is T.succ[u][v]
    assert nx.shortest_path(T, 0, vgoal, weight="dist") == nx.shortest_path(R, 0, vgoal, weight="dist")
    assert T.number_of_edges() == R.number_of_edges() and T.in_degree(vgoal) == R.in_degree(vgoal)


def test_route2gv_parent_walk_equals_dijkstra():
    """route2gv reads the unique root path off the parent pointers of a planner tree and must agree with the reference's
    Dijkstra (rrt.py:87-107); a graph that is not such a tree goes through networkx."""
    import networkx as nx

    rng = np.random.default_rng(3)
    T = nx.DiGraph()
    n = 400
    for v in range(n):
        T.add_node(v, pt=rng.integers(0, 50, 2))
    for v in range(1, n - 20):  # the last 20 vertices stay isolated like the reference's sentinel rows
        T.add_edge(int(rng.integers(0, v)), v, dist=float(rng.random()), cost=0.0)
    p = amd.RRTStandard.__new__(amd.RRTStandard)
    for gv in (0, 1, 17, n - 21):
        assert p.route2gv(T, gv) == nx.shortest_path(T, source=0, target=gv, weight="dist")
    with pytest.raises(nx.NetworkXNoPath):
        p.route2gv(T, n - 1)
    T.add_edge(3, 17, dist=1e-9, cost=0.0)  # a second parent: no longer a tree
    assert p.route2gv(T, 17) == nx.shortest_path(T, source=0, target=17, weight="dist")


def test_tree_digraph_is_lazy_and_equal_to_build_graph():
    """plan() returns a TreeDiGraph: route2gv / vertices_as_ndarray work from the result arrays, and the first touch of the
    graph itself fills exactly what build_graph (reference rrt.py:334-369) builds."""
    import pickle

    import networkx as nx

    rng = np.random.default_rng(5)
    n, live = 300, 260
    points = np.full((n + 1, 2), hostprep.INT64_MIN, dtype=np.int64)
    points[:live] = rng.integers(0, 90, (live, 2))
    points[n] = points[7]
    vcosts = np.full(n + 1, np.inf)
    vcosts[:live] = rng.random(live)
    vcosts[n] = vcosts[7]
    parent = np.array([-1] + [int(rng.integers(0, c)) for c in range(1, live)], dtype=np.int64)
    vgoal = live - 1
    p = amd.RRTStandard.__new__(amd.RRTStandard)
    T = amd.TreeDiGraph.from_arrays(vgoal, points, parent, vcosts)
    assert isinstance(T, nx.DiGraph) and T.lazy_points() is not None
    path = p.route2gv(T, vgoal)
    segs = p.vertices_as_ndarray(T, path)
    assert T.lazy_points() is not None  # still only arrays
    ref = p.build_graph(vgoal, points, {0: None, **{c: int(parent[c]) for c in range(1, live)}}, vcosts)
    assert path == nx.shortest_path(ref, 0, vgoal, weight="dist")
    assert np.array_equal(segs, np.array([[points[a], points[b]] for a, b in zip(path[:-1], path[1:])]))
    T2 = pickle.loads(pickle.dumps(T))  # lazy state survives pickling
    assert T.number_of_nodes() == ref.number_of_nodes() and T.lazy_points() is None
    for G in (T, T2, T.copy()):
        assert list(G.nodes) == list(ref.nodes) and list(G.edges) == list(ref.edges)
        assert all(G.edges[e] == ref.edges[e] for e in ref.edges)
        assert all(np.array_equal(G.nodes[v]["pt"], ref.nodes[v]["pt"]) for v in ref.nodes)
    assert p.route2gv(T, vgoal) == path  # and after materialisation
    T.add_edge(3, 9999, dist=1.0)  # an ordinary DiGraph from here on
    assert T.has_edge(3, 9999)


def test_informed_sampler_helpers_match_the_reference_tables():
    """The public sampler helpers of RRTStarInformed (rrt.py:579-651) on the host, against the reference's 2400-row
    sample_ellipse table and rotation matrices (tests/golden/primitives.npz)."""
    import orchelp

    P = orchelp.golden("primitives.npz").z
    W, H = P["ell_WH"]
    p = amd.RRTStarInformed(np.zeros((int(W), int(H)), dtype=int), 10, 10, 5, pbar=False, seed=0)
    for s, Cm in zip(P["rot_sg"], P["rot_C"]):
        assert np.array_equal(p.rotation_to_world_frame(s[:2].astype(np.int64), s[2:].astype(np.int64)), Cm)
    for row, want in zip(P["ell_in"][::7], P["ell_out"][::7]):
        xs, xg = row[:2].astype(np.int64), row[2:4].astype(np.int64)
        p.unitball = lambda u=row[5:7]: u
        assert p.sample_ellipse(xs, xg, row[4]).tolist() == want.tolist()
    del p.unitball
    # unitball(): two scalar uniform draws, like rrt.py:582-586
    g = np.random.default_rng(0)
    r, th = g.uniform(0, 1), 2 * np.pi * g.uniform(0, 1)
    assert np.array_equal(p.unitball(), np.array([np.sqrt(r) * np.cos(th), np.sqrt(r) * np.sin(th)]))
    assert p.rand_gen.bit_generator.state == g.bit_generator.state
    # get_ellipse_for_plt vs the vectorised form plan() uses for self.ellipses
    xs, xg = np.array([10, 10]), np.array([250, 150])
    Cm = p.rotation_to_world_frame(xs, xg)
    xc, ma, mi, an = p.get_ellipse_for_plt(xs, xg, 300.0)
    xc2, ma2, mi2, an2 = hostprep.ellipse_plot_params(Cm, xs, xg, np.array([300.0]))
    assert np.array_equal(xc, xc2) and np.allclose([ma, mi, an], [ma2[0], mi2[0], an2[0]], rtol=1e-12)
    assert amd.RRTStarInformed.least_cost(np.array([5.0, 1.0, 1.0, 3.0]), [0, 2, 1, 3]) == (2, 1.0)
    assert amd.RRTStarInformed.least_cost(np.array([5.0, 1.0]), [1]) == (1, 1.0)
    assert amd.RRTStarInformed.rad2deg(np.pi) == 180.0


def test_host_go2goal_helper_follows_the_reference_contract():
    """RRT.go2goal (rrt.py:284-332) as a host helper: cheapest visible vertex, arrays grown by one row, vgoal = j."""
    from collections import defaultdict

    og = np.zeros((20, 20), dtype=int)
    og[10, 2:20] = 1  # a wall with a gap at y < 2
    p = amd.RRTStandard(og, 4, pbar=False)
    points = np.array([[2, 10], [5, 10], [9, 1], [2, 2]], dtype=np.int64)
    vcosts = np.array([0.0, 3.0, 12.0, 8.0])
    children, parents = defaultdict(list), {0: None, 1: 0, 2: 1, 3: 0}
    vgoal, children, parents, pts2, vc2 = p.go2goal(vcosts, points, np.array([15, 1]), 4, children, parents)
    assert vgoal == 4 and parents[4] == 2 and children[2] == [4]  # only vertex 2 sees the goal through the gap
    assert pts2.shape == (5, 2) and pts2[4].tolist() == [15, 1] and vc2[4] == 12.0 + 6.0
    # nothing sees the goal: vgoal = 0 (rrt.py:330-331), arrays unchanged
    og[10, :] = 1
    vgoal, _, _, pts3, vc3 = p.go2goal(vcosts, points, np.array([15, 1]), 4, defaultdict(list), {0: None})
    assert vgoal == 0 and pts3.shape == (4, 2) and vc3.shape == (4,)


def test_unknown_networkx_layout_falls_back_to_public_calls(monkeypatch):
    """TreeDiGraph / the fast build_graph write networkx's private dictionaries; when a scratch graph does not look
    the way they assume, both go through add_node / add_edge and give the same graph."""
    import networkx as nx

    assert amd._networkx_layout_ok() and amd._NX_FAST  # the networkx of this image (3.x); the reference pins 2.6.3
    points = np.array([[1, 1], [4, 5], [7, 1], [9, 9], [amd.INT64_MIN, amd.INT64_MIN]], dtype=np.int64)
    vcosts = np.array([0.0, 5.0, 6.0, 13.0, np.inf])
    parent = np.array([-1, 0, 0, 1])
    fast = amd.TreeDiGraph.from_arrays(3, points, parent, vcosts)
    monkeypatch.setattr(amd, "_NX_FAST", False)
    slow = amd.TreeDiGraph.from_arrays(3, points, parent, vcosts)
    assert type(slow) is nx.DiGraph and isinstance(fast, amd.TreeDiGraph)
    assert list(slow.nodes) == list(fast.nodes) == [3, 0, 1, 2, 4]
    assert list(slow.edges(data=True)) == list(fast.edges(data=True))
    assert all(np.array_equal(slow.nodes[v]["pt"], fast.nodes[v]["pt"]) for v in fast.nodes)
    p = amd.RRTStandard(np.zeros((10, 10), dtype=int), 4, pbar=False)
    assert p.route2gv(slow, 3) == p.route2gv(fast, 3) == [0, 1, 3]
    g = p.build_graph(3, points, {0: None, 1: 0, 2: 0, 3: 1}, vcosts)  # the public method, also on the slow path now
    assert list(g.edges(data=True)) == list(fast.edges(data=True))


@pytest.mark.parametrize("tag", ["std", "star", "inf"])
def test_general_path_has_the_kernels_semantics(tag):
    """Problems beyond the expansion kernels' limits (grids over 2048 x 2048, n over 262143) run the host-driven loop with the
    default cost in numpy form (hostloop.py).  Forced onto a normal-size problem here, with the numpy stand-in for the device
    primitives, it must give what the kernels' semantics (the oracle) give: graph, goal vertex, costs, generator state, ellipses."""
    import oracle
    import orchelp
    from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

    og = perlin_occupancygrid(200, 200, seed=1)
    og8 = oracle.og_u8(og)
    xs, xg = random_connected_pair(og, np.random.default_rng(3))
    cls, kw = {"std": (amd.RRTStandard, {}), "star": (amd.RRTStar, dict(r_rewire=25.5)), "inf": (amd.RRTStarInformed, dict(r_rewire=30, r_goal=10))}[tag]
    p = cls(og, 1500, pbar=False, seed=4, **kw)
    assert not p._beyond_the_kernels()
    p._costfn_provider = orchelp.NumpyProvider(og8)
    p._beyond_the_kernels = lambda: True
    T, gv = p.plan(xs, xg)
    q = orchelp.use_oracle(cls(og, 1500, pbar=False, seed=4, **kw))
    To, go = q.plan(xs, xg)
    assert gv == go and list(T.nodes) == list(To.nodes) and list(T.edges) == list(To.edges)
    assert [d["cost"] for *_, d in T.edges(data=True)] == [d["cost"] for *_, d in To.edges(data=True)]
    assert p.rand_gen.bit_generator.state == q.rand_gen.bit_generator.state
    if tag == "inf":
        assert list(p.ellipses) == list(q.ellipses) and len(p.ellipses) > 10
    big = amd.RRTStar(np.zeros((2049, 10), dtype=int), 10, 5, pbar=False)
    assert big._beyond_the_kernels() and amd.RRTStar(np.zeros((10, 10), dtype=int), 262144, 5, pbar=False)._beyond_the_kernels()
