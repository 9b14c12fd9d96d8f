"""Planners with a caller-supplied cost function (reference rrt.py:55, :70-80) against goldens from the REAL reference
(tests/golden/make_golden_costfn.py): the loop stays on the host with the device answering near()[0] / within() / collisionfree
once per iteration (rrtplanner_amd/hostloop.py, rrt_tree_query).  With `discount` / `downhill` the reference's rewire block really
fires -- stale costs, vertices dropped from child lists, the swallowed (RRTStar) or escaping (RRTStarInformed) ValueError of a
second rewire -- and all of it has to come out the same: points, parents in dict order, costs bit for bit, goal vertex, generator
state, the graph, the exception.

CPU: the loop with a numpy stand-in for the device primitives.  GPU: the same cases on the device primitives, and the
primitives themselves against numpy."""
import numpy as np
import pytest

import oracle
import orchelp
from costfns import COSTFNS
from rrtplanner_amd import rrt as amd

G = orchelp.golden("plans_costfn_A.npz")


def _run_case(meta, provider_of):
    og8 = G.grid(meta["grid"])
    og = og8.astype(np.int64)
    p = orchelp.make_planner(amd, meta, og, costfn=COSTFNS[meta["costfn"]])
    prov = provider_of(p, og8)
    if prov is not None:
        p._costfn_provider = prov
    xs, xg = np.array(meta["xstart"]), np.array(meta["xgoal"])
    if meta.get("raises"):
        with pytest.raises({"ValueError": ValueError, "IndexError": IndexError}[meta["raises"]]):
            p.plan(xs, xg)
        assert orchelp.rng_state_tuple(p.rand_gen) == meta["rng_state"], "the generator stopped elsewhere than the reference's"
        return p
    T, gv = p.plan(xs, xg)
    assert type(gv).__name__ == meta["gv_type"]
    orchelp.check_plan_against_golden(G, meta, p, T, gv)
    return p


@pytest.mark.parametrize("meta", G.manifest, ids=[m["id"] for m in G.manifest])
def test_host_loop_equals_the_reference(meta):
    """The loop with the numpy stand-in for the device primitives."""
    if meta["n"] > 600 and meta["costfn"] == "manhattan":
        pytest.skip("covered by the GPU run (the numpy stand-in is slow)")
    p = _run_case(meta, lambda p, og8: orchelp.NumpyProvider(og8))
    assert p._costfn_provider.queries > 0


def test_rewires_really_fire_in_the_goldens():
    """What the default cost can never do (rrt.py:536 is never true with it): parents younger than their children."""
    fired = [m for m in G.manifest if m.get("n_rewired_visible", 0) > 0]
    raised = [m for m in G.manifest if m.get("raises") == "ValueError"]
    assert len(fired) >= 10 and len(raised) >= 6 and all(m["alg"] == 2 for m in raised)
    assert all(m.get("n_rewired_visible", 0) == 0 for m in G.manifest if m["costfn"] == "manhattan")


def test_custom_cost_and_the_opt_in_rewire_do_not_mix():
    og = G.grid("corner40").astype(np.int64)
    p = amd.RRTStar(og, 50, 10, costfn=COSTFNS["discount"], pbar=False, rewire="correct")
    p._costfn_provider = orchelp.NumpyProvider(G.grid("corner40"))
    with pytest.raises(ValueError):
        p.plan(np.array((2, 3)), np.array((35, 36)))


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("meta", G.manifest, ids=[m["id"] for m in G.manifest])
def test_device_primitives_under_the_host_loop_equal_the_reference(meta):
    _run_case(meta, lambda p, og8: None)  # the planner's own DeviceProvider


@pytest.mark.gpu
def test_tree_query_equals_numpy(gpu_ctx):
    """rrt_tree_query against numpy / the oracle's line walk: nearest with ties, ascending within lists (short and longer than
    the first copy of 256), every line of sight, appends in between."""
    from rrtplanner_amd import _ffi

    rng = np.random.default_rng(5)
    og8 = (rng.uniform(size=(300, 200)) < 0.12).astype(np.uint8)
    gpu_ctx.set_grid(og8)
    free = np.argwhere(og8 == 0)
    tree = _ffi.DeviceTree(gpu_ctx, 6000)
    ref = orchelp.NumpyProvider(og8)
    for rep in range(2):
        tree.reset()
        ref.reset()
        for k in range(5000):
            x, y = free[rng.integers(0, free.shape[0] if k % 7 else 40)]  # (every 7th from a handful of cells: ties)
            assert tree.append(x, y) == ref.append(x, y) == k
            if k % 250 == 0 or k < 5:
                for r2 in (0, 1, 400, 3000, 10 ** 6):
                    q = free[rng.integers(0, free.shape[0])]
                    nn, idx, f0, fl = tree.query(q[0], q[1], r2)
                    nn2, idx2, g0, gl = ref.query(q[0], q[1], r2)
                    assert nn == nn2 and np.array_equal(idx, idx2) and f0 == g0 and np.array_equal(fl, gl), (k, r2)
    with pytest.raises(_ffi.RRTError):
        tree.query(300, 0, 10)  # outside the grid
    tree.reset()
    with pytest.raises(_ffi.RRTError):
        tree.query(1, 1, 10)  # empty tree
    tree.close()
