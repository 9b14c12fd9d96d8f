"""plan() for a planner with a caller-supplied cost function (reference rrt.py:55, :70-80: any Python callable
``costfn(vcosts, points, v, x) -> float``).

A Python callable cannot run on the device, so for such a planner the expansion loop stays on the host -- statement for
statement what rrt.py:418-437 (RRTStandard), :498-548 (RRTStar) and :690-748 (RRTStarInformed) do with the callable, including
what the default cost never exercises: the rewire block (:531-546) really fires, prices with possibly stale costs, removes
the vertex from its old parent's child list without entering it in the new one, and the ``ValueError`` of a second rewire of
the same vertex is swallowed by RRTStar (:541-546) and not by RRTStarInformed (:740).  What the loop asks of the tree and the
grid comes from the device, once per iteration (``rrt_tree_query``, include/rrt_hip.h): ``near()[0]``, ``within()`` and the
lines of sight of :424/:506, :519 and :537 (one direction, vertex -> sample, serves both loops).  There is no CPU fallback:
the provider is an ``_ffi.DeviceTree``; the tests substitute a numpy stand-in to check the loop itself on a machine without a GPU.

The same loop is the GENERAL path for problems the expansion kernels do not take (grids beyond 2048 x 2048, n beyond 262 143:
packed 24-bit keys) with the reference's DEFAULT cost: there the per-candidate Python calls are replaced by their numpy form
(`vcosts[v] + sqrt(d2)`, rrt.py:72-78), and the rewire block, which cannot fire with that cost (rrt.py:532-536), is skipped.
Slower than the kernels by orders of magnitude, same results: the class has no size at which it refuses.

Tie policy as everywhere: lowest index among equal distance / equal cost.
"""
import math
from collections import defaultdict

import numpy as np

from . import hostprep
from .hostprep import INT64_MIN


class DeviceProvider:
    """near()[0] / within() / collisionfree from the device (rrt_tree_* and rrt_prim_collisionfree of the C ABI)."""

    def __init__(self, ctx, capacity):
        from . import _ffi

        self.ctx = ctx
        self.tree = _ffi.DeviceTree(ctx, capacity)

    def reset(self):
        self.tree.reset()

    def append(self, x, y):
        return self.tree.append(x, y)

    def query(self, x, y, r2):
        return self.tree.query(x, y, r2)

    def collisionfree_many(self, ab):
        return self.ctx.prim_collisionfree(ab)[0]

    def close(self):
        self.tree.close()


def _sentinel_fault(W):
    # rrt.py:218 indexes og[INT64_MIN, ...] when handed an unfilled row (pure Python; unchecked under a real Numba install)
    return IndexError(f"index {INT64_MIN} is out of bounds for axis 0 with size {W}")


def plan_with_costfn(planner, alg, xstart, xgoal, prov, pbar=None):
    """The reference's plan() body for `planner` (alg 0 RRTStandard, 1 RRTStar, 2 RRTStarInformed) with planner.cost a custom
    callable.  Returns (vgoal, points, parents, vcosts) as handed to build_graph (rrt.py:334)."""
    n = int(planner.n)
    cost = planner.cost
    default_cost = not getattr(planner, "_custom_cost", True)  # the closure of rrt.py:72-78: evaluated with numpy, rewire skipped
    og = np.asarray(planner.og)
    W = og.shape[0]
    star, informed = alg >= 1, alg == 2
    sampled = set()
    vsoln = []
    points = np.full((n, 2), INT64_MIN, dtype=np.int64)  # np.full(dtype=int, fill_value=inf) rows (rrt.py:408)
    vcosts = np.full((n,), np.inf)
    children, parents = defaultdict(list), {}
    points[0] = xstart
    vcosts[0] = 0
    parents[0] = None
    prov.reset()
    prov.append(int(points[0, 0]), int(points[0, 1]))
    r = planner.r_rewire if star else 0
    R2 = hostprep.radius_threshold(r) if star else 0  # (d2 < r * r) <=> (d2 < R2) for integer d2
    i, j = 0, 1
    while i < n:
        if pbar is not None:
            pbar.update(1)
        if informed and len(vsoln) > 0:  # rrt.py:697-701
            vbest, cbest = planner.least_cost(vcosts, list(vsoln))
            cbest += planner_r2norm(xgoal - points[vbest])
            xnew = planner.sample_ellipse(xstart, xgoal, cbest)
            planner.ellipses[j] = planner.get_ellipse_for_plt(xstart, xgoal, cbest)
        else:
            xnew = planner.sample_all_free()
        x0, x1 = int(xnew[0]), int(xnew[1])
        vnearest, vlive, nocoll, los_live = prov.query(x0, x1, R2)  # near()[0], within() over the live rows, the lines of sight
        if nocoll and (x0, x1) not in sampled and j != n:  # rrt.py:425 / :507 / :707
            sampled.add((x0, x1))
            vbest = vnearest
            cbest = cost(vcosts, points, vbest, xnew)
            if star and default_cost:
                # rrt.py:513-521 with the default cost, all candidates at once: the walk ends at the (cost, index)-smallest visible
                # entry strictly below the cost through the nearest vertex (unfilled rows cost inf and never qualify)
                vl = np.asarray(vlive, dtype=np.int64)
                if vl.size:
                    d = points[vl] - xnew
                    cn = vcosts[vl] + np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float64))
                    ok = np.flatnonzero(np.asarray(los_live, dtype=bool) & (cn < cbest))
                    if ok.size:
                        k = ok[np.argmin(cn[ok])]  # first minimum: lowest index among equal costs
                        vbest, cbest = int(vl[k]), cn[k]
            elif star:
                # within() runs over all n rows (rrt.py:176-181): an unfilled row holds INT64_MIN, its squared distance wraps to
                # |xnew|^2, so every unfilled row (row j included) is "within" when the sample lies that close to the origin
                vnear = [int(v) for v in vlive]
                free_of = dict(zip(vnear, (bool(f) for f in los_live)))
                if x0 * x0 + x1 * x1 < r * r:
                    vnear.extend(range(j, n))

                def sees(vn):  # collisionfree(og, points[vn], xnew), rrt.py:519 / :537
                    if vn in free_of:
                        return free_of[vn]
                    if points[vn, 0] == INT64_MIN:
                        raise _sentinel_fault(W)
                    return og[x0, x1] == 0  # row j, filled with xnew by now: the walk is the one cell

                for vn in vnear:  # choose parent, rrt.py:515-521
                    cn = cost(vcosts, points, vn, xnew)
                    if cn < cbest:
                        if sees(vn):
                            vbest = vn
                            cbest = cn
            vnew = j
            points[vnew] = xnew
            vcosts[vnew] = cbest
            parents[vnew] = vbest
            children[vbest].append(vnew)
            if star and not default_cost:
                for vn in vnear:  # rewire, rrt.py:531-546 / :731-742 (never true with the default cost: rrt.py:532-536)
                    cn = vcosts[vn]
                    cmaybe = cost(vcosts, points, vn, xnew)
                    if cmaybe < cn:
                        if sees(vn):
                            parent = parents[vn]
                            if parent is not None:
                                if informed:  # rrt.py:740: a second rewire of one vertex raises ValueError out of plan()
                                    children[parent].remove(vn)
                                    parents[vn] = vnew
                                    vcosts[vn] = cmaybe
                                else:
                                    try:
                                        children[parent].remove(vn)
                                        parents[vn] = vnew
                                        vcosts[vn] = cmaybe
                                    except ValueError:
                                        pass
            if informed and planner_r2norm(xnew - xgoal) < planner.r_goal:  # rrt.py:744-745
                vsoln.append(vnew)
            prov.append(x0, x1)
            j += 1
        i += 1
    # go2goal, rrt.py:311-332: the callable prices every row, then the rows are tried in stable (cost, index) order
    costs = np.empty(vcosts.shape)
    if default_cost:
        with np.errstate(over="ignore"):
            d = points - np.asarray(xgoal)
            costs[:] = vcosts + np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float64))  # (unfilled rows: inf + anything)
    else:
        for k in range(points.shape[0]):
            costs[k] = cost(vcosts, points, k, xgoal)
    order = np.argsort(costs, kind="stable")
    xg0, xg1 = int(xgoal[0]), int(xgoal[1])
    vgoal = None
    for b0 in range(0, order.size, 64):  # 64 lines of sight per device call, consumed in order
        chunk = order[b0:b0 + 64]
        live = chunk[points[chunk, 0] != INT64_MIN]
        free = prov.collisionfree_many(np.column_stack([points[live], np.full(live.size, xg0), np.full(live.size, xg1)])) if live.size else []
        free_of = dict(zip(live.tolist(), (bool(f) for f in free)))
        for idx in chunk.tolist():
            if idx not in free_of:
                raise _sentinel_fault(W)
            if free_of[idx]:
                vgoal = j
                points = np.concatenate((points, np.asarray(xgoal)[np.newaxis, :]), axis=0)
                vcosts = np.concatenate((vcosts, [costs[idx]]), axis=0)
                points[vgoal] = xgoal
                vcosts[vgoal] = costs[idx]
                children[idx].append(vgoal)
                parents[vgoal] = idx
                break
        if vgoal is not None:
            break
    if vgoal is None:
        vgoal = np.argmin(np.linalg.norm(points - xgoal))  # rrt.py:328-331 (no axis: one number, vertex 0)
    return vgoal, points, parents, vcosts


def planner_r2norm(v):
    """r2norm of the reference (rrt.py:24) on an integer difference"""
    return math.sqrt(v[0] * v[0] + v[1] * v[1])
