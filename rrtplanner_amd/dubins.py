"""Dubins-vehicle RRT / RRT* planners on the MI355X (BASELINE.json configs[4]).

The reference advertises a "Dubins Vehicle RRT Planner" and a "Dubins Vehicle RRT(star) Planner"
(/root/reference/README.md:12,18-19) but ships no such module, so there is NO REFERENCE PARITY
here: the semantics are this build's own (include/rrt_dubins.h).  The classes
follow the reference's planner surface -- ``RRTDubins(og, n, rho, ...)`` /
``RRTStarDubins(og, n, r_rewire, rho, ...)``, ``plan(xstart, xgoal) -> (nx.DiGraph, goal_vertex)``,
``route2gv`` -- with poses ``(x, y, h)``: an integer grid cell and one of ``n_headings`` discrete
headings (``theta = 2 pi h / n_headings``).  A tree edge is the shortest Dubins word of turning
radius ``rho`` (cells) from the parent's pose to the child's; its ``dist`` is the arc length.

Nearest vertex, near set, accept test and the choose-parent walk are the reference's
(rrt.py:418-437, :498-548) on the (x, y) part of the poses; the expansion loop runs on the device
(no CPU fallback).  The numpy geometry in this file (``dubins_shortest``, ``dubins_polyline``)
serves the host side only: edge lengths of the returned graph and polylines for plotting / path
following.  It uses libm, so it agrees with the device's fixed-order arithmetic to ~1e-12, not bit
for bit (tests/test_dubins.py compares the two).
"""
import math
from typing import Tuple

import networkx as nx
import numpy as np
from tqdm import tqdm

from . import _ffi, hostprep
from .rrt import RRT, TreeDiGraph

__all__ = ["RRTDubins", "RRTStarDubins", "dubins_shortest", "dubins_polyline", "WORDS"]

WORDS = ("LSL", "LSR", "RSL", "RSR", "RLR", "LRL")
_KINDS = {"L": 1, "S": 0, "R": -1}
_TWOPI = 2.0 * math.pi


def _mod2pi(a):
    return a - _TWOPI * np.floor(a / _TWOPI)


def dubins_shortest(x0, y0, th0, x1, y1, th1, rho):
    """Shortest Dubins word between poses (vectorised over numpy arrays).  Returns (t, p, q, length, word index into
    WORDS); t, p, q in units of rho.  Ties between words go to the first in WORDS order."""
    x0, y0, th0, x1, y1, th1 = np.broadcast_arrays(*[np.asarray(v, dtype=np.float64) for v in (x0, y0, th0, x1, y1, th1)])
    dx, dy = x1 - x0, y1 - y0
    d = np.hypot(dx, dy) / rho
    theta = _mod2pi(np.arctan2(dy, dx))
    al, be = _mod2pi(th0 - theta), _mod2pi(th1 - theta)
    sa, ca, sb, cb, cab = np.sin(al), np.cos(al), np.sin(be), np.cos(be), np.cos(al - be)
    dsq = d * d
    cands = []
    with np.errstate(invalid="ignore"):
        psq = 2 + dsq - 2 * cab + 2 * d * (sa - sb)
        tmp = np.arctan2(cb - ca, d + sa - sb)
        cands.append((psq >= 0, _mod2pi(tmp - al), np.sqrt(psq), _mod2pi(be - tmp)))
        psq = -2 + dsq + 2 * cab + 2 * d * (sa + sb)
        p = np.sqrt(psq)
        tmp = np.arctan2(-ca - cb, d + sa + sb) - np.arctan2(-2.0, p)
        cands.append((psq >= 0, _mod2pi(tmp - al), p, _mod2pi(tmp - _mod2pi(be))))
        psq = -2 + dsq + 2 * cab - 2 * d * (sa + sb)
        p = np.sqrt(psq)
        tmp = np.arctan2(ca + cb, d - sa - sb) - np.arctan2(2.0, p)
        cands.append((psq >= 0, _mod2pi(al - tmp), p, _mod2pi(be - tmp)))
        psq = 2 + dsq - 2 * cab + 2 * d * (sb - sa)
        tmp = np.arctan2(ca - cb, d - sa + sb)
        cands.append((psq >= 0, _mod2pi(al - tmp), np.sqrt(psq), _mod2pi(tmp - be)))
        tmp = (6 - dsq + 2 * cab + 2 * d * (sa - sb)) / 8
        p = _mod2pi(_TWOPI - np.arccos(tmp))
        t = _mod2pi(al - np.arctan2(ca - cb, d - sa + sb) + _mod2pi(p / 2))
        cands.append((np.abs(tmp) <= 1, t, p, _mod2pi(al - be - t + _mod2pi(p))))
        tmp = (6 - dsq + 2 * cab + 2 * d * (sb - sa)) / 8
        p = _mod2pi(_TWOPI - np.arccos(tmp))
        t = _mod2pi(-al - np.arctan2(ca - cb, d + sa - sb) + p / 2)
        cands.append((np.abs(tmp) <= 1, t, p, _mod2pi(_mod2pi(be) - al - t + _mod2pi(p))))
    best = np.full(d.shape, np.inf)
    bt, bp, bq = np.zeros(d.shape), np.zeros(d.shape), np.zeros(d.shape)
    bw = np.full(d.shape, len(WORDS), dtype=np.int64)
    for w, (ok, t, p, q) in enumerate(cands):
        s = np.where(ok, t + p + q, np.inf)
        take = s < best
        best, bt, bp, bq, bw = np.where(take, s, best), np.where(take, t, bt), np.where(take, p, bp), np.where(take, q, bq), np.where(take, w, bw)
    return bt, bp, bq, best * rho, bw


def dubins_polyline(pose0, pose1, rho, n_headings, ds=0.5):
    """(M, 2) float array of points every `ds` cells of arc length along the shortest Dubins word from pose0 to pose1
    (poses are (x, y, heading index)), end point included."""
    th0, th1 = _TWOPI * pose0[2] / n_headings, _TWOPI * pose1[2] / n_headings
    t, p, q, length, w = (float(v) for v in dubins_shortest(pose0[0], pose0[1], th0, pose1[0], pose1[1], th1, rho))
    kinds = [_KINDS[c] for c in WORDS[int(w)]]
    s = np.append(np.arange(0.0, length, ds), length) / rho
    out = np.empty((s.size, 2))
    x, y, th, start = 0.0, 0.0, th0, 0.0
    for kind, tau_len in zip(kinds, (t, p, q)):
        m = (s >= start) & (s <= start + tau_len + 1e-12)
        tau = s[m] - start
        if kind == 0:
            out[m, 0], out[m, 1] = x + np.cos(th) * tau, y + np.sin(th) * tau
            x, y = x + math.cos(th) * tau_len, y + math.sin(th) * tau_len
        else:
            out[m, 0] = x + kind * (np.sin(th + kind * tau) - math.sin(th))
            out[m, 1] = y - kind * (np.cos(th + kind * tau) - math.cos(th))
            x, y = x + kind * (math.sin(th + kind * tau_len) - math.sin(th)), y - kind * (math.cos(th + kind * tau_len) - math.cos(th))
            th = th + kind * tau_len
        start += tau_len
    return np.asarray(pose0[:2], dtype=np.float64) + out * rho


class _DubinsBase(RRT):
    _STAR = False

    def __init__(self, og: np.ndarray, n: int, rho: float, n_headings: int = 64, costfn: callable = None, pbar: bool = True, seed: int = 0):
        super().__init__(og, n, costfn=costfn, pbar=pbar, seed=seed)
        if not (rho > 0) or not (1 <= int(n_headings) <= 256):
            raise ValueError("rho must be positive and 1 <= n_headings <= 256")
        self.rho = float(rho)
        self.n_headings = int(n_headings)

    def sample_all_free(self):
        """One sample pose: a uniform free cell (rrt.py:231-240) and a uniform heading index."""
        return np.append(super().sample_all_free(), self.rand_gen.integers(0, self.n_headings))

    def _pose(self, x, name):
        p = np.asarray(x)
        if p.shape != (3,):
            raise ValueError(f"{name} must be a pose (x, y, heading index)")
        xy = hostprep.as_int_point(p[:2], name)
        h = int(p[2])
        if h != p[2] or not 0 <= h < self.n_headings:
            raise ValueError(f"{name}: heading index must be an integer in [0, {self.n_headings})")
        W, H = np.asarray(self.og).shape
        if not (0 <= xy[0] < W and 0 <= xy[1] < H):
            raise ValueError(f"{name}={xy.tolist()} lies outside the {W}x{H} occupancy grid")
        return np.array([xy[0], xy[1], h], dtype=np.int64)

    def _run_dubins(self, xstart, xgoal, r_rewire=None, logs=False):
        if self._custom_cost:
            raise NotImplementedError("the Dubins planners run with the arc-length cost on the device; a Python costfn cannot be lowered")
        xs, xg = self._pose(xstart, "xstart"), self._pose(xgoal, "xgoal")
        n = int(self.n)
        ctx = self._device()
        # the sample stream: n free cells (like rrt.py:240, one block), then n heading indices
        samples = hostprep.draw_free_samples(self.rand_gen, self.free, n)
        headings = self.rand_gen.integers(0, self.n_headings, size=n)
        alg = _ffi.ALG_DUBINS_STAR if self._STAR else _ffi.ALG_DUBINS
        query, keep = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=hostprep.radius_threshold(r_rewire) if r_rewire is not None else 0,
                                      headings=headings, rho=self.rho, nh=self.n_headings)
        rc, res = ctx.plan(query, n, logs=logs)
        if rc == _ffi.RRT_E_GOAL_UNREACHABLE:
            raise IndexError(f"index {hostprep.INT64_MIN} is out of bounds for axis 0 with size {np.asarray(self.og).shape[0]}")  # as rrt.py:317-318
        res.xs, res.xg = xs, xg
        self.last_stats = {k: getattr(res, k) for k in ("j", "sum_j", "sum_cells_nn", "sum_near", "sum_cells_cand", "n_los_cand")}
        return res

    def _materialise_dubins(self, res) -> Tuple[nx.DiGraph, int]:
        """The reference's graph contract (rrt.py:334-369) with poses: node attributes `pt` (int64 (2,)) and `heading` (int),
        edge attributes `dist` (arc length of the Dubins word, float) and `cost` (np.float64)."""
        n, j, found = res.n, res.j, bool(res.found)
        rows = n + 1 if found else n
        live = j + 1 if found else j
        points = np.full((rows, 2), hostprep.INT64_MIN, dtype=np.int64)
        vcosts = np.full((rows,), np.inf)
        heads = np.zeros(rows, dtype=np.int64)
        points[:live], vcosts[:live], heads[:live] = res.pts[:live], res.vcost[:live], res.head[:live]
        vgoal = int(res.vgoal)
        if found:
            points[n], vcosts[n], heads[n] = res.xg[:2], vcosts[vgoal], res.xg[2]
        T = DubinsTree.from_arrays(vgoal, points, np.array(res.parent[:live], dtype=np.int64), vcosts)
        T.__dict__["_dub"] = (heads, self.rho, self.n_headings)
        return T, vgoal

    def _plan_dubins(self, xstart, xgoal, **kw):
        bar = tqdm(total=self.n) if self.pbar else None
        try:
            out = self._materialise_dubins(self._run_dubins(xstart, xgoal, **kw))
            if bar is not None:
                bar.update(self.n)
        finally:
            if bar is not None:
                bar.close()
        return out

    def path_points(self, T: nx.DiGraph, path: list, ds: float = 0.5) -> np.ndarray:
        """(M, 2) float polyline of the vehicle's path along the vertices of `path` (from route2gv)."""
        legs = []
        for u, v in zip(path[:-1], path[1:]):
            a, b = T.nodes[u], T.nodes[v]
            legs.append(dubins_polyline((*a["pt"], a["heading"]), (*b["pt"], b["heading"]), self.rho, self.n_headings, ds))
        return np.concatenate(legs) if legs else np.zeros((0, 2))


class DubinsTree(TreeDiGraph):
    """TreeDiGraph whose nodes also carry `heading` and whose edge `dist` is the Dubins arc length."""

    def _materialise(self):
        dub = self.__dict__.get("_dub")
        lazy = self.__dict__.get("_lazy")
        super()._materialise()
        if lazy is None or dub is None:
            return
        heads, rho, nh = dub
        _, points, parent, _ = lazy
        node, succ = self.__dict__["_td_node"], self.__dict__["_td_adj"]
        for v, a in node.items():
            a["heading"] = int(heads[v])
        ch = np.arange(1, len(parent))
        if ch.size:
            pa = parent[1:]
            L = dubins_shortest(points[pa, 0], points[pa, 1], _TWOPI * heads[pa] / nh, points[ch, 0], points[ch, 1], _TWOPI * heads[ch] / nh, rho)[3]
            for p, c, d in zip(pa.tolist(), ch.tolist(), L.tolist()):
                succ[p][c]["dist"] = d


class RRTDubins(_DubinsBase):
    """Dubins-vehicle RRT: the parent of a new pose is the nearest vertex (by cell distance), connected by the shortest
    Dubins word (the reference's RRTStandard loop, rrt.py:418-437, with Dubins edges)."""

    def plan(self, xstart: np.ndarray, xgoal: np.ndarray) -> Tuple[nx.DiGraph, int]:
        return self._plan_dubins(xstart, xgoal)


class RRTStarDubins(_DubinsBase):
    """Dubins-vehicle RRT*: choose-parent over the vertices within r_rewire cells (the reference's RRTStar loop,
    rrt.py:498-548, with Dubins edges; like there the rewire scan cannot fire)."""

    _STAR = True

    def __init__(self, og: np.ndarray, n: int, r_rewire: float, rho: float, n_headings: int = 64, costfn: callable = None, pbar: bool = True,
                 seed: int = 0):
        super().__init__(og, n, rho, n_headings=n_headings, costfn=costfn, pbar=pbar, seed=seed)
        self.r_rewire = r_rewire

    def plan(self, xstart: np.ndarray, xgoal: np.ndarray) -> Tuple[nx.DiGraph, int]:
        return self._plan_dubins(xstart, xgoal, r_rewire=self.r_rewire)
