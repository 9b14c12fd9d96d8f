"""Planner classes with the reference's surface, running the expansion loop on an MI355X.

Mirrors rrtplanner/rrt.py of the reference: ``RRT`` (:50), ``RRTStandard`` (:375), ``RRTStar``
(:453), ``RRTStarInformed`` (:562), ``r2norm`` (:11), ``random_point_og`` (:27).  ``plan()`` keeps
the reference's contract -- ``(networkx.DiGraph, goal_vertex)`` with node attribute ``pt`` and edge
attributes ``dist`` / ``cost`` (rrt.py:334-369) -- but its body is one call into the HIP engine
(include/rrt_hip.h) instead of the Python loop.  The numpy ``Generator`` stays the source of
randomness on the host, so the sample stream is the reference's bit for bit.

A caller-supplied ``costfn`` keeps the loop on the host (``hostloop.py``) with the device answering its per-iteration questions.
There is no CPU fallback for ``plan()``: without the HIP library or a GPU it raises.
Tie policy: nearest node / goal connection pick the lowest index among
equal distance / cost (== a stable argsort in rrt.py:154 and :317).
"""
import itertools
import math
from collections.abc import Mapping
from typing import List, Tuple

import networkx as nx
import numpy as np
from tqdm import tqdm

from . import _ffi, hostprep
from .hostprep import INT64_MIN

__all__ = ["r2norm", "random_point_og", "RRT", "RRTStandard", "RRTStar", "RRTStarInformed", "TreeDiGraph"]


def _networkx_layout_ok() -> bool:
    """The fast graph construction below writes networkx's private dictionaries (`_node`, `_adj` / `_succ`, `_pred`).  Check on
    a scratch graph that they are laid out the way this file assumes (true for networkx 2.x and 3.x as tested; the reference pins
    2.6.3): plain dicts, `_succ` the same object as `_adj`, add_node / add_edge filling them as dict-of-dict with ONE shared
    attribute dict per edge.  Anything else (a future networkx) switches to the public add_node / add_edge calls."""
    try:
        g = nx.DiGraph()
        g.add_node(7, pt=1)
        g.add_edge(7, 8, dist=1.0, cost=2.0)
        return (type(g._node) is dict and type(g._adj) is dict and type(g._pred) is dict and g._succ is g._adj and g._node[7] == {"pt": 1}
                and g._adj[7][8] is g._pred[8][7] and g._adj[7][8] == {"dist": 1.0, "cost": 2.0} and g._adj[8] == {} and g._pred[7] == {})
    except Exception:
        return False


_NX_FAST = _networkx_layout_ok()


def _fill_graph_public(T, vgoal, points, parents, vcosts):
    """build_graph of the reference call by call (rrt.py:357-369), through networkx's public interface only."""
    T.add_node(vgoal, pt=points[vgoal])
    for i, p in enumerate(points):
        T.add_node(i, pt=p)
    for child, parent in parents.items():
        if parent is not None:
            T.add_edge(parent, child, dist=r2norm(points[child] - points[parent]), cost=vcosts[child])


def _fill_graph(T, vgoal, points, parents, vcosts, node_dicts=None, edge_dicts=None):
    """Fill the node / adjacency dictionaries of DiGraph `T` like build_graph of the reference (rrt.py:357-369): node order
    [vgoal, 0, 1, ...], one edge per tree link in the order of `parents`, attributes `pt` (int64 row view), `dist` (float),
    `cost` (np.float64).  The dictionaries are written directly (same dict-of-dict layout `add_node` / `add_edge` produce, one
    shared attribute dict per edge in `_succ` and `_pred`), ~3x faster than 100 000 add_node / add_edge calls;
    tests/test_host_logic.py compares it with the call-by-call construction.
    node_dicts / edge_dicts: attribute dictionaries the array-backed views of a TreeDiGraph have already handed out (by vertex /
    by child); they become the graph's own, so that whatever a caller wrote into them stays."""
    if not _NX_FAST:
        return _fill_graph_public(T, vgoal, points, parents, vcosts)
    rows = len(points)
    order = [vgoal] + [i for i in range(rows) if i != vgoal] if 0 <= vgoal < rows else [vgoal] + list(range(rows))
    node, succ, pred = T._node, T._succ, T._pred
    if node_dicts is None:
        pts = list(points)  # row views, like `for i, p in enumerate(points)`
        for v in order:
            node[v] = {"pt": pts[v]}
            succ[v] = {}
            pred[v] = {}
    else:
        for v in order:
            node[v] = node_dicts[v]
            succ[v] = {}
            pred[v] = {}
    kids = [c for c, p in parents.items() if p is not None]
    if kids:
        ch = np.asarray(kids, dtype=np.int64)
        pa = np.asarray([parents[c] for c in kids], dtype=np.int64)
        if edge_dicts is None:
            for p, c, e in zip(pa.tolist(), kids, _edge_attr_dicts(points, pa, ch, vcosts)):
                succ[p][c] = e
                pred[c][p] = e
        else:
            for p, c in zip(pa.tolist(), kids):
                e = edge_dicts[c]
                succ[p][c] = e
                pred[c][p] = e


def _edge_attr_dicts(points, pa, ch, vcosts):
    """{"dist": float, "cost": np.float64} of the edges pa[k] -> ch[k] (rrt.py:366-369)"""
    d = points[ch] - points[pa]
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float64)).tolist()
    return [{"dist": dd, "cost": cc} for dd, cc in zip(dist, vcosts[ch])]


class _ArrayNodeView(Mapping):
    """`T.nodes` of a TreeDiGraph that still only holds arrays: order [vgoal, 0, 1, ...], `T.nodes[v]` -> {"pt": row v}
    (rrt.py:357-361).  The attribute dictionaries are made once and become the graph's own when it is materialised; whatever this
    class does not serve itself (set algebra, data="pt", ...) goes to networkx's NodeView of the materialised graph, and so does
    every call once the graph has been materialised (a view somebody kept)."""

    __slots__ = ("_g", "_nd", "_rows", "_vgoal")

    def __init__(self, g):
        self._g = g
        self._nd = None  # attribute dictionaries by vertex, made on first use
        self._vgoal, points = g.__dict__["_lazy"][0], g.__dict__["_lazy"][1]
        self._rows = len(points)

    def _real(self):
        return nx.reportviews.NodeView(self._g)  # (reads g._node: materialises)

    def _stale(self):
        return self._g.__dict__["_lazy"] is None

    def __len__(self):
        return len(self._real()) if self._stale() else self._rows

    def __iter__(self):
        if self._stale():
            return iter(self._real())
        vgoal, rows = self._vgoal, self._rows
        return itertools.chain((vgoal,), range(vgoal), range(vgoal + 1, rows))

    def __contains__(self, v):
        if self._stale():
            return v in self._real()
        try:
            return 0 <= v < self._rows and int(v) == v
        except (TypeError, ValueError):
            return False

    def __getitem__(self, v):
        nd = self._nd
        if nd is None or self._g.__dict__["_lazy"] is None or v.__class__ is not int or v < 0:
            return self._getitem_slow(v)
        try:
            return nd[v]
        except IndexError:
            raise KeyError(v) from None

    def _getitem_slow(self, v):
        if self._stale() or isinstance(v, slice):
            return self._real()[v]  # (networkx raises its own error for slices)
        if v not in self:
            raise KeyError(v)
        if self._nd is None:
            self._nd = self._g._arr_node_dicts()
        return self._nd[int(v)]

    def __call__(self, data=False, default=None):
        if self._stale() or (data is not False and data is not True):
            return self._real()(data=data, default=default)
        return _ArrayNodeDataView(self) if data else self

    def data(self, data=True, default=None):
        return self(data=data, default=default)

    def __getattr__(self, name):  # anything else NodeView offers
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self._real(), name)

    def __repr__(self):
        return f"NodeView({tuple(self)})"


class _ArrayNodeDataView:
    __slots__ = ("_v",)

    def __init__(self, v):
        self._v = v

    def __len__(self):
        return len(self._v)

    def __iter__(self):
        v = self._v
        if v._stale():
            return iter(v._real()(data=True))
        nd = v._g._arr_node_dicts()
        return zip(v, map(nd.__getitem__, v))

    def __contains__(self, item):
        try:
            v, d = item
            return v in self._v and self._v[v] == d
        except (TypeError, ValueError):
            return False

    def __getitem__(self, v):
        return self._v[v]


class _ArrayEdgeView:
    """`T.edges` of a TreeDiGraph that still only holds arrays: the edges parent -> child in networkx's order (by source vertex in
    node order, then by insertion = child index), `T.edges[u, v]` -> {"dist", "cost"} (rrt.py:363-369), `T.edges(data=True)`.
    An edge (u, v) exists iff u is v's parent, so lookups need no table.  Everything else goes to networkx's OutEdgeView of the
    materialised graph, and so does every call once the graph has been materialised."""

    __slots__ = ("_g", "_par", "_ed")

    def __init__(self, g):
        self._g = g
        self._par = None  # parent of every live vertex as a list, the edges' attribute dictionaries by child: made on first use
        self._ed = None

    def _real(self):
        return nx.reportviews.OutEdgeView(self._g)

    def _stale(self):
        return self._g.__dict__["_lazy"] is None

    def __len__(self):
        return len(self._real()) if self._stale() else max(self._g._arr_live() - 1, 0)

    def __iter__(self):
        if self._stale():
            return iter(self._real())
        us, vs = self._g._arr_edge_order()
        return zip(us, vs)

    def _child_of(self, e):
        try:
            u, v = e
            if self._par is None:
                self._par = self._g.__dict__["_lazy"][2].tolist()
            if 1 <= v < len(self._par) and int(v) == v and self._par[int(v)] == u:
                return int(v)
        except (TypeError, ValueError):
            pass
        return None

    def __contains__(self, e):
        return (e in self._real()) if self._stale() else self._child_of(e) is not None

    def __getitem__(self, e):
        if self._g.__dict__["_lazy"] is None or isinstance(e, slice):
            return self._real()[e]
        par, ed = self._par, self._ed
        if ed is not None and e.__class__ is tuple and len(e) == 2:  # fast path: plain ints, as the views' own iteration hands out
            v = e[1]
            if v.__class__ is int and 0 < v < len(par) and par[v] == e[0]:
                return ed[v]
        c = self._child_of(e)
        if c is None:
            raise KeyError(e)
        if self._ed is None:
            self._ed = self._g._arr_edge_dicts()
        return self._ed[c]

    def __call__(self, nbunch=None, data=False, default=None):
        if self._stale() or nbunch is not None or (data is not False and data is not True):
            return self._real()(nbunch=nbunch, data=data, default=default)
        return _ArrayEdgeDataView(self) if data else self

    def data(self, data=True, default=None, nbunch=None):
        return self(nbunch=nbunch, data=data, default=default)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self._real(), name)

    def __repr__(self):
        return f"OutEdgeView({list(self)})"


class _ArrayEdgeDataView:
    __slots__ = ("_v",)

    def __init__(self, v):
        self._v = v

    def __len__(self):
        return len(self._v)

    def __iter__(self):
        v = self._v
        if v._stale():
            return iter(v._real()(data=True))
        us, vs = v._g._arr_edge_order()
        ed = v._g._arr_edge_dicts()
        return zip(us, vs, map(ed.__getitem__, vs))

    def __contains__(self, item):
        try:
            u, v, d = item
            return (u, v) in self._v and self._v[u, v] == d
        except (TypeError, ValueError):
            return False


class TreeDiGraph(nx.DiGraph):
    """The nx.DiGraph that plan() returns, with its dictionaries filled on first use.

    At n = 50 000 building 100 000 Python dictionaries takes several times as long as the whole expansion on the device, and
    the usual next calls only need the parent pointers and the coordinates.  The instance therefore keeps the result arrays:
    route2gv / vertices_as_ndarray read them directly, and `T.nodes` / `T.edges` -- what the reference's consumers use
    (plots.py:27-29, :40-41; anim.py:101-102) -- are array-backed views (`for u, v in T.edges`, `T.edges(data=True)`, `T.nodes[v]["pt"]`,
    `T.edges[u, v]["cost"]`) that make each attribute dictionary once and never build the adjacency structure.  Anything else
    (adjacency, mutation, copy, algorithms) first materialises `_node` / `_adj` / `_succ` / `_pred` with exactly the content
    build_graph gives (tests compare both), reusing the dictionaries already handed out; from then on it is an ordinary DiGraph."""

    def __init__(self, incoming_graph_data=None, **attr):
        self.__dict__["_lazy"] = None
        super().__init__(incoming_graph_data, **attr)

    @classmethod
    def from_arrays(cls, vgoal, points, parent, vcosts):
        """points (rows, 2) int64, parent (live,) int (-1 for the root), vcosts (rows,) float64; rows >= live"""
        if not _NX_FAST:  # unknown networkx layout: an ordinary, eagerly built DiGraph (same content, see _networkx_layout_ok)
            T = nx.DiGraph()
            parents = {0: None}
            for child, p in enumerate(np.asarray(parent).tolist()):
                if child > 0:
                    parents[child] = p
            _fill_graph_public(T, int(vgoal), points, parents, vcosts)
            return T
        T = cls()
        T.__dict__["_lazy"] = (int(vgoal), points, parent, vcosts)
        return T

    def _materialise(self):
        lazy = self.__dict__.get("_lazy")
        if lazy is not None:
            self.__dict__["_lazy"] = None
            vgoal, points, parent, vcosts = lazy
            parents = {0: None}
            par = parent.tolist()
            for child in range(1, len(par)):
                parents[child] = par[child]
            _fill_graph(self, vgoal, points, parents, vcosts, self.__dict__.pop("_arr_nd", None), self.__dict__.pop("_arr_ed", None))
            for k in ("_arr_eo", "_arr_nv", "_arr_ev"):
                self.__dict__.pop(k, None)

    def __getstate__(self):  # (the cached views point back at the graph and are remade on demand)
        return {k: v for k, v in self.__dict__.items() if k not in ("_arr_nv", "_arr_ev")}

    def __setstate__(self, state):
        self.__dict__.update(state)

    # ---- array-backed views (only while the graph is lazy, and only for the plain tree: subclasses add attributes) ----
    def _arr_views(self):
        lazy = self.__dict__.get("_lazy")
        return type(self) is TreeDiGraph and lazy is not None and 0 <= lazy[0] < len(lazy[1])

    def _arr_vgoal(self):
        return self.__dict__["_lazy"][0]

    def _arr_rows(self):
        return len(self.__dict__["_lazy"][1])

    def _arr_live(self):
        return len(self.__dict__["_lazy"][2])

    def _arr_node_dicts(self):
        nd = self.__dict__.get("_arr_nd")
        if nd is None:
            nd = self.__dict__["_arr_nd"] = [{"pt": p} for p in self.__dict__["_lazy"][1]]  # row views, like build_graph's
        return nd

    def _arr_edge_dicts(self):
        """by child vertex (entry 0: the root has no edge)"""
        ed = self.__dict__.get("_arr_ed")
        if ed is None:
            _, points, parent, vcosts = self.__dict__["_lazy"]
            ch = np.arange(1, len(parent), dtype=np.int64)
            ed = self.__dict__["_arr_ed"] = [None] + _edge_attr_dicts(points, np.asarray(parent[1:], dtype=np.int64), ch, vcosts)
        return ed

    def _arr_edge_order(self):
        """(sources, targets) of all edges in networkx's iteration order: sources in node order [vgoal, 0, 1, ...], the out-edges
        of one source in insertion order (ascending child)"""
        eo = self.__dict__.get("_arr_eo")
        if eo is None:
            vgoal, _, parent, _ = self.__dict__["_lazy"]
            pa = np.asarray(parent[1:], dtype=np.int64)
            rank = np.where(pa == vgoal, 0, np.where(pa < vgoal, pa + 1, pa))
            o = np.argsort(rank * len(parent) + np.arange(len(pa)))  # (unique keys: any sort is the stable sort by rank)
            eo = self.__dict__["_arr_eo"] = (pa[o].tolist(), (o + 1).tolist())
        return eo

    @property
    def nodes(self):
        d = self.__dict__
        if d.get("_lazy") is None:
            return nx.reportviews.NodeView(self)
        v = d.get("_arr_nv")
        if v is None:
            if not self._arr_views():
                return nx.reportviews.NodeView(self)
            v = d["_arr_nv"] = _ArrayNodeView(self)
        return v

    @property
    def edges(self):
        d = self.__dict__
        if d.get("_lazy") is None:
            return nx.reportviews.OutEdgeView(self)
        v = d.get("_arr_ev")
        if v is None:
            if not self._arr_views():
                return nx.reportviews.OutEdgeView(self)
            v = d["_arr_ev"] = _ArrayEdgeView(self)
        return v

    out_edges = edges

    def root_path(self, gv):
        """[0, ..., gv] from the parent array while the graph is still lazy, else None"""
        lazy = self.__dict__.get("_lazy")
        if lazy is None:
            return None
        parent = lazy[2]
        gv = int(gv)
        if not 0 <= gv < len(parent):
            return None
        path, v = [gv], gv
        while v != 0:
            v = int(parent[v])
            if v < 0 or len(path) > len(parent):
                return None
            path.append(v)
        path.reverse()
        return path

    def lazy_points(self):
        lazy = self.__dict__.get("_lazy")
        return None if lazy is None else lazy[1]


def _lazy_dict(name, aliases=()):
    """Data descriptor for one of DiGraph's dictionaries: reading it materialises a lazy tree first.  (networkx keeps `_succ`
    as an alias of `_adj`, set together; both names are served from the same slot.)"""
    keys = ("_td" + name,) + tuple("_td" + a for a in aliases)

    def get(self):
        self._materialise()
        return self.__dict__[keys[0]]

    def set_(self, value):
        for k in keys:
            self.__dict__[k] = value

    return property(get, set_)


TreeDiGraph._node = _lazy_dict("_node")
TreeDiGraph._pred = _lazy_dict("_pred")
TreeDiGraph._adj = _lazy_dict("_adj", aliases=("_succ",))
TreeDiGraph._succ = _lazy_dict("_succ", aliases=("_adj",))


def r2norm(x) -> float:
    """2-norm of a (2,) vector (reference rrt.py:10-24)."""
    return math.sqrt(x[0] * x[0] + x[1] * x[1])


def random_point_og(og: np.ndarray, rnd_gen: np.random.Generator = None) -> np.ndarray:
    """A uniformly random free cell of the occupancy grid (reference rrt.py:27-44)."""
    free = np.argwhere(og == 0)
    draw = np.random.randint if rnd_gen is None else rnd_gen.integers
    return free[draw(low=0, high=free.shape[0])]


class RRT(object):
    """Base class: holds the grid, the sample budget and the RNG (reference rrt.py:50-86)."""

    _ALG = None

    def __init__(self, og: np.ndarray, n: int, costfn: callable = None, pbar: bool = True, seed: int = 0):
        self.pbar = pbar
        self.n = n
        self.free = np.argwhere(og == 0)
        self.og = og
        self._custom_cost = costfn is not None
        if costfn is None:

            def costfn(vcosts: np.ndarray, points: np.ndarray, v: int, x: np.ndarray) -> float:
                return vcosts[v] + r2norm(points[v] - x)

        self.cost = costfn
        self.not_a_point = [np.inf, np.inf]
        self.not_a_dist = np.inf
        self.rand_gen = np.random.default_rng(seed)
        # device side, created on first plan()
        self._ctx = None
        self._grid_dirty = True
        self.device_id = 0
        self.last_stats = None

    # ------------------------------------------------------------------ graph helpers
    def route2gv(self, T: nx.DiGraph, gv) -> List[int]:
        """Vertices of the shortest path root -> gv (reference rrt.py:87-107: Dijkstra on `dist`).

        In the tree `plan()` returns every vertex has one parent, so the shortest path is the unique root path and is read
        off the parent pointers; any other graph goes through networkx like the reference."""
        if isinstance(T, TreeDiGraph):
            fast = T.root_path(gv)
            if fast is not None:
                return fast
        path, v, pred = [gv], gv, T.pred
        while v != 0:
            ps = pred[v] if v in pred else ()
            if len(ps) != 1 or len(path) > len(pred):
                return nx.shortest_path(T, source=0, target=gv, weight="dist")  # not our tree (or unreachable: raises like the reference)
            (v,) = ps
            path.append(v)
        path.reverse()
        return path

    def vertices_as_ndarray(self, T: nx.DiGraph, path: list) -> np.ndarray:
        """(M-1, 2, 2) array of consecutive path segment endpoints (reference rrt.py:109-129)."""
        lazy_pts = T.lazy_points() if isinstance(T, TreeDiGraph) else None
        pts = [lazy_pts[v] for v in path] if lazy_pts is not None else [T.nodes[v]["pt"] for v in path]
        return np.array([[pts[k], pts[k + 1]] for k in range(len(path) - 1)])

    # ------------------------------------------------------------------ static primitives
    @staticmethod
    def near(points: np.ndarray, x: np.ndarray) -> np.ndarray:
        """Indices of `points` by ascending distance to `x` (reference rrt.py:131-155);
        equal distances keep index order (the canonical tie policy)."""
        return np.argsort(np.linalg.norm(points - x, axis=1), kind="stable")

    @staticmethod
    def within(points: np.ndarray, x: np.ndarray, r: float) -> np.ndarray:
        """Indices of `points` strictly closer than `r` to `x` (reference rrt.py:157-181)."""
        d = points - x
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        return np.atleast_1d(np.squeeze(np.argwhere(d2 < r * r)))

    @staticmethod
    def collisionfree(og, a, b) -> bool:
        """True iff every cell of the reference's line walk a -> b (rrt.py:183-229) is free.
        Uses the closed form of that walk (include/rrt_line.h)."""
        x0, y0, x1, y1 = int(a[0]), int(a[1]), int(b[0]), int(b[1])
        adx, ady = abs(x1 - x0), abs(y1 - y0)
        sx = 1 if x0 < x1 else -1
        sy = 1 if y0 < y1 else -1
        major, minor = max(adx, ady), min(adx, ady)
        k = np.arange(major + 1, dtype=np.int64)
        m = (2 * minor * k + major) // (2 * major) if major > 0 else np.zeros(1, dtype=np.int64)
        if adx >= ady:
            xs, ys = x0 + sx * k, y0 + sy * m
        else:
            xs, ys = x0 + sx * m, y0 + sy * k
        return not np.any(np.asarray(og)[xs, ys] != 0)

    # ------------------------------------------------------------------ sampling / setters
    def sample_all_free(self):
        """One uniform free-space sample (reference rrt.py:231-240)."""
        return self.free[self.rand_gen.choice(self.free.shape[0])]

    def plan(self, xstart: np.ndarray, xgoal: np.ndarray):
        raise NotImplementedError("This method is not implemented in the base class.")

    def set_og(self, og_new: np.ndarray):
        """New occupancy grid; free space is recomputed (reference rrt.py:261-272) and the grid
        is re-uploaded to the device before the next plan()."""
        self.og = og_new
        self.free = np.argwhere(og_new == 0)
        self._grid_dirty = True

    def set_n(self, n: int):
        self.n = n

    def set_og_resident(self, grids, k: int = 0):
        """Like set_og(grids.host[k]) for grids generated on this planner's device (oggen.DeviceGrids): frame k
        becomes the active grid without an upload.  Raises RuntimeError when the frames are no longer on the device
        (a set_og() + plan() or another DeviceGrids on the same context replaced them)."""
        if self._ctx is None or grids.ctx is not self._ctx:
            raise ValueError("grids were generated on a different device context: use oggen.DeviceGrids(planner.device_context(), ...)")
        grids.select(k)  # checks the context's grid generation
        self.og = grids.host[k]
        self.free = np.argwhere(self.og == 0)
        self._grid_dirty = False

    def device_context(self) -> "_ffi.Context":
        """The planner's device context (created on demand, without uploading a grid)."""
        if self._ctx is None:
            self._ctx = _ffi.Context(self.device_id)
            self._grid_dirty = True
        return self._ctx

    # ------------------------------------------------------------------ device plumbing
    def _device(self) -> "_ffi.Context":
        if self._ctx is None:
            self._ctx = _ffi.Context(self.device_id)
            self._grid_dirty = True
        if self._grid_dirty:
            self._ctx.set_grid(hostprep.og_nonzero(self.og))
            self._grid_dirty = False
        return self._ctx

    def _run(self, alg: int, xstart, xgoal, r_rewire=None, r_goal=None, logs=False, rewire=False):
        """Drive one query through the C ABI.  Returns the ResultArrays (+ ellipse log inputs)."""
        xs = hostprep.as_int_point(xstart, "xstart")
        xg = hostprep.as_int_point(xgoal, "xgoal")
        W, H = np.asarray(self.og).shape
        for p, name in ((xs, "xstart"), (xg, "xgoal")):
            if not (0 <= p[0] < W and 0 <= p[1] < H):
                raise ValueError(f"{name}={p.tolist()} lies outside the {W}x{H} occupancy grid")
        n = int(self.n)
        ctx = self._device()
        bitgen = self.rand_gen.bit_generator
        state0 = bitgen.state
        if getattr(self, "_free_packed_of", None) is not self.free:  # (free is replaced, never edited in place: set_og / set_og_resident)
            self._free_packed, self._free_packed_of = hostprep.pack_cells(self.free), self.free
        samples = hostprep.draw_free_samples_packed(self.rand_gen, self._free_packed, n)  # n draws, as rrt.py:421/502/696
        Cm = None
        if alg == _ffi.ALG_INFORMED:
            try:
                with np.errstate(all="ignore"):
                    Cm = hostprep.rotation_to_world_frame(xs, xg)
                if not np.all(np.isfinite(Cm)):
                    Cm = None
            except np.linalg.LinAlgError:
                Cm = None
        query, keep = _ffi.make_query(
            alg, n, xs, xg, samples,
            r2_rewire=hostprep.radius_threshold(r_rewire) if r_rewire is not None else 0,
            goal_d2=hostprep.goal_threshold(r_goal) if r_goal is not None else 0,
            Cmat=Cm,
        )
        kw = {"rewire": True} if rewire else {}
        rc, res = ctx.plan(query, n, logs=logs or alg == _ffi.ALG_INFORMED, **kw)
        if rc == _ffi.RRT_NEED_UNITBALL:
            # The tree reached the goal region at iteration i_switch: from there on the reference
            # draws two uniforms per iteration instead of one free-space index (rrt.py:695-700).
            # Rewind the generator to the state after exactly i_switch free draws, then draw the
            # unit-ball stream for the remaining iterations and resume on the device.
            i_sw = res.i_switch
            bitgen.state = state0
            hostprep.draw_free_samples_packed(self.rand_gen, self._free_packed, i_sw)
            if Cm is None:
                hostprep.rotation_to_world_frame(xs, xg)  # raises like rrt.py:609-612 would
                raise np.linalg.LinAlgError("rotation_to_world_frame is not finite (xstart == xgoal?)")
            ub = hostprep.draw_unitball(self.rand_gen, n - i_sw)
            rc = ctx.plan_resume(ub, res)
        if rc == _ffi.RRT_E_GOAL_UNREACHABLE:
            # rrt.py:317-318: the next argsort entry is an unfilled row -> og[INT64_MIN, ...]
            raise IndexError(f"index {INT64_MIN} is out of bounds for axis 0 with size {W}")
        res.xs, res.xg, res.Cm = xs, xg, Cm
        self.last_stats = {k: getattr(res, k) for k in ("j", "i_switch", "sum_j", "sum_cells_nn", "sum_near",
                                                          "sum_cells_cand", "n_los_cand", "n_rewired", "n_propagated")}
        return res

    def _materialise(self, res) -> Tuple[nx.DiGraph, int]:
        """points / vcosts / parents as the reference's plan() holds them after go2goal
        (rrt.py:320-325), then the DiGraph of build_graph (rrt.py:334-369)."""
        n, j, found = res.n, res.j, bool(res.found)
        rows = n + 1 if found else n
        live = j + 1 if found else j
        points = np.full((rows, 2), INT64_MIN, dtype=np.int64)  # np.full(dtype=int, fill_value=inf) rows
        vcosts = np.full((rows,), np.inf)
        points[:live] = res.pts[:live]
        vcosts[:live] = res.vcost[:live]
        vgoal = int(res.vgoal)
        if found:
            points[n] = res.xg
            vcosts[n] = vcosts[vgoal]
        return TreeDiGraph.from_arrays(vgoal, points, np.array(res.parent[:live], dtype=np.int64), vcosts), vgoal

    def build_graph(self, vgoal, points, parents, vcosts) -> nx.DiGraph:
        """DiGraph with every row of `points` as a node (`pt`) and one edge per tree link with
        `dist` (float) and `cost` (np.float64) -- node order [vgoal, 0, 1, ...] and edge order of the
        `parents` dict, as reference rrt.py:357-369.

        (The tree plan() returns is a TreeDiGraph, which fills the same dictionaries on first use.)"""
        T = nx.DiGraph()
        _fill_graph(T, vgoal, points, parents, vcosts)
        return T

    def _plan_costfn(self, alg, xstart, xgoal):
        """plan() with a caller-supplied cost function (rrt.py:55, :70-80): the callable stays in Python, so the loop runs on the
        host (rrtplanner_amd/hostloop.py, the reference's statements incl. its real rewire block) and takes near()[0],
        within() and every line of sight from the device, once per iteration (rrt_tree_query).  Not accelerated like the
        default cost, but the same device primitives; there is no CPU path."""
        from . import hostloop

        xs = hostprep.as_int_point(xstart, "xstart")
        xg = hostprep.as_int_point(xgoal, "xgoal")
        W, H = np.asarray(self.og).shape
        for p, name in ((xs, "xstart"), (xg, "xgoal")):
            if not (0 <= p[0] < W and 0 <= p[1] < H):
                raise ValueError(f"{name}={p.tolist()} lies outside the {W}x{H} occupancy grid")
        prov = getattr(self, "_costfn_provider", None)  # (tests: a stand-in for the device primitives)
        own = prov is None
        if own:
            prov = hostloop.DeviceProvider(self._device(), int(self.n))
        bar = tqdm(total=self.n) if self.pbar else None
        try:
            vgoal, points, parents, vcosts = hostloop.plan_with_costfn(self, alg, xs, xg, prov, bar)
        finally:
            if bar is not None:
                bar.close()
            if own:
                prov.close()
        return self.build_graph(vgoal, points, parents, vcosts), vgoal

    FAST_GRID_MAX = 2048    # the expansion kernels: packed 24-bit squared distances
    FAST_N_MAX = 262143     # ... and node index + chunk tag in one 32-bit key

    def _beyond_the_kernels(self) -> bool:
        W, H = np.asarray(self.og).shape
        return W > self.FAST_GRID_MAX or H > self.FAST_GRID_MAX or int(self.n) > self.FAST_N_MAX

    def _plan(self, alg, xstart, xgoal, **kw):
        if self._custom_cost or self._beyond_the_kernels():
            # the host-driven loop over the device primitives (hostloop.py): a custom cost function, or a problem larger than the
            # expansion kernels take (there with the default cost in numpy form) -- slower, same results, never a refusal
            if kw.get("rewire"):
                raise ValueError('rewire="correct" runs on the expansion kernels only (default cost, grids up to 2048 x 2048, n up to 262143)')
            return None, self._plan_costfn(alg, xstart, xgoal)
        bar = tqdm(total=self.n) if self.pbar else None
        try:
            res = self._run(alg, xstart, xgoal, **kw)
            out = self._materialise(res)
            if bar is not None:
                bar.update(self.n)
        finally:
            if bar is not None:
                bar.close()
        return res, out

    def go2goal(self, vcosts, points, xgoal, j, children, parents):
        """Connect the goal to the cheapest tree vertex that sees it (reference rrt.py:284-332), on the host arrays the
        caller passes -- the helper of the reference's public surface; plan() itself does this step on the device.
        Returns (vgoal, children, parents, points, vcosts) with the arrays grown by one row when a vertex was found, else
        vgoal = 0 like rrt.py:330-331.  Equal costs are tried in index order (the canonical tie policy)."""
        costs = np.array([self.cost(vcosts, points, i, xgoal) for i in range(points.shape[0])], dtype=np.float64).reshape(vcosts.shape)
        for idx in np.argsort(costs, kind="stable"):
            if self.collisionfree(self.og, points[idx], xgoal):
                vgoal = j
                points = np.concatenate((points, xgoal[np.newaxis, :]), axis=0)
                vcosts = np.concatenate((vcosts, [costs[idx]]), axis=0)
                points[vgoal], vcosts[vgoal] = xgoal, costs[idx]
                children[idx].append(vgoal)
                parents[vgoal] = idx
                return vgoal, children, parents, points, vcosts
        return int(np.argmin(np.linalg.norm(points - xgoal))), children, parents, points, vcosts


class RRTStandard(RRT):
    """Plain RRT: the parent of a new node is its nearest node (reference rrt.py:375-447)."""

    def __init__(self, og: np.ndarray, n: int, costfn: callable = None, pbar=True, seed: int = 0):
        super().__init__(og, n, costfn=costfn, pbar=pbar, seed=seed)

    def plan(self, xstart: np.ndarray, xgoal: np.ndarray) -> Tuple[nx.DiGraph, int]:
        return self._plan(_ffi.ALG_STANDARD, xstart, xgoal)[1]


def _rewire_mode(rewire) -> bool:
    if rewire not in ("reference", "correct"):
        raise ValueError('rewire must be "reference" (the reference\'s behaviour, default) or "correct"')
    return rewire == "correct"


class RRTStar(RRT):
    """RRT with choose-parent inside r_rewire (reference rrt.py:453-556; its rewire step never
    fires with the default cost).

    ``rewire="correct"`` (not in the reference, default ``"reference"``) opts into a true RRT* rewire with cost propagation:
    after an insertion every near vertex that gets cheaper through the new node and sees it is re-parented, and the costs
    of its descendants are recomputed.  Trees then differ from the reference's on purpose; everything else (sampling,
    acceptance, choose-parent, go2goal, the returned graph) is unchanged."""

    def __init__(self, og: np.ndarray, n: int, r_rewire: float, costfn: callable = None, pbar=True, seed: int = 0, rewire: str = "reference"):
        super().__init__(og, n, costfn=costfn, pbar=pbar, seed=seed)
        self.r_rewire = r_rewire
        self.rewire = rewire
        _rewire_mode(rewire)

    def plan(self, xstart: np.ndarray, xgoal: np.ndarray):
        return self._plan(_ffi.ALG_STAR, xstart, xgoal, r_rewire=self.r_rewire, rewire=_rewire_mode(self.rewire))[1]


class RRTStarInformed(RRT):
    """RRT* that samples the start/goal ellipse once a node lies within r_goal of the goal
    (reference rrt.py:562-758)."""

    def __init__(self, og: np.ndarray, n: int, r_rewire: float, r_goal: float, costfn: callable = None,
                 pbar: bool = True, seed: int = 0, rewire: str = "reference"):
        super().__init__(og, n, costfn=costfn, pbar=pbar, seed=seed)
        self.r_rewire = r_rewire
        self.r_goal = r_goal
        self.ellipses = {}  # j -> (xcent, major axis, minor axis, angle in degrees), for plotting
        self.rewire = rewire  # "correct": opt-in true rewire, see RRTStar
        _rewire_mode(rewire)

    def plan(self, xstart: np.ndarray, xgoal: np.ndarray):
        res, out = self._plan(_ffi.ALG_INFORMED, xstart, xgoal, r_rewire=self.r_rewire, r_goal=self.r_goal, rewire=_rewire_mode(self.rewire))
        if res is not None:  # (a custom costfn records self.ellipses in its host loop, rrt.py:701)
            self._record_ellipses(res)
        return out

    # ---- the sampler's helpers of the reference's public surface (rrt.py:579-651), on the host.  plan() draws the same
    # numbers in blocks (hostprep.draw_unitball) and applies the ellipse transform on the device.
    def unitball(self) -> np.ndarray:
        """One point of the unit disc, two uniform draws from the planner's generator (rrt.py:579-587)."""
        return hostprep.draw_unitball(self.rand_gen, 1)[0]

    def rotation_to_world_frame(self, xstart, xgoal) -> np.ndarray:
        return hostprep.rotation_to_world_frame(np.asarray(xstart), np.asarray(xgoal))

    def get_ellipse_xform(self, xstart, xgoal, cmax) -> np.ndarray:
        """C @ diag(cmax / 2, sqrt(|cmax^2 - |xstart - xgoal|^2|) / 2)   (rrt.py:615-625)"""
        d = np.asarray(xstart) - np.asarray(xgoal)
        return np.dot(self.rotation_to_world_frame(xstart, xgoal), np.diag([cmax / 2, np.sqrt(abs(cmax * cmax - np.dot(d.T, d))) / 2]))

    def sample_ellipse(self, xstart, xgoal, c, clamp=True) -> np.ndarray:
        """A sample of the ellipse with foci xstart / xgoal and path length c, clamped to the grid (rrt.py:589-599)."""
        x, y = tuple(np.dot(self.get_ellipse_xform(xstart, xgoal, c), self.unitball()) + (np.asarray(xstart) + np.asarray(xgoal)) / 2)
        if clamp:
            x = int(max(0, min(self.og.shape[0] - 1, x)))
            y = int(max(0, min(self.og.shape[1] - 1, y)))
        return np.array((x, y))

    @staticmethod
    def least_cost(vcosts, vsoln):
        """(vertex, cost) of the cheapest solution vertex, the first one among equals (rrt.py:627-633)."""
        k = int(np.argmin(vcosts[vsoln])) if len(vsoln) > 1 else 0
        return vsoln[k], vcosts[vsoln[k]]

    @staticmethod
    def rad2deg(a):
        return a * 180 / np.pi

    def get_ellipse_for_plt(self, xstart, xgoal, cmax):
        """(centre, major axis, minor axis, angle in degrees) for matplotlib's Ellipse (rrt.py:639-651)."""
        CL = self.get_ellipse_xform(xstart, xgoal, cmax)
        a, b = CL[:, 0], CL[:, 1]
        return (np.asarray(xgoal) + np.asarray(xstart)) / 2, 2 * np.linalg.norm(a), 2 * np.linalg.norm(b), self.rad2deg(np.arctan2(a[1], a[0]))

    def _record_ellipses(self, res):
        """self.ellipses[j] = get_ellipse_for_plt(...) of every ellipse iteration (rrt.py:701); a
        later iteration with the same j overwrites the value, the key keeps its first position."""
        i_sw, n = res.i_switch, res.n
        if i_sw >= n:
            return
        jl = res.j_log[i_sw:n]
        cl = res.cbest_log[i_sw:n]
        last = np.flatnonzero(np.r_[jl[1:] != jl[:-1], True])  # last iteration of each run of equal j
        xcent, maj, mnr, ang = hostprep.ellipse_plot_params(res.Cm, res.xs, res.xg, cl[last])
        for k, a, b, g in zip(jl[last].tolist(), maj, mnr, ang):
            self.ellipses[k] = (xcent.copy(), a, b, g)
