"""ctypes binding of the C ABI in include/rrt_hip.h (librrt_hip.so).

There is no CPU fallback: if the HIP library is missing or cannot be loaded, or no GPU
is visible, the functions here raise.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C rrtplanner_amd/csrc``.
"""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRT_HIP_LIB", os.path.join(_HERE, "librrt_hip.so"))  # override: diagnostic builds only

RRT_OK = 0
RRT_NEED_UNITBALL = 1
RRT_E_ARG = -1
RRT_E_GOAL_UNREACHABLE = -2
RRT_E_HIP = -3
RRT_E_NOGRID = -4
RRT_E_UNSUPPORTED = -5
RRT_E_COMM = -6

ALG_STANDARD, ALG_STAR, ALG_INFORMED = 0, 1, 2
ALG_DUBINS, ALG_DUBINS_STAR = 3, 4  # no reference counterpart (README only), see include/rrt_dubins.h
FLAG_LOGS = 1
FLAG_SERIAL = 2
FLAG_NOTEAM = 4
FLAG_TEAM_FAULT = 8
FLAG_NOPIPE = 16
FLAG_REWIRE = 32
FLAG_DUBINS = 64
FLAG_NOPIPE1 = 32768


def kernel_flags(logs=False, serial=False, team=None, team_fault=False, pipe=True, rewire=False, dubins=False, pipe1=True):
    """flags word of rrt_plan / rrt_batch_create.  team: None = as many CUs per query as fit (up to 64), 1 = one CU,
    2..64 = cap on the team's workers; pipe = False: no pipelined teams (workers + one committing CU); team_fault = the
    fault-injection flag of the tests."""
    f = (FLAG_LOGS if logs else 0) | (FLAG_SERIAL if serial else 0) | (FLAG_TEAM_FAULT if team_fault else 0) | (0 if pipe else FLAG_NOPIPE)
    f |= FLAG_REWIRE if rewire else 0  # the opt-in true rewire (not the reference's behaviour)
    f |= FLAG_DUBINS if dubins else 0
    f |= 0 if pipe1 else FLAG_NOPIPE1  # one CU per query: the block kernel instead of the barrier-free pipeline (a cross-check)
    if team == 1:
        f |= FLAG_NOTEAM
    elif team is not None:
        if team not in (2, 3, 4, 8, 16, 32, 64):
            raise ValueError("team must be None, 1, 2, 3, 4, 8, 16, 32 or 64")
        f |= int(team) << 8
    return f

# every symbol include/rrt_hip.h declares (tests/test_capi_symbols.py checks the library exports them)
SYMBOLS = (
    "rrt_ctx_create", "rrt_ctx_destroy", "rrt_last_error_string", "rrt_set_grid", "rrt_noise_grids", "rrt_select_frame",
    "rrt_grid_generation", "rrt_ctx_sync",
    "rrt_comm_use_library", "rrt_comm_unique_id", "rrt_comm_init", "rrt_comm_destroy", "rrt_comm_allreduce_f64", "rrt_gather", "rrt_gather_fetch",
    "rrt_batch_create", "rrt_batch_destroy", "rrt_batch_set_query", "rrt_batch_set_unitball",
    "rrt_batch_rearm", "rrt_batch_launch", "rrt_batch_sync", "rrt_batch_team", "rrt_batch_team_info", "rrt_batch_pipelined", "rrt_batch_kernel_name", "rrt_batch_elapsed_ms",
    "rrt_batch_get_result", "rrt_batch_result_block", "rrt_batch_debug_cycles",
    "rrt_plan", "rrt_plan_resume", "rrt_plan_batch",
    "rrt_tree_create", "rrt_tree_destroy", "rrt_tree_reset", "rrt_tree_append", "rrt_tree_query",
    "rrt_prim_collisionfree", "rrt_prim_nearest_within", "rrt_prim_sqrt_u32", "rrt_prim_sqrt_u24", "rrt_prim_sqrt_f64",
)


class RRTError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librrt_hip error {code}: {msg}")
        self.code = code


class Query(C.Structure):
    _fields_ = [
        ("alg", C.c_int32), ("n", C.c_int32),
        ("xs", C.c_int32 * 2), ("xg", C.c_int32 * 2),
        ("r2_rewire", C.c_int64), ("goal_d2", C.c_int64),
        ("samples", C.c_void_p),
        ("C", C.c_double * 4),
        ("headings", C.c_void_p), ("rho", C.c_double), ("nh", C.c_int32), ("hs", C.c_int32), ("hg", C.c_int32), ("pad_", C.c_int32),
        ("samples_packed", C.c_void_p),
    ]


class Result(C.Structure):
    _fields_ = [
        ("pts", C.c_void_p), ("vcost", C.c_void_p), ("parent", C.c_void_p),
        ("nearest_log", C.c_void_p), ("accept_log", C.c_void_p), ("cbest_log", C.c_void_p), ("j_log", C.c_void_p),
        ("status", C.c_int32), ("j", C.c_int32), ("vgoal", C.c_int32), ("found", C.c_int32),
        ("i_switch", C.c_int32), ("rows", C.c_int32),
        ("sum_j", C.c_int64), ("sum_cells_nn", C.c_int64), ("sum_near", C.c_int64), ("sum_cells_cand", C.c_int64),
        ("n_los_cand", C.c_int64), ("n_rewired", C.c_int64), ("n_propagated", C.c_int64),
        ("head", C.c_void_p),
        ("n_words", C.c_int64),
    ]


_lib = None


def lib():
    """Load librrt_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  rrtplanner_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        vp, i32, u32, i64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_int64
        sig = {
            "rrt_ctx_create": ([i32, C.POINTER(vp)], C.c_int),
            "rrt_ctx_destroy": ([vp], C.c_int),
            "rrt_last_error_string": ([vp], C.c_char_p),
            "rrt_set_grid": ([vp, vp, i32, i32], C.c_int),
            "rrt_noise_grids": ([vp, i32, i32, i32, C.c_float, i32, vp, vp, vp, vp, vp], C.c_int),
            "rrt_select_frame": ([vp, i32], C.c_int),
            "rrt_grid_generation": ([vp, C.POINTER(C.c_uint64)], C.c_int),
            "rrt_ctx_sync": ([vp], C.c_int),
            "rrt_comm_use_library": ([C.c_char_p], C.c_int),
            "rrt_comm_unique_id": ([vp], C.c_int),
            "rrt_comm_init": ([vp, i32, i32, vp], C.c_int),
            "rrt_comm_destroy": ([vp], C.c_int),
            "rrt_comm_allreduce_f64": ([vp, vp, i32, i32], C.c_int),
            "rrt_gather": ([vp, C.POINTER(vp), C.POINTER(i64)], C.c_int),
            "rrt_gather_fetch": ([vp, i32, i32, C.POINTER(Result)], C.c_int),
            "rrt_batch_create": ([vp, i32, i32, u32, C.POINTER(vp)], C.c_int),
            "rrt_batch_destroy": ([vp], C.c_int),
            "rrt_batch_set_query": ([vp, i32, C.POINTER(Query)], C.c_int),
            "rrt_batch_set_unitball": ([vp, i32, vp, i32, i32], C.c_int),
            "rrt_batch_rearm": ([vp], C.c_int),
            "rrt_batch_launch": ([vp], C.c_int),
            "rrt_batch_sync": ([vp], C.c_int),
            "rrt_batch_team": ([vp, C.POINTER(i32), C.POINTER(i32)], C.c_int),
            "rrt_batch_team_info": ([vp, C.POINTER(i32 * 4)], C.c_int),
            "rrt_batch_pipelined": ([vp, C.POINTER(i32)], C.c_int),
            "rrt_batch_kernel_name": ([vp, C.c_char_p, i32], C.c_int),
            "rrt_batch_elapsed_ms": ([vp, C.POINTER(C.c_float)], C.c_int),
            "rrt_batch_get_result": ([vp, i32, C.POINTER(Result)], C.c_int),
            "rrt_batch_result_block": ([vp, C.POINTER(vp), C.POINTER(i64)], C.c_int),
            "rrt_batch_debug_cycles": ([vp, i32, C.POINTER(C.c_uint64 * 38)], C.c_int),
            "rrt_plan": ([vp, C.POINTER(Query), u32, C.POINTER(Result)], C.c_int),
            "rrt_plan_resume": ([vp, vp, i32, C.POINTER(Result)], C.c_int),
            "rrt_plan_batch": ([vp, i32, C.POINTER(Query), C.POINTER(Result)], C.c_int),
            "rrt_tree_create": ([vp, i32, C.POINTER(vp)], C.c_int),
            "rrt_tree_destroy": ([vp], C.c_int),
            "rrt_tree_reset": ([vp], C.c_int),
            "rrt_tree_append": ([vp, i32, i32, C.POINTER(i32)], C.c_int),
            "rrt_tree_query": ([vp, i32, i32, i64, C.POINTER(i32), C.POINTER(i32), vp, vp, i32], C.c_int),
            "rrt_prim_collisionfree": ([vp, vp, i32, vp, vp], C.c_int),
            "rrt_prim_nearest_within": ([vp, vp, i32, vp, i32, i64, vp, vp, vp], C.c_int),
            "rrt_prim_sqrt_u32": ([vp, u32, u32, vp], C.c_int),
            "rrt_prim_sqrt_u24": ([vp, u32, u32, vp], C.c_int),
            "rrt_prim_sqrt_f64": ([vp, vp, u32, vp], C.c_int),
        }
        for name, (argtypes, restype) in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = L
    return _lib


def _check(ctx_handle, rc, ok=(RRT_OK,)):
    if rc in ok:
        return rc
    msg = lib().rrt_last_error_string(ctx_handle)
    raise RRTError(rc, msg.decode() if msg else "")


COMM_ID_BYTES = 128


def comm_use_library(path):
    """Before the first communicator of the process: open `path` instead of librccl.so.1 (None: the default).  For ranks that
    share one GPU (tests/fake_rccl); see include/rrt_hip.h."""
    _check(None, lib().rrt_comm_use_library(None if path is None else os.fsencode(path)))


def comm_unique_id() -> bytes:
    """A fresh RCCL communicator id (rank 0 creates it and ships it to the other ranks, see multi.exchange_unique_id)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(None, lib().rrt_comm_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


class ResultArrays:
    """Host buffers for one query's result + the filled-in rrt_result."""

    def __init__(self, n, logs=False, headings=False):
        self.n = n
        if headings:
            self.head = np.zeros(n + 1, dtype=np.int32)
        self.pts = np.zeros((n + 1, 2), dtype=np.int32)
        self.vcost = np.zeros(n + 1, dtype=np.float64)
        self.parent = np.full(n + 1, -1, dtype=np.int32)
        self.c = Result()
        self.c.pts, self.c.vcost, self.c.parent = self.pts.ctypes.data, self.vcost.ctypes.data, self.parent.ctypes.data
        if headings:
            self.c.head = self.head.ctypes.data
        if logs:
            self.nearest_log = np.full(n, -1, dtype=np.int32)
            self.accept_log = np.zeros(n, dtype=np.uint8)
            self.cbest_log = np.full(n, np.nan)
            self.j_log = np.zeros(n, dtype=np.int32)
            self.c.nearest_log, self.c.accept_log = self.nearest_log.ctypes.data, self.accept_log.ctypes.data
            self.c.cbest_log, self.c.j_log = self.cbest_log.ctypes.data, self.j_log.ctypes.data

    def __getattr__(self, k):  # scalars live in the C struct
        if k in ("status", "j", "vgoal", "found", "i_switch", "rows", "sum_j", "sum_cells_nn", "sum_near",
                 "sum_cells_cand", "n_los_cand", "n_rewired", "n_propagated", "n_words"):
            return getattr(self.c, k)
        raise AttributeError(k)


def make_query(alg, n, xs, xg, samples, r2_rewire=0, goal_d2=0, Cmat=None, headings=None, rho=0.0, nh=0):
    """Build an rrt_query; returns (Query, keepalive).  Dubins queries (alg 3 / 4): xs / xg are (x, y, heading index),
    `headings` the heading index of every sample, rho the turning radius in cells, nh the number of headings."""
    q = Query()
    if isinstance(samples, np.ndarray) and samples.dtype == np.uint32 and samples.shape == (n,):
        s = np.ascontiguousarray(samples)  # already packed: x | y << 16
        q.samples_packed = s.ctypes.data
    else:
        s = np.ascontiguousarray(samples, dtype=np.int32)
        if s.shape != (n, 2):
            raise ValueError(f"samples must have shape ({n}, 2), got {s.shape}")
        q.samples = s.ctypes.data
    q.alg, q.n = int(alg), int(n)
    q.xs[0], q.xs[1] = int(xs[0]), int(xs[1])
    q.xg[0], q.xg[1] = int(xg[0]), int(xg[1])
    q.r2_rewire, q.goal_d2 = int(r2_rewire), int(goal_d2)
    if Cmat is not None:
        cm = np.asarray(Cmat, dtype=np.float64).reshape(4)
        for k in range(4):
            q.C[k] = float(cm[k])
    if headings is not None:
        hd = np.ascontiguousarray(headings, dtype=np.int32)
        if hd.shape != (n,):
            raise ValueError(f"headings must have shape ({n},), got {hd.shape}")
        q.headings, q.rho, q.nh, q.hs, q.hg = hd.ctypes.data, float(rho), int(nh), int(xs[2]), int(xg[2])
        return q, (s, hd)
    return q, s


class Context:
    """One device context: a HIP stream + the device-resident occupancy grid."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        rc = lib().rrt_ctx_create(int(device_id), C.byref(self._h))
        _check(None, rc)
        self.shape = None

    @property
    def handle(self):
        return self._h

    def close(self):
        for b in list(getattr(self, "_batches", ())):  # batches hold a pointer to the context: they go first
            b.close()
        if self._h:
            lib().rrt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_grid(self, og_nonzero):
        g = np.ascontiguousarray(og_nonzero, dtype=np.uint8)
        if g.ndim != 2:
            raise ValueError("occupancy grid must be 2-D")
        _check(self._h, lib().rrt_set_grid(self._h, g.ctypes.data, g.shape[0], g.shape[1]))
        self.shape = g.shape

    def noise_grids(self, W, H, frames, thresh, dims, cells, amps, grads):
        """Generate `frames` noise grids on the device; returns the host copy (frames, W, H) uint8."""
        dims = np.ascontiguousarray(dims, dtype=np.int32).reshape(-1, 3)
        cells = np.ascontiguousarray(cells, dtype=np.float64)
        amps = np.ascontiguousarray(amps, dtype=np.float64)
        grads = np.ascontiguousarray(grads, dtype=np.float64)
        out = np.zeros((frames, W, H), dtype=np.uint8)
        _check(self._h, lib().rrt_noise_grids(self._h, int(W), int(H), int(frames), float(thresh), dims.shape[0], dims.ctypes.data,
                                              cells.ctypes.data, amps.ctypes.data, grads.ctypes.data, out.ctypes.data))
        self.shape = (W, H)
        return out

    def select_frame(self, k):
        _check(self._h, lib().rrt_select_frame(self._h, int(k)))

    def grid_generation(self):
        """How often the context's grid storage has been rewritten (set_grid / noise_grids); resident frames are only
        valid while this has the value it had right after they were generated."""
        g = C.c_uint64(0)
        _check(self._h, lib().rrt_grid_generation(self._h, C.byref(g)))
        return g.value

    def sync(self):
        _check(self._h, lib().rrt_ctx_sync(self._h))

    # ---- multi-GPU (RCCL) ----
    def comm_init(self, rank, world, unique_id: bytes):
        """Collective: join the communicator `unique_id` (comm_unique_id() of rank 0) as `rank` of `world`."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique id must have {COMM_ID_BYTES} bytes")
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(self._h, lib().rrt_comm_init(self._h, int(rank), int(world), C.cast(buf, C.c_void_p)))
        self.rank, self.world = int(rank), int(world)

    def comm_destroy(self):
        _check(self._h, lib().rrt_comm_destroy(self._h))

    def allreduce(self, values, op="sum"):
        """All-reduce a few float64 values over the ranks (sum / max / min); synchronous, so it doubles as the barrier."""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        _check(self._h, lib().rrt_comm_allreduce_f64(self._h, v.ctypes.data, v.size, {"sum": 0, "max": 1, "min": 2}[op]))
        return v

    def barrier(self):
        self.allreduce([0.0])

    # ---- one-shot ----
    def plan(self, query, n, logs=False, serial=False, team=None, team_fault=False, pipe=True, rewire=False, pipe1=True):
        res = ResultArrays(n, logs, headings=query.alg >= ALG_DUBINS)
        flags = kernel_flags(logs, serial, team, team_fault, pipe, rewire, pipe1=pipe1)
        rc = lib().rrt_plan(self._h, C.byref(query), flags, C.byref(res.c))
        _check(self._h, rc, ok=(RRT_OK, RRT_NEED_UNITBALL, RRT_E_GOAL_UNREACHABLE))
        return rc, res

    def plan_resume(self, unitball, res):
        ub = np.ascontiguousarray(unitball, dtype=np.float64)
        rc = lib().rrt_plan_resume(self._h, ub.ctypes.data, ub.shape[0], C.byref(res.c))
        _check(self._h, rc, ok=(RRT_OK, RRT_NEED_UNITBALL, RRT_E_GOAL_UNREACHABLE))
        return rc

    def plan_batch(self, queries, ns):
        """rrt_plan_batch: Q independent queries on this context's grid in one call (RRTStandard / RRTStar; an Informed
        query stops at RRT_NEED_UNITBALL -- use Batch for the staged hand-over).  Returns (rc, [ResultArrays])."""
        Q = len(queries)
        qarr = (Query * Q)(*queries)
        res = [ResultArrays(int(n)) for n in ns]
        rarr = (Result * Q)(*[r.c for r in res])
        rc = lib().rrt_plan_batch(self._h, Q, qarr, rarr)
        _check(self._h, rc, ok=(RRT_OK, RRT_NEED_UNITBALL, RRT_E_GOAL_UNREACHABLE))
        for r, c in zip(res, rarr):
            r.c = c
        return rc, res

    # ---- primitives ----
    def prim_collisionfree(self, ab):
        ab = np.ascontiguousarray(ab, dtype=np.int32).reshape(-1, 4)
        m = ab.shape[0]
        free = np.zeros(m, dtype=np.uint8)
        cells = np.zeros(m, dtype=np.int32)
        _check(self._h, lib().rrt_prim_collisionfree(self._h, ab.ctypes.data, m, free.ctypes.data, cells.ctypes.data))
        return free.astype(bool), cells

    def prim_nearest_within(self, pts, xq, r2):
        pts = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, 2)
        xq = np.ascontiguousarray(xq, dtype=np.int32).reshape(-1, 2)
        m = xq.shape[0]
        nn = np.zeros(m, dtype=np.int32)
        cnt = np.zeros(m, dtype=np.int32)
        isum = np.zeros(m, dtype=np.int64)
        _check(self._h, lib().rrt_prim_nearest_within(self._h, pts.ctypes.data, pts.shape[0], xq.ctypes.data, m, int(r2),
                                                      nn.ctypes.data, cnt.ctypes.data, isum.ctypes.data))
        return nn, cnt, isum

    def prim_sqrt_u32(self, lo, count):
        out = np.zeros(count, dtype=np.float64)
        _check(self._h, lib().rrt_prim_sqrt_u32(self._h, int(lo), int(count), out.ctypes.data))
        return out

    def prim_sqrt_u24(self, lo, count):
        out = np.zeros(count, dtype=np.float64)
        _check(self._h, lib().rrt_prim_sqrt_u24(self._h, int(lo), int(count), out.ctypes.data))
        return out

    def prim_sqrt_f64(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        _check(self._h, lib().rrt_prim_sqrt_f64(self._h, x.ctypes.data, x.size, out.ctypes.data))
        return out


class DeviceTree:
    """The device-resident vertex list of a host-driven planner (rrt_tree_*): the planner keeps its loop -- and its Python
    cost function -- on the host and asks the device once per iteration for near()[0], within() and the lines of sight."""

    def __init__(self, ctx: Context, capacity: int):
        self.ctx, self.capacity = ctx, int(capacity)
        self._h = C.c_void_p()
        _check(ctx.handle, lib().rrt_tree_create(ctx.handle, self.capacity, C.byref(self._h)))
        if not hasattr(ctx, "_batches"):
            ctx._batches = weakref.WeakSet()
        ctx._batches.add(self)  # closed before the context, like a batch
        self._idx = np.zeros(self.capacity, dtype=np.int32)
        self._los = np.zeros(self.capacity + 1, dtype=np.uint8)

    def close(self):
        if self._h:
            lib().rrt_tree_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _check(self.ctx.handle, lib().rrt_tree_reset(self._h))

    def append(self, x, y) -> int:
        j = C.c_int32(-1)
        _check(self.ctx.handle, lib().rrt_tree_append(self._h, int(x), int(y), C.byref(j)))
        return j.value

    def query(self, x, y, r2):
        """(nearest vertex, ascending indices of the vertices with d2 < r2, line of sight nearest -> (x, y),
        lines of sight of those vertices -> (x, y))"""
        nn, cnt = C.c_int32(-1), C.c_int32(0)
        _check(self.ctx.handle, lib().rrt_tree_query(self._h, int(x), int(y), int(r2), C.byref(nn), C.byref(cnt), self._idx.ctypes.data,
                                                     self._los.ctypes.data, self.capacity))
        m = cnt.value
        return nn.value, self._idx[:m].copy(), bool(self._los[0]), self._los[1:1 + m].astype(bool)


class Batch:
    """Q independent queries resident on the device (rrt_batch_*)."""

    def __init__(self, ctx: Context, Q: int, n_cap: int, logs: bool = False, serial: bool = False, team=None, team_fault: bool = False,
                 pipe: bool = True, rewire: bool = False, dubins: bool = False, pipe1: bool = True):
        self.ctx, self.Q, self.n_cap, self.logs, self.dubins = ctx, int(Q), int(n_cap), logs, dubins
        self._h = C.c_void_p()
        flags = kernel_flags(logs, serial, team, team_fault, pipe, rewire, dubins, pipe1)
        _check(ctx.handle, lib().rrt_batch_create(ctx.handle, self.Q, self.n_cap, flags, C.byref(self._h)))
        if not hasattr(ctx, "_batches"):
            ctx._batches = weakref.WeakSet()
        ctx._batches.add(self)
        self._n = [0] * self.Q

    def close(self):
        if self._h:
            lib().rrt_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_query(self, q, query):
        _check(self.ctx.handle, lib().rrt_batch_set_query(self._h, int(q), C.byref(query)))
        self._n[q] = query.n

    def set_unitball(self, q, unitball, ub_offset):
        ub = np.ascontiguousarray(unitball, dtype=np.float64)
        _check(self.ctx.handle, lib().rrt_batch_set_unitball(self._h, int(q), ub.ctypes.data, ub.shape[0], int(ub_offset)))

    def rearm(self):
        _check(self.ctx.handle, lib().rrt_batch_rearm(self._h))

    def launch(self):
        _check(self.ctx.handle, lib().rrt_batch_launch(self._h))

    def sync(self):
        _check(self.ctx.handle, lib().rrt_batch_sync(self._h))

    def team(self):
        """(CUs per query, launches repeated with one CU per query after a team hand-off timed out)"""
        g, f = C.c_int32(0), C.c_int32(0)
        _check(self.ctx.handle, lib().rrt_batch_team(self._h, C.byref(g), C.byref(f)))
        return g.value, f.value

    def team_info(self):
        """dict: workers per query at creation / of the last launch, hand-off timeouts, launches that ran a smaller team because
        other launches held compute units of the device"""
        out = (C.c_int32 * 4)()
        _check(self.ctx.handle, lib().rrt_batch_team_info(self._h, C.byref(out)))
        return dict(created=out[0], last=out[1], timeouts=out[2], shrunk=out[3])

    def pipelined(self):
        """True if the last launch ran the pipelined team kernel"""
        v = C.c_int32(0)
        _check(self.ctx.handle, lib().rrt_batch_pipelined(self._h, C.byref(v)))
        return bool(v.value)

    def kernel_name(self):
        """the expansion kernel of the last launch, as rocprofv3 names it"""
        buf = C.create_string_buffer(128)
        _check(self.ctx.handle, lib().rrt_batch_kernel_name(self._h, buf, 128))
        return buf.value.decode()

    def elapsed_ms(self):
        ms = C.c_float(0)
        _check(self.ctx.handle, lib().rrt_batch_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def get_result(self, q, arrays=True):
        res = ResultArrays(self._n[q], self.logs, headings=self.dubins) if arrays else None
        if res is None:
            res = ResultArrays.__new__(ResultArrays)
            res.n = self._n[q]
            res.c = Result()
        rc = lib().rrt_batch_get_result(self._h, int(q), C.byref(res.c))
        _check(self.ctx.handle, rc, ok=(RRT_OK, RRT_E_GOAL_UNREACHABLE))
        return res

    def debug_cycles(self, q):
        out = (C.c_uint64 * 38)()
        _check(self.ctx.handle, lib().rrt_batch_debug_cycles(self._h, int(q), C.byref(out)))
        return list(out)

    def result_block(self):
        p, nbytes = C.c_void_p(), C.c_int64()
        _check(self.ctx.handle, lib().rrt_batch_result_block(self._h, C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    def gather(self):
        """All-gather the result slabs of this batch over the context's communicator (asynchronous on its stream).
        Returns (device pointer of the gathered slabs, bytes per rank)."""
        p, nbytes = C.c_void_p(), C.c_int64()
        _check(self.ctx.handle, lib().rrt_gather(self._h, C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    def gather_fetch(self, rank, q):
        """Query q of rank `rank` out of the gathered slabs (host arrays; status / j / vgoal / found filled).  The arrays
        always hold the batch's full capacity (n_cap + 1 rows): how many rows the remote query has is only known from its
        slab, and the C side refuses to write more rows than `rows` announces."""
        res = ResultArrays(self.n_cap)
        res.c.rows = self.n_cap + 1
        _check(self.ctx.handle, lib().rrt_gather_fetch(self._h, int(rank), int(q), C.byref(res.c)))
        return res
