"""ctypes binding of the CPU oracle (oracle/rrt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under rrtplanner_amd/ imports this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

ORC_OK = 0
ORC_NEED_UNITBALL = 1
ORC_E_ARG = -1
ORC_E_GOAL_UNREACHABLE = -2


def build(force: bool = False) -> str:
    """Compile rrt_oracle.c -> liboracle.so with gcc (in-tree, git-ignored)."""
    srcs = [os.path.join(_HERE, "rrt_oracle.c"), os.path.join(_HERE, "dubins_oracle.c"),
            os.path.join(os.path.dirname(_HERE), "include", "rrt_dubins.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _Plan(C.Structure):
    _fields_ = [
        ("alg", C.c_int32), ("n", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("og", C.c_void_p),
        ("xs", C.c_int32 * 2), ("xg", C.c_int32 * 2),
        ("r2_rewire", C.c_int64), ("r_goal", C.c_double),
        ("samples", C.c_void_p), ("unitball", C.c_void_p), ("ub_offset", C.c_int32),
        ("C", C.c_double * 4),
        ("pts", C.c_void_p), ("vcost", C.c_void_p), ("parent", C.c_void_p),
        ("nearest_log", C.c_void_p), ("accept_log", C.c_void_p), ("cbest_log", C.c_void_p), ("jlog", C.c_void_p),
        ("j", C.c_int32), ("vgoal", C.c_int32), ("found", C.c_int32), ("i_switch", C.c_int32), ("rows", C.c_int32),
        ("sum_j", C.c_int64), ("sum_cells_nn", C.c_int64), ("sum_near", C.c_int64), ("sum_cells_cand", C.c_int64),
        ("n_rewired", C.c_int64),
        ("rewire", C.c_int32), ("pad_", C.c_int32), ("n_propagated", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_plan.argtypes = [C.POINTER(_Plan)]
        _lib.orc_plan.restype = C.c_int
        _lib.orc_collisionfree.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.POINTER(C.c_int64)]
        _lib.orc_collisionfree.restype = C.c_int
        _lib.orc_bresenham_cells.argtypes = [C.c_int32] * 4 + [C.c_void_p, C.c_int32]
        _lib.orc_bresenham_cells.restype = C.c_int32
        _lib.orc_nearest.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        _lib.orc_nearest.restype = C.c_int32
        _lib.orc_within.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
        _lib.orc_within.restype = C.c_int32
        _lib.orc_sample_ellipse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double,
                                            C.c_double, C.c_double, C.c_void_p]
        _lib.orc_sample_ellipse.restype = None
    return _lib


def og_u8(og) -> np.ndarray:
    """The reference treats anything != 0 as an obstacle (rrt.py:191,218)."""
    return np.ascontiguousarray((np.asarray(og) != 0).astype(np.uint8))


def collisionfree(og8, a, b):
    """(free: bool, cells_read: int) of rrt.py:202-229 from a to b."""
    cells = C.c_int64(0)
    r = lib().orc_collisionfree(og8.ctypes.data, og8.shape[0], og8.shape[1], int(a[0]), int(a[1]), int(b[0]), int(b[1]),
                                C.byref(cells))
    return bool(r), cells.value


def bresenham_cells(a, b):
    cap = max(abs(int(b[0]) - int(a[0])), abs(int(b[1]) - int(a[1]))) + 1
    out = np.empty((cap, 2), dtype=np.int32)
    c = lib().orc_bresenham_cells(int(a[0]), int(a[1]), int(b[0]), int(b[1]), out.ctypes.data, cap)
    assert c == cap
    return out


def nearest(pts, x):
    p = np.ascontiguousarray(pts, dtype=np.int32)
    return lib().orc_nearest(p.ctypes.data, p.shape[0], int(x[0]), int(x[1]))


def within(pts, x, r2):
    p = np.ascontiguousarray(pts, dtype=np.int32)
    out = np.empty(p.shape[0], dtype=np.int32)
    m = lib().orc_within(p.ctypes.data, p.shape[0], int(x[0]), int(x[1]), int(r2), out.ctypes.data)
    return out[:m].copy()


def sample_ellipse(Cm, xs, xg, W, H, c, u):
    Cm = np.ascontiguousarray(Cm, dtype=np.float64).reshape(4)
    xs = np.ascontiguousarray(xs, dtype=np.int32)
    xg = np.ascontiguousarray(xg, dtype=np.int32)
    out = np.zeros(2, dtype=np.int32)
    lib().orc_sample_ellipse(Cm.ctypes.data, xs.ctypes.data, xg.ctypes.data, W, H, float(c), float(u[0]), float(u[1]),
                             out.ctypes.data)
    return out


class PlanResult:
    pass


def plan(og8, n, alg, xs, xg, samples, r2_rewire=0, r_goal=0.0, unitball=None, ub_offset=0, Cmat=None, logs=True, rewire=False):
    """Run orc_plan once.  Returns (status, PlanResult).  rewire=True: the opt-in true RRT* rewire (not the reference's)."""
    og8 = np.ascontiguousarray(og8, dtype=np.uint8)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    assert samples.shape == (n, 2)
    p = _Plan()
    p.alg, p.n, p.W, p.H = alg, n, og8.shape[0], og8.shape[1]
    p.og = og8.ctypes.data
    p.xs[0], p.xs[1] = int(xs[0]), int(xs[1])
    p.xg[0], p.xg[1] = int(xg[0]), int(xg[1])
    p.r2_rewire, p.r_goal = int(r2_rewire), float(r_goal)
    p.samples = samples.ctypes.data
    p.rewire = 1 if rewire else 0
    if unitball is not None:
        unitball = np.ascontiguousarray(unitball, dtype=np.float64)
        p.unitball = unitball.ctypes.data
    p.ub_offset = ub_offset
    if Cmat is not None:
        cm = np.ascontiguousarray(Cmat, dtype=np.float64).reshape(4)
        for k in range(4):
            p.C[k] = cm[k]
    r = PlanResult()
    r.pts = np.empty((n + 1, 2), dtype=np.int32)
    r.vcost = np.empty(n + 1, dtype=np.float64)
    r.parent = np.empty(n + 1, dtype=np.int32)
    p.pts, p.vcost, p.parent = r.pts.ctypes.data, r.vcost.ctypes.data, r.parent.ctypes.data
    if logs:
        r.nearest_log = np.full(n, -1, dtype=np.int32)
        r.accept_log = np.zeros(n, dtype=np.uint8)
        r.cbest_log = np.full(n, np.nan)
        r.jlog = np.zeros(n, dtype=np.int32)
        p.nearest_log, p.accept_log = r.nearest_log.ctypes.data, r.accept_log.ctypes.data
        p.cbest_log, p.jlog = r.cbest_log.ctypes.data, r.jlog.ctypes.data
    status = lib().orc_plan(C.byref(p))
    for k in ("j", "vgoal", "found", "i_switch", "rows", "sum_j", "sum_cells_nn", "sum_near", "sum_cells_cand",
              "n_rewired", "n_propagated"):
        setattr(r, k, getattr(p, k))
    return status, r


# ---------------------------------------------------------------------------------------------- Dubins (no reference parity)
class _DubPlan(C.Structure):
    _fields_ = [
        ("star", C.c_int32), ("n", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("og", C.c_void_p),
        ("xs", C.c_int32 * 3), ("xg", C.c_int32 * 3),
        ("r2_rewire", C.c_int64), ("rho", C.c_double), ("nh", C.c_int32), ("pad_", C.c_int32),
        ("samples", C.c_void_p), ("headings", C.c_void_p),
        ("pts", C.c_void_p), ("head", C.c_void_p), ("vcost", C.c_void_p), ("parent", C.c_void_p),
        ("nearest_log", C.c_void_p), ("accept_log", C.c_void_p),
        ("j", C.c_int32), ("vgoal", C.c_int32), ("found", C.c_int32), ("rows", C.c_int32),
        ("sum_j", C.c_int64), ("sum_cells_nn", C.c_int64), ("sum_near", C.c_int64), ("sum_cells_cand", C.c_int64),
        ("n_dubins", C.c_int64),
        ("counters", C.c_int32), ("pad2_", C.c_int32),
        ("near_of_rejected", C.c_int64), ("lb_static_skip", C.c_int64), ("lb_evals", C.c_int64), ("lb_sweeps", C.c_int64),
        ("lb_violations", C.c_int64), ("lb_mismatch", C.c_int64),
    ]


def _dub_lib():
    L = lib()
    if not getattr(L, "_dub_bound", False):
        L.orc_dubins_plan.argtypes = [C.POINTER(_DubPlan)]
        L.orc_dubins_plan.restype = C.c_int
        L.orc_dub_shortest.argtypes = [C.c_double] * 7 + [C.c_void_p]
        L.orc_dub_shortest.restype = None
        L.orc_dub_sweep_cells.argtypes = [C.c_double] * 7 + [C.c_void_p, C.c_int32]
        L.orc_dub_sweep_cells.restype = C.c_int32
        L.orc_dub_sincos.argtypes = [C.c_double, C.c_void_p]
        L.orc_dub_sincos.restype = None
        L.orc_dub_atan2.argtypes = [C.c_double, C.c_double]
        L.orc_dub_atan2.restype = C.c_double
        L._dub_bound = True
    return L


def dub_shortest(x0, y0, th0, x1, y1, th1, rho):
    """(t, p, q, length, word) of include/rrt_dubins.h's shortest word."""
    out = np.zeros(5)
    _dub_lib().orc_dub_shortest(float(x0), float(y0), float(th0), float(x1), float(y1), float(th1), float(rho), out.ctypes.data)
    return out[0], out[1], out[2], out[3], int(out[4])


def dub_sweep_cells(x0, y0, th0, x1, y1, th1, rho, cap=100000):
    out = np.zeros((cap, 2), dtype=np.int32)
    m = _dub_lib().orc_dub_sweep_cells(float(x0), float(y0), float(th0), float(x1), float(y1), float(th1), float(rho), out.ctypes.data, cap)
    return out[:m].copy()


def dub_sincos(a):
    out = np.zeros(2)
    _dub_lib().orc_dub_sincos(float(a), out.ctypes.data)
    return out[0], out[1]


def dub_atan2(y, x):
    return _dub_lib().orc_dub_atan2(float(y), float(x))


def dubins_plan(og8, n, star, xs, xg, samples, headings, r2_rewire=0, rho=8.0, nh=64, logs=True, counters=False):
    """Run orc_dubins_plan once: xs / xg are (x, y, heading index).  Returns (status, PlanResult)."""
    og8 = np.ascontiguousarray(og8, dtype=np.uint8)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    headings = np.ascontiguousarray(headings, dtype=np.int32)
    assert samples.shape == (n, 2) and headings.shape == (n,)
    p = _DubPlan()
    p.star, p.n, p.W, p.H = int(bool(star)), n, og8.shape[0], og8.shape[1]
    p.og = og8.ctypes.data
    for k in range(3):
        p.xs[k], p.xg[k] = int(xs[k]), int(xg[k])
    p.r2_rewire, p.rho, p.nh = int(r2_rewire), float(rho), int(nh)
    p.counters = 1 if counters else 0  # study of the chord lower bound (see dubins_oracle.c), results unchanged
    p.samples, p.headings = samples.ctypes.data, headings.ctypes.data
    r = PlanResult()
    r.pts = np.empty((n + 1, 2), dtype=np.int32)
    r.head = np.empty(n + 1, dtype=np.int32)
    r.vcost = np.empty(n + 1, dtype=np.float64)
    r.parent = np.empty(n + 1, dtype=np.int32)
    p.pts, p.head, p.vcost, p.parent = r.pts.ctypes.data, r.head.ctypes.data, r.vcost.ctypes.data, r.parent.ctypes.data
    if logs:
        r.nearest_log = np.full(n, -1, dtype=np.int32)
        r.accept_log = np.zeros(n, dtype=np.uint8)
        p.nearest_log, p.accept_log = r.nearest_log.ctypes.data, r.accept_log.ctypes.data
    status = _dub_lib().orc_dubins_plan(C.byref(p))
    for k in ("j", "vgoal", "found", "rows", "sum_j", "sum_cells_nn", "sum_near", "sum_cells_cand", "n_dubins",
              "near_of_rejected", "lb_static_skip", "lb_evals", "lb_sweeps", "lb_violations", "lb_mismatch"):
        setattr(r, k, getattr(p, k))
    return status, r


# ---------------------------------------------------------------------------------- independent Dubins check (oracle/dubins_ref.c)
_DUBREF_SO = os.path.join(_HERE, "libdubref.so")
_dubref = None


class _DubAudit(C.Structure):
    _fields_ = [
        ("star", C.c_int32), ("n", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("og", C.c_void_p), ("r2_rewire", C.c_int64), ("rho", C.c_double), ("nh", C.c_int32), ("pad_", C.c_int32),
        ("samples", C.c_void_p), ("headings", C.c_void_p),
        ("pts", C.c_void_p), ("head", C.c_void_p), ("vcost", C.c_void_p), ("parent", C.c_void_p),
        ("j", C.c_int32), ("pad2_", C.c_int32),
        ("n_accepted", C.c_int64), ("accept_mismatch", C.c_int64), ("accept_ambiguous", C.c_int64), ("nearest_mismatch", C.c_int64),
        ("parent_is_argmin", C.c_int64), ("parent_within_tol", C.c_int64), ("parent_wrong", C.c_int64), ("parent_blocked", C.c_int64),
        ("parent_blocked_ambiguous", C.c_int64), ("cost_mismatch", C.c_int64), ("max_cost_err", C.c_double),
        ("words", C.c_int64), ("sweeps", C.c_int64), ("first_bad_iter", C.c_int64),
    ]


def dubref_lib():
    """libdubref.so: textbook Dubins words on libm with their own sweep; does NOT include include/rrt_dubins.h."""
    global _dubref
    if _dubref is None:
        src = os.path.join(_HERE, "dubins_ref.c")
        if not os.path.exists(_DUBREF_SO) or os.path.getmtime(_DUBREF_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libdubref.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(_DUBREF_SO)
        L.dubref_audit.argtypes = [C.POINTER(_DubAudit)]
        L.dubref_audit.restype = C.c_int
        L.dubref_shortest.argtypes = [C.c_double] * 7 + [C.c_void_p]
        L.dubref_shortest.restype = None
        _dubref = L
    return _dubref


def dubref_shortest(x0, y0, th0, x1, y1, th1, rho):
    out = np.zeros(5)
    dubref_lib().dubref_shortest(float(x0), float(y0), float(th0), float(x1), float(y1), float(th1), float(rho), out.ctypes.data)
    return out[0], out[1], out[2], out[3], int(out[4])


def dubins_audit(og8, n, star, samples, headings, pts, head, vcost, parent, j, r2_rewire=0, rho=8.0, nh=64):
    """Recompute every decision behind the tree (pts, head, vcost, parent; j vertices) with dubins_ref.c's own arithmetic and count
    where they differ (see that file's head comment).  Returns a dict of the counters."""
    og8 = np.ascontiguousarray(og8, dtype=np.uint8)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    headings = np.ascontiguousarray(headings, dtype=np.int32)
    pts = np.ascontiguousarray(pts, dtype=np.int32)
    head = np.ascontiguousarray(head, dtype=np.int32)
    vcost = np.ascontiguousarray(vcost, dtype=np.float64)
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    a = _DubAudit()
    a.star, a.n, a.W, a.H = int(bool(star)), int(n), og8.shape[0], og8.shape[1]
    a.og, a.r2_rewire, a.rho, a.nh = og8.ctypes.data, int(r2_rewire), float(rho), int(nh)
    a.samples, a.headings = samples.ctypes.data, headings.ctypes.data
    a.pts, a.head, a.vcost, a.parent, a.j = pts.ctypes.data, head.ctypes.data, vcost.ctypes.data, parent.ctypes.data, int(j)
    rc = dubref_lib().dubref_audit(C.byref(a))
    if rc != 0:
        raise ValueError("dubref_audit: bad argument")
    return {k: getattr(a, k) for k, _ in _DubAudit._fields_[17:]}
