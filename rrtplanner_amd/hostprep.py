"""Host-side preparation of a planning query: everything in the reference's plan() that is
independent of the tree and therefore stays on the host as numpy (bit-exact for free):
the PCG64 sample stream, the radius thresholds, the Informed rotation matrix.
"""
import math

import numpy as np

INT64_MIN = np.iinfo(np.int64).min
_D2_CAP = (1 << 31) - 1  # larger than any squared distance of 15-bit coordinates (the expansion kernels clamp to 2^24 themselves: grids up to 2048 x 2048)


def og_nonzero(og) -> np.ndarray:
    """uint8 (W,H), 1 where the reference sees an obstacle (`og[x, y] != 0`, rrt.py:218)."""
    g = np.asarray(og)
    if g.ndim != 2:
        raise ValueError("occupancy grid must be 2-D (w, h)")
    return np.ascontiguousarray(g != 0, dtype=np.uint8)


def radius_threshold(r) -> int:
    """Smallest integer R such that, for every integer d2 >= 0, `d2 < r * r` (rrt.py:180) <=> d2 < R."""
    rr = r * r
    if isinstance(rr, (int, np.integer)):
        R = int(rr)
    else:
        rr = float(rr)
        if math.isnan(rr):
            return 0
        if math.isinf(rr):
            return _D2_CAP
        R = math.ceil(rr)
    return max(0, min(int(R), _D2_CAP))


def goal_threshold(r_goal) -> int:
    """Smallest integer G such that `sqrt(d2) < r_goal` (rrt.py:744, r2norm of an integer vector)
    <=> d2 < G, for every integer d2 >= 0.  math.sqrt is correctly rounded and monotone."""
    rg = float(r_goal)
    if math.isnan(rg) or rg <= 0.0:
        return 0
    if rg * rg >= _D2_CAP:
        return _D2_CAP
    g = max(0, int(rg * rg) - 2)
    while math.sqrt(g) < rg:
        g += 1
    while g > 0 and not (math.sqrt(g - 1) < rg):
        g -= 1
    return g


def draw_free_samples(rand_gen: np.random.Generator, free: np.ndarray, count: int) -> np.ndarray:
    """`count` consecutive `free[rand_gen.choice(F)]` draws (rrt.py:240) as one call.
    Generator.choice(F, size=k) consumes the PCG64 stream exactly like k scalar calls
    (tests/test_host_logic.py::test_rng_block_draws)."""
    F = free.shape[0]
    idx = rand_gen.choice(F, size=count)
    return np.ascontiguousarray(free[idx], dtype=np.int64)


def pack_cells(cells: np.ndarray) -> np.ndarray:
    """(F,2) integer cells -> uint32 x | y << 16 (the device's node / sample format; grids are at most 2048 wide)."""
    c = np.asarray(cells)
    return np.ascontiguousarray(c[:, 0].astype(np.uint32) | (c[:, 1].astype(np.uint32) << np.uint32(16)))


def draw_free_samples_packed(rand_gen: np.random.Generator, free_packed: np.ndarray, count: int) -> np.ndarray:
    """draw_free_samples on the packed free-cell table: the same `count` draws from the generator (one call), a 4-byte gather
    instead of the (count, 2) int64 one."""
    return free_packed[rand_gen.choice(free_packed.shape[0], size=count)]


def draw_unitball(rand_gen: np.random.Generator, count: int) -> np.ndarray:
    """`count` consecutive unit-ball points (rrt.py:579-587): r = U(0,1), theta = 2*pi*U(0,1),
    (sqrt(r)cos(theta), sqrt(r)sin(theta)).  One uniform block == 2*count scalar draws."""
    u = rand_gen.uniform(0, 1, size=2 * count).reshape(count, 2)
    r = u[:, 0]
    theta = 2 * np.pi * u[:, 1]
    s = np.sqrt(r)
    return np.ascontiguousarray(np.stack([s * np.cos(theta), s * np.sin(theta)], axis=1))


def rotation_to_world_frame(xstart: np.ndarray, xgoal: np.ndarray) -> np.ndarray:
    """Rotation used by the Informed sampler (rrt.py:601-613): SVD of outer(a1, e1) with
    a1 the unit vector start->goal; C = U diag(det U, det V) V^T.  Same numpy calls as the
    reference so the matrix is bit-identical on the same numpy build."""
    d = xgoal - xstart
    a1 = np.atleast_2d(d / np.linalg.norm(d))
    M = np.outer(a1, np.atleast_2d([1, 0]))
    try:
        U, _, V = np.linalg.svd(M)
    except np.linalg.LinAlgError:
        U, _, V = np.linalg.svd(M, full_matrices=False)
    return U @ np.diag([np.linalg.det(U), np.linalg.det(V)]) @ V.T


def ellipse_plot_params(Cm: np.ndarray, xstart: np.ndarray, xgoal: np.ndarray, c: np.ndarray):
    """Vectorised get_ellipse_for_plt (rrt.py:639-651) for an array of costs c.
    Returns (xcent, majax[], minax[], ang_deg[]); plot metadata, floating point to ~1e-12."""
    xcent = (xgoal + xstart) / 2
    d = xstart - xgoal
    d2 = float(np.dot(d.T, d))
    r1 = c / 2
    r2 = np.sqrt(np.abs(c * c - d2)) / 2
    a0, a1 = Cm[0, 0] * r1, Cm[1, 0] * r1
    b0, b1 = Cm[0, 1] * r2, Cm[1, 1] * r2
    majax = 2 * np.sqrt(a0 * a0 + a1 * a1)
    minax = 2 * np.sqrt(b0 * b0 + b1 * b1)
    ang = np.arctan2(a1, a0) * 180 / np.pi
    return xcent, majax, minax, ang


def as_int_point(x, name: str) -> np.ndarray:
    p = np.asarray(x)
    if p.shape != (2,):
        raise ValueError(f"{name} must have shape (2,), got {p.shape}")
    if not np.issubdtype(p.dtype, np.integer):
        if not np.all(p == np.floor(p)):
            raise ValueError(f"{name} must hold integer grid coordinates")
    return p.astype(np.int64)
