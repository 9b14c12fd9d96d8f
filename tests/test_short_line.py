"""The quotient of the device's short-segment line of sight (rrtplanner_amd/csrc/rrt_device.h: short_line_cell).

The kernels evaluate cell k of a segment shorter than 64 steps as base + k * stride_major + m_k * stride_minor with
m_k = (int)(float(2 minor k + major) * rcp + 0.5 rcp) and rcp ~ 1 / (2 major) from v_rcp_f32 -- no integer fix-up.  This checks, for
every such segment, every k and a reciprocal that is off by several ulp, fused or not, that the value equals the closed form of
include/rrt_line.h (floor((2 minor k + major) / (2 major)), the walk of the reference's collisionfree, rrt.py:202-229), and that the
closed form itself is the reference's walk (against the oracle's serial Bresenham)."""
import numpy as np
import pytest

import oracle


def _exact(major, minor, k):
    den = 2 * major
    return (2 * minor * k + major) // den if den else np.zeros_like(k)


@pytest.mark.parametrize("ulp", [-4, -1, 0, 1, 4])
def test_biased_float_quotient_is_exact_for_short_segments(ulp):
    bad = 0
    for major in range(0, 64):
        d = np.float32(2 * major if major else 1)
        rcp = np.float32((np.float32(1.0) / d) * (1.0 + ulp * 2.0**-23))
        half = np.float32(np.float32(0.5) * rcp)
        k = np.arange(0, major + 1)
        for minor in range(0, major + 1):
            num = 2 * minor * k + major
            exact = _exact(major, minor, k)
            unfused = (num.astype(np.float32) * rcp + half).astype(np.float32)
            fused = (num.astype(np.float64) * np.float64(rcp) + np.float64(half)).astype(np.float32)
            bad += int((unfused.astype(np.int32) != exact).sum()) + int((fused.astype(np.int32) != exact).sum())
    assert bad == 0


def test_closed_form_cells_are_the_serial_walk():
    """base + k * stride_major + m_k * stride_minor visits the cells the serial walk reads, in order (all short segments from the centre
    of a free 127 x 127 grid: the oracle's cell count is the whole walk, and a single obstacle on the predicted cell k stops it there)."""
    W = H = 127
    cx = cy = 63
    rng = np.random.default_rng(3)
    for _ in range(300):
        dx, dy = int(rng.integers(-63, 64)), int(rng.integers(-63, 64))
        x1, y1 = cx + dx, cy + dy
        adx, ady = abs(dx), abs(dy)
        xm = adx >= ady
        major, minor = (adx, ady) if xm else (ady, adx)
        sx, sy = (1 if cx < x1 else -1), (1 if cy < y1 else -1)
        k = int(rng.integers(0, major + 1))
        m = int(_exact(major, minor, np.array([k]))[0])
        px, py = (cx + sx * k, cy + sy * m) if xm else (cx + sx * m, cy + sy * k)
        g = np.zeros((W, H), np.uint8)
        ok, cells = oracle.collisionfree(g, np.array([cx, cy]), np.array([x1, y1]))
        assert ok and cells == major + 1
        g[px, py] = 1
        ok, cells = oracle.collisionfree(g, np.array([cx, cy]), np.array([x1, y1]))
        assert (not ok) and cells == k + 1
