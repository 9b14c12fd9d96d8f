/*
 * rrt_line.h -- closed form of the reference's line-of-sight walk.
 *
 * The reference (rrtplanner/rrt.py:202-229) walks an all-octant integer Bresenham line
 * from a to b with an error accumulator.  Cell k of that walk (k = 0 .. L,
 * L = max(|dx|,|dy|)) has the closed form
 *
 *     major axis :  a_major + s_major * k
 *     minor axis :  a_minor + s_minor * floor((2*minor*k + major) / (2*major))
 *
 * with major = max(|dx|,|dy|), minor = min(|dx|,|dy|) (x is the major axis when
 * |dx| >= |dy|), s = +1 if a < b else -1 per axis (rrt.py:207-215; the sign for an equal
 * coordinate is -1 but its axis never moves).  Derivation: with m minor steps after k
 * iterations err = major*(m+1) - minor*(k+1) (x-major case), the minor step fires iff
 * 2*err <= major  <=>  (2m+1)*major <= 2*minor*(k+1), i.e. m_k = floor((2*minor*k + major) /
 * (2*major)); the major axis steps every iteration.  The walk is direction sensitive
 * (round-half-up measured from a), exactly like the reference.
 *
 * Having cell k in closed form lets the 64 lanes of a wavefront test 64 cells of one
 * segment at once (ballot = any hit / first hit) instead of chasing the error
 * accumulator serially.  tests/test_line_closed_form.py checks this header against the
 * oracle's literal walk for every ordered pair of a grid and for long random segments.
 *
 * Plain C, usable from HIP device code and from gcc.
 */
#ifndef RRT_LINE_H
#define RRT_LINE_H

#include <stdint.h>

#ifdef __HIPCC__
#define RRT_LINE_FN __host__ __device__ static inline
#else
#define RRT_LINE_FN static inline
#endif

typedef struct {
    int32_t x0, y0;  /* start cell a */
    int32_t sx, sy;  /* per-axis step */
    int32_t major;   /* L = number of steps; the walk has L+1 cells */
    int32_t minor;
    int32_t xmajor;  /* 1: x advances every step */
    float rcp_den;   /* 1 / (2*major) (approximate, fixed up in rrt_line_cell) */
} rrt_line_t;

RRT_LINE_FN rrt_line_t rrt_line_setup(int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
    rrt_line_t l;
    int32_t dx = x1 - x0, dy = y1 - y0;
    int32_t adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    l.x0 = x0;
    l.y0 = y0;
    l.sx = (x0 < x1) ? 1 : -1;
    l.sy = (y0 < y1) ? 1 : -1;
    l.xmajor = adx >= ady;
    l.major = l.xmajor ? adx : ady;
    l.minor = l.xmajor ? ady : adx;
    l.rcp_den = 1.0f / (float)(l.major > 0 ? 2 * l.major : 1);
    return l;
}

/* Cell k (0 <= k <= major) of the walk.  Requires 2*minor*k + major < 2^24 (grids up to
 * 2048 x 2048): the float estimate of the quotient is then within 1 and the integer
 * remainder fixes it exactly. */
RRT_LINE_FN void rrt_line_cell(const rrt_line_t *l, int32_t k, int32_t *x, int32_t *y) {
    int32_t den = 2 * l->major;
    int32_t num = 2 * l->minor * k + l->major;
    int32_t m = (int32_t)((float)num * l->rcp_den);
    int32_t r = num - m * den;
    m += (r >= den) ? 1 : 0;
    m -= (r < 0) ? 1 : 0;
    if (den == 0) m = 0;
    if (l->xmajor) {
        *x = l->x0 + l->sx * k;
        *y = l->y0 + l->sy * m;
    } else {
        *x = l->x0 + l->sx * m;
        *y = l->y0 + l->sy * k;
    }
}

/* The same cell without the 2^24 limit (grids up to 32767 x 32767: the slow general path of the host-driven planners): the
 * quotient by 64-bit integer division. */
RRT_LINE_FN void rrt_line_cell_wide(const rrt_line_t *l, int32_t k, int32_t *x, int32_t *y) {
    const int64_t den = 2 * (int64_t)l->major;
    const int64_t num = 2 * (int64_t)l->minor * (int64_t)k + (int64_t)l->major;
    const int32_t m = den == 0 ? 0 : (int32_t)(num / den);
    if (l->xmajor) {
        *x = l->x0 + l->sx * k;
        *y = l->y0 + l->sy * m;
    } else {
        *x = l->x0 + l->sx * m;
        *y = l->y0 + l->sy * k;
    }
}

#endif /* RRT_LINE_H */
