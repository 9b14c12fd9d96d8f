/*
 * rrt_dubins.h -- Dubins-vehicle geometry for the Dubins-RRT / Dubins-RRT* planners (BASELINE.json configs[4],
 * SURVEY.md 8(f) row 3).
 *
 * NO REFERENCE PARITY: rland93/rrtplanner only advertises "Dubins Vehicle RRT / RRT(star) Planner" (README.md:12,18-19);
 * no such module is in its tree.  The semantics are defined by this build:
 *
 *   state     (x, y, h): integer grid cell and one of `nh` discrete headings, theta = 2 pi h / nh
 *   edge      the shortest of the six Dubins words LSL, LSR, RSL, RSR, RLR, LRL of turning radius rho (cells) from the
 *             parent's pose to the child's pose; ties between words go to the first in that order
 *   cost      vcost[parent] + rho * (t + p + q), the arc length of that path
 *   collision the path is sampled every DUB_DS cells of arc length from its start (k * DUB_DS, k = 0, 1, ...) and at its
 *             end pose; a sample occupies the cell round-half-up of its coordinates; the path is free iff every such cell
 *             lies inside the grid and is free.  "cells read" = samples up to and including the first blocked one.
 *
 * Plain C, usable from HIP device code and from gcc.  The CPU oracle (oracle/dubins_oracle.c) and the HIP kernel include
 * this one header on purpose: the planner's decisions compare path lengths, so "bit-exact tree" needs bit-identical
 * arithmetic, and libm's and the device's sin / atan2 / acos are not that.  Everything here is built from + - * /, sqrt,
 * floor and EXPLICIT fused multiply-adds (DUB_FMA: IEEE fma, one rounding -- v_fma_f64 on the device, fma() on the host) in a
 * fixed order; both sides compile with -ffp-contract=off, so nothing else is fused and nothing is reassociated, and gcc on
 * the host and hipcc on gfx950 produce the same bits.  (Round 4: until then every a * b + c was two instructions -- the
 * polynomials, the reductions and the rotations of a word evaluation were 8 400 separate f64 multiplies and adds in the kernel.)  What the oracle adds is its own sequential loop; the FORMULAS are
 * checked separately against numpy / libm (tests/test_dubins.py: every reported word, integrated forward with numpy,
 * ends in the goal pose; lengths agree with an independent implementation to 1e-9).
 */
#ifndef RRT_DUBINS_H
#define RRT_DUBINS_H

#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define RRT_DUB_FN __host__ __device__ static inline
#define RRT_DUB_BIG_FN __host__ __device__ static __attribute__((noinline)) /* one copy in the kernel: a word evaluation is ~1 400 instructions */
#else
#define RRT_DUB_FN static inline
#define RRT_DUB_BIG_FN static inline
#endif

#define DUB_FMA(a, b, c) __builtin_fma((a), (b), (c)) /* a * b + c with ONE rounding, on both sides */
/* a * b + K for a CONSTANT K (the Horner steps).  The same fma; on the device written out as the three-operand instruction with
 * K in scalar registers: left to itself the compiler picks the two-operand v_fmac_f64, whose addend is its destination, and copies
 * every coefficient into vector registers first -- two moves per step, a quarter of a word evaluation's vector instructions. */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double dub_fma_k(double a, double b, double k) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#define DUB_FMA_K(a, b, k) dub_fma_k((a), (b), (k))
#else
#define DUB_FMA_K(a, b, k) __builtin_fma((a), (b), (k))
#endif

#define DUB_PI 3.141592653589793
#define DUB_TWOPI 6.283185307179586
#define DUB_DS 0.5 /* arc-length step of the collision sweep, cells */

enum { DUB_LSL = 0, DUB_LSR = 1, DUB_RSL = 2, DUB_RSR = 3, DUB_RLR = 4, DUB_LRL = 5, DUB_NONE = 6 };

typedef struct {
    double t, p, q; /* the three segment lengths in units of rho (angles for arcs) */
    double len;     /* rho * (t + p + q), cells; +inf when no word applies (cannot happen for rho > 0) */
    int32_t word;
} dub_path_t;

/* ---- elementary functions with a fixed operation order ------------------------------------------------------------ */
/* a mod 2 pi in [0, 2 pi).  The quotient comes from a multiplication by 1 / (2 pi) (a word evaluation makes 21 of these: as
 * divisions they were a seventh of its instructions); the product may land on the other side of an integer by an ulp, which the
 * two selects behind it put right. */
#define DUB_INV_TWOPI 0.15915494309189535
RRT_DUB_FN double dub_mod2pi(double a) {
    double r = DUB_FMA(-DUB_TWOPI, floor(a * DUB_INV_TWOPI), a);
    r = r < 0.0 ? r + DUB_TWOPI : r;
    return r >= DUB_TWOPI ? r - DUB_TWOPI : r;
}

/* sin and cos of a (|a| up to a few hundred): quadrant by Cody-Waite reduction with a two-part pi/2, then the Taylor
 * polynomials on [-pi/4, pi/4] (truncation below 1e-17). */
RRT_DUB_FN void dub_sincos(double a, double *s, double *c) {
    const double k = floor(DUB_FMA(a, 0.6366197723675814, 0.5)); /* 2 / pi */
    const double r = DUB_FMA(-k, 6.077100506506192e-11, DUB_FMA(-k, 1.5707963267341256, a)); /* pi/2 = hi + lo, hi has 33 significant bits */
    const double z = r * r;
    double ps = -1.0 / 1307674368000.0, pc = 1.0 / 20922789888000.0; /* Horner, one fused step per coefficient */
    ps = DUB_FMA_K(z, ps, 1.0 / 6227020800.0);
    ps = DUB_FMA_K(z, ps, -1.0 / 39916800.0);
    ps = DUB_FMA_K(z, ps, 1.0 / 362880.0);
    ps = DUB_FMA_K(z, ps, -1.0 / 5040.0);
    ps = DUB_FMA_K(z, ps, 1.0 / 120.0);
    ps = DUB_FMA_K(z, ps, -1.0 / 6.0);
    ps = DUB_FMA_K(z, ps, 1.0);
    pc = DUB_FMA_K(z, pc, -1.0 / 87178291200.0);
    pc = DUB_FMA_K(z, pc, 1.0 / 479001600.0);
    pc = DUB_FMA_K(z, pc, -1.0 / 3628800.0);
    pc = DUB_FMA_K(z, pc, 1.0 / 40320.0);
    pc = DUB_FMA_K(z, pc, -1.0 / 720.0);
    pc = DUB_FMA_K(z, pc, 1.0 / 24.0);
    pc = DUB_FMA_K(z, pc, -0.5);
    const double sp = r * ps;
    const double cp = DUB_FMA_K(z, pc, 1.0);
    const double kq = DUB_FMA(-4.0, floor(k * 0.25), k); /* k mod 4 in {0, 1, 2, 3} */
    if (kq == 0.0) {
        *s = sp;
        *c = cp;
    } else if (kq == 1.0) {
        *s = cp;
        *c = -sp;
    } else if (kq == 2.0) {
        *s = -sp;
        *c = -cp;
    } else {
        *s = -cp;
        *c = sp;
    }
}

/* atan(num / den) for 0 <= num <= den, den > 0, with ONE division: the quotient's sixteenth of pi is found by comparing num with
 * den * tan((2k - 1) pi / 32), then atan(num / den) = k pi / 16 + atan(w) with w = (num - t den) / (den + t num), t = tan(k pi / 16),
 * |w| <= tan(pi / 32) = 0.0985, and the alternating series in w^2 needs 8 terms (truncation below 5e-18 relative).  (Round 2:
 * quotient, one reduction at tan(pi / 8), 21 terms -- two divisions and 42 dependent operations per arctangent.) */
RRT_DUB_FN double dub_atan_ratio(double num, double den) {
    double t = 0.0, base = 0.0;
    if (num > den * 0.8206787908286602) { /* tan(7 pi / 32) */
        t = 1.0;
        base = 0.7853981633974483;
    } else if (num > den * 0.5345111359507916) { /* tan(5 pi / 32) */
        t = 0.6681786379192989; /* tan(3 pi / 16) */
        base = 0.5890486225480862;
    } else if (num > den * 0.3033466836073424) { /* tan(3 pi / 32) */
        t = 0.41421356237309503; /* tan(pi / 8) */
        base = 0.39269908169872414;
    } else if (num > den * 0.09849140335716425) { /* tan(pi / 32) */
        t = 0.198912367379658; /* tan(pi / 16) */
        base = 0.19634954084936207;
    }
    const double w = DUB_FMA(-t, den, num) / DUB_FMA(t, num, den);
    const double u = -(w * w);
    double acc = 1.0 / 15.0;
    acc = DUB_FMA_K(u, acc, 1.0 / 13.0);
    acc = DUB_FMA_K(u, acc, 1.0 / 11.0);
    acc = DUB_FMA_K(u, acc, 1.0 / 9.0);
    acc = DUB_FMA_K(u, acc, 1.0 / 7.0);
    acc = DUB_FMA_K(u, acc, 1.0 / 5.0);
    acc = DUB_FMA_K(u, acc, 1.0 / 3.0);
    acc = DUB_FMA_K(u, acc, 1.0);
    return DUB_FMA(w, acc, base);
}

/* atan2(y, x) in (-pi, pi]; atan2(0, 0) = 0 */
RRT_DUB_FN double dub_atan2(double y, double x) {
    const double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    if (ax == 0.0 && ay == 0.0) return 0.0;
    /* ONE series whichever octant: the smaller magnitude over the larger */
    const int steep = !(ay <= ax);
    const double t = dub_atan_ratio(steep ? ax : ay, steep ? ay : ax);
    double a = steep ? 1.5707963267948966 - t : t;
    if (x < 0.0) a = DUB_PI - a;
    return y < 0.0 ? -a : a;
}

/* acos(x) for |x| <= 1 */
RRT_DUB_FN double dub_acos(double x) { return dub_atan2(sqrt((1.0 - x) * (1.0 + x)), x); }

/* heading index -> angle */
RRT_DUB_FN double dub_heading(int32_t h, int32_t nh) { return DUB_TWOPI * (double)h / (double)nh; }

/* ---- the shortest Dubins word from pose 0 to pose 1 ------------------------------------------------------------------ */
RRT_DUB_FN void dub_take(dub_path_t *best, int32_t word, double t, double p, double q) {
    const double sum = t + p + q;
    if (sum < best->len) { /* strict: an equal sum keeps the earlier word */
        best->len = sum;
        best->t = t;
        best->p = p;
        best->q = q;
        best->word = word;
    }
}

/* (s0, c0), (s1, c1): sine and cosine of the two headings (dub_sincos(th0), dub_sincos(th1): a caller with discrete headings keeps
 * them in a table).  The sines and cosines of alpha = th0 - theta and beta = th1 - theta come from those by the rotation with
 * (cos theta, sin theta) = (dx, dy) / D, and cos(alpha - beta) from them: no series (round 2 ran three per word). */
RRT_DUB_BIG_FN dub_path_t dub_shortest_sc(double x0, double y0, double th0, double s0, double c0, double x1, double y1, double th1, double s1, double c1,
                                      double rho) {
    const double dx = x1 - x0, dy = y1 - y0;
    const double D = sqrt(DUB_FMA(dx, dx, dy * dy));
    const double d = D / rho;
    const double theta = dub_mod2pi(dub_atan2(dy, dx));
    const double alpha = dub_mod2pi(th0 - theta), beta = dub_mod2pi(th1 - theta);
    const double ct = D > 0.0 ? dx / D : 1.0, st = D > 0.0 ? dy / D : 0.0; /* (atan2(0, 0) = 0) */
    const double sa = DUB_FMA(s0, ct, -(c0 * st)), ca = DUB_FMA(c0, ct, s0 * st);
    const double sb = DUB_FMA(s1, ct, -(c1 * st)), cb = DUB_FMA(c1, ct, s1 * st);
    const double cab = DUB_FMA(ca, cb, sa * sb);
    const double d2 = 2.0 * d; /* (exact) */
    const double dsq = d * d;
    dub_path_t best;
    best.t = best.p = best.q = 0.0;
    best.len = HUGE_VAL;
    best.word = DUB_NONE;
    /* two arctangents serve four words: LSL and LRL turn about the same pair of circles, RSR and RLR too.  dub_atan2 is odd in
     * y except at y = 0 (where it returns its value for +0 whatever the sign), so LRL's angle is LSL's negated or, for equal
     * cosines, LSL's itself: the very values the four separate calls returned */
    const double at_l = dub_atan2(cb - ca, d + sa - sb), at_r = dub_atan2(ca - cb, d - sa + sb);
    { /* LSL */
        const double psq = DUB_FMA(d2, sa - sb, DUB_FMA(-2.0, cab, 2.0 + dsq));
        if (psq >= 0.0) {
            const double tmp = at_l;
            dub_take(&best, DUB_LSL, dub_mod2pi(tmp - alpha), sqrt(psq), dub_mod2pi(beta - tmp));
        }
    }
    { /* LSR */
        const double psq = DUB_FMA(d2, sa + sb, DUB_FMA(2.0, cab, -2.0 + dsq));
        if (psq >= 0.0) {
            const double p = sqrt(psq);
            /* atan2(A, B) - atan2(-2, p) as ONE arctangent: the angle of (B, A) turned back by that of (p, -2), i.e. of
             * (B p - 2 A, A p + 2 B); the two differ by a multiple of 2 pi at most, which the reductions below take out */
            const double A = -ca - cb, B = d + sa + sb;
            const double tmp = dub_atan2(DUB_FMA(A, p, 2.0 * B), DUB_FMA(B, p, -(2.0 * A)));
            dub_take(&best, DUB_LSR, dub_mod2pi(tmp - alpha), p, dub_mod2pi(tmp - dub_mod2pi(beta)));
        }
    }
    { /* RSL */
        const double psq = DUB_FMA(-d2, sa + sb, DUB_FMA(2.0, cab, -2.0 + dsq));
        if (psq >= 0.0) {
            const double p = sqrt(psq);
            /* atan2(A, B) - atan2(2, p): the angle of (B p + 2 A, A p - 2 B) */
            const double A = ca + cb, B = d - sa - sb;
            const double tmp = dub_atan2(DUB_FMA(A, p, -(2.0 * B)), DUB_FMA(B, p, 2.0 * A));
            dub_take(&best, DUB_RSL, dub_mod2pi(alpha - tmp), p, dub_mod2pi(beta - tmp));
        }
    }
    { /* RSR */
        const double psq = DUB_FMA(d2, sb - sa, DUB_FMA(-2.0, cab, 2.0 + dsq));
        if (psq >= 0.0) {
            const double tmp = at_r;
            dub_take(&best, DUB_RSR, dub_mod2pi(alpha - tmp), sqrt(psq), dub_mod2pi(tmp - beta));
        }
    }
    { /* RLR */
        const double tmp = DUB_FMA(d2, sa - sb, DUB_FMA(2.0, cab, 6.0 - dsq)) * 0.125;
        if (tmp <= 1.0 && tmp >= -1.0) {
            const double phi = at_r;
            const double p = dub_mod2pi(DUB_TWOPI - dub_acos(tmp));
            const double t = dub_mod2pi(alpha - phi + dub_mod2pi(p / 2.0));
            dub_take(&best, DUB_RLR, t, p, dub_mod2pi(alpha - beta - t + dub_mod2pi(p)));
        }
    }
    { /* LRL */
        const double tmp = DUB_FMA(d2, sb - sa, DUB_FMA(2.0, cab, 6.0 - dsq)) * 0.125;
        if (tmp <= 1.0 && tmp >= -1.0) {
            const double phi = (ca - cb == 0.0) ? at_l : -at_l; /* = dub_atan2(ca - cb, d + sa - sb) */
            const double p = dub_mod2pi(DUB_TWOPI - dub_acos(tmp));
            const double t = dub_mod2pi(-alpha - phi + p / 2.0);
            dub_take(&best, DUB_LRL, t, p, dub_mod2pi(dub_mod2pi(beta) - alpha - t + dub_mod2pi(p)));
        }
    }
    best.len = best.len * rho; /* (t + p + q) * rho; stays +inf when no word applied */
    return best;
}

RRT_DUB_FN dub_path_t dub_shortest(double x0, double y0, double th0, double x1, double y1, double th1, double rho) {
    double s0, c0, s1, c1;
    dub_sincos(th0, &s0, &c0);
    dub_sincos(th1, &s1, &c1);
    return dub_shortest_sc(x0, y0, th0, s0, c0, x1, y1, th1, s1, c1, rho);
}

/* ---- points of a path ---------------------------------------------------------------------------------------------- */
/* segment kinds of the words: +1 left arc, -1 right arc, 0 straight */
RRT_DUB_FN int32_t dub_seg_kind(int32_t word, int32_t k) {
    /* (kind + 1) in two bits per entry, entry 3 * word + k:  LSL 1,0,1  LSR 1,0,-1  RSL -1,0,1  RSR -1,0,-1  RLR -1,1,-1  LRL 1,-1,1 */
    return (int32_t)((0x8881241a6ull >> (2 * (3 * word + k))) & 3ull) - 1;
}

/* advance the normalised pose (x, y, th) (unit turning radius) by tau along a segment of the given kind */
RRT_DUB_FN void dub_advance(double x, double y, double th, int32_t kind, double tau, double *ox, double *oy, double *oth) {
    double s0, c0;
    dub_sincos(th, &s0, &c0);
    if (kind == 0) {
        *ox = DUB_FMA(c0, tau, x);
        *oy = DUB_FMA(s0, tau, y);
        *oth = th;
    } else if (kind > 0) {
        double s1, c1;
        dub_sincos(th + tau, &s1, &c1);
        *ox = x + (s1 - s0);
        *oy = y - (c1 - c0);
        *oth = th + tau;
    } else {
        double s1, c1;
        dub_sincos(th - tau, &s1, &c1);
        *ox = x - (s1 - s0);
        *oy = y + (c1 - c0);
        *oth = th - tau;
    }
}

/* the sweep of one path, prepared once per path: the poses at the two junctions (normalised, relative to the start) */
typedef struct {
    double x0, y0, th0, rho, inv_rho;
    double t, p, q;
    double x1, y1, th1; /* pose after the first segment */
    double x2, y2, th2; /* pose after the second segment */
    int32_t k0, k1, k2; /* segment kinds */
    int32_t nsamples;   /* samples at k * DUB_DS, k = 0 .. nsamples - 1 (the end pose is one more, tested as the goal cell) */
    double sn0, cs0, sn1, cs1, sn2, cs2; /* sin / cos of the heading at the start of each segment (what dub_advance computes first) */
} dub_sweep_t;

/* dub_advance with the sine and cosine of th given, position only.  Both arc directions go through ONE dub_sincos call: adding
 * sg * tau and sg * (difference) with sg = +-1 is exactly the subtraction dub_advance writes for a right arc, so the values are
 * the same bit for bit, and a wavefront whose samples lie on different segments evaluates one series, not one per branch. */
RRT_DUB_FN void dub_advance_pre(double x, double y, double th, double s0, double c0, int32_t kind, double tau, double *ox, double *oy) {
    if (kind == 0) {
        *ox = DUB_FMA(c0, tau, x);
        *oy = DUB_FMA(s0, tau, y);
        return;
    }
    const double sg = kind > 0 ? 1.0 : -1.0;
    double s1, c1;
    dub_sincos(DUB_FMA(sg, tau, th), &s1, &c1); /* (sg = +-1: the product is exact, the sum th +- tau as before) */
    *ox = DUB_FMA(sg, s1 - s0, x);
    *oy = DUB_FMA(-sg, c1 - c0, y);
}

/* dub_advance with sin / cos of th given and sin / cos of the new heading returned */
RRT_DUB_FN void dub_advance_sc(double x, double y, double th, double s0, double c0, int32_t kind, double tau, double *ox, double *oy, double *oth,
                               double *os, double *oc) {
    if (kind == 0) {
        *ox = DUB_FMA(c0, tau, x);
        *oy = DUB_FMA(s0, tau, y);
        *oth = th;
        *os = s0;
        *oc = c0;
    } else if (kind > 0) {
        double s1, c1;
        dub_sincos(th + tau, &s1, &c1);
        *ox = x + (s1 - s0);
        *oy = y - (c1 - c0);
        *oth = th + tau;
        *os = s1;
        *oc = c1;
    } else {
        double s1, c1;
        dub_sincos(th - tau, &s1, &c1);
        *ox = x - (s1 - s0);
        *oy = y + (c1 - c0);
        *oth = th - tau;
        *os = s1;
        *oc = c1;
    }
}

/* the set-up with the sine and cosine of th0 given (a caller that keeps a table of the discrete headings' values: the same numbers
 * dub_sincos(th0) returns) */
RRT_DUB_FN dub_sweep_t dub_sweep_setup_sc(double x0, double y0, double th0, double sn0, double cs0, const dub_path_t *path, double rho) {
    dub_sweep_t s;
    s.x0 = x0;
    s.y0 = y0;
    s.th0 = th0;
    s.rho = rho;
    s.inv_rho = 1.0 / rho;
    s.t = path->t;
    s.p = path->p;
    s.q = path->q;
    s.k0 = dub_seg_kind(path->word, 0);
    s.k1 = dub_seg_kind(path->word, 1);
    s.k2 = dub_seg_kind(path->word, 2);
    /* the two junction poses as dub_advance gives them; the sine and cosine of a junction's heading are the ones the advance
     * that reaches it has just evaluated (same argument), so three series serve the whole set-up */
    s.sn0 = sn0;
    s.cs0 = cs0;
    dub_advance_sc(0.0, 0.0, th0, s.sn0, s.cs0, s.k0, s.t, &s.x1, &s.y1, &s.th1, &s.sn1, &s.cs1);
    dub_advance_sc(s.x1, s.y1, s.th1, s.sn1, s.cs1, s.k1, s.p, &s.x2, &s.y2, &s.th2, &s.sn2, &s.cs2);
    s.nsamples = (int32_t)floor(path->len / DUB_DS) + 1;
    return s;
}

RRT_DUB_FN dub_sweep_t dub_sweep_setup(double x0, double y0, double th0, const dub_path_t *path, double rho) {
    double sn0, cs0;
    dub_sincos(th0, &sn0, &cs0);
    return dub_sweep_setup_sc(x0, y0, th0, sn0, cs0, path, rho);
}

/* grid cell of sample k (arc length k * DUB_DS from the start) */
RRT_DUB_FN void dub_sweep_cell(const dub_sweep_t *s, int32_t k, int32_t *cx, int32_t *cy) {
    const double tau = ((double)k * DUB_DS) * s->inv_rho; /* normalised arc length (one division per sweep, not per sample) */
    /* the segment the sample lies on is selected first, then ONE advance (same operands as one call per branch) */
    double bx = s->x2, by = s->y2, bth = s->th2, bs = s->sn2, bc = s->cs2, dt = tau - (s->t + s->p), x, y;
    int32_t kind = s->k2;
    if (tau < s->t) {
        bx = 0.0;
        by = 0.0;
        bth = s->th0;
        bs = s->sn0;
        bc = s->cs0;
        kind = s->k0;
        dt = tau;
    } else if (tau < s->t + s->p) {
        bx = s->x1;
        by = s->y1;
        bth = s->th1;
        bs = s->sn1;
        bc = s->cs1;
        kind = s->k1;
        dt = tau - s->t;
    }
    dub_advance_pre(bx, by, bth, bs, bc, kind, dt, &x, &y);
    *cx = (int32_t)floor(DUB_FMA(x, s->rho, s->x0) + 0.5);
    *cy = (int32_t)floor(DUB_FMA(y, s->rho, s->y0) + 0.5);
}

#endif /* RRT_DUBINS_H */
