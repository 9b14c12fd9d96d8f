/*
 * rrt_oracle.c -- CPU restatement of rland93/rrtplanner's tree-expansion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rrtplanner_amd/ may import, link or call
 * this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / CPU comparator.
 *
 * Pinning: this restatement is checked against the real reference
 * (/root/reference/rrtplanner/rrt.py executed in the build container) through the
 * golden vectors under tests/golden/ (see tests/golden/make_golden.py and
 * tests/test_oracle_golden.py).
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/).  Plain scalar C, one thread, written for clarity first.
 *
 * Canonical tie policy ("policy A"): wherever the reference
 * takes element [0] of / iterates an np.argsort (rrt.py:154, rrt.py:317) the order
 * among equal keys is the ascending index (== np.argsort(kind="stable")).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_NEED_UNITBALL 1      /* Informed: ellipse mode starts at *i_switch, no unit-ball data given */
#define ORC_E_GOAL_UNREACHABLE -2 /* rrt.py:317-318 would index og[INT64_MIN, ...] (j < n, no line of sight) */
#define ORC_E_ARG -1

typedef struct {
    /* ---- inputs ---- */
    int32_t alg;            /* 0 RRTStandard (rrt.py:386), 1 RRTStar (rrt.py:466), 2 RRTStarInformed (rrt.py:653) */
    int32_t n;              /* attempted samples == node capacity (rrt.py:62) */
    int32_t W, H;           /* og.shape */
    const uint8_t *og;      /* (W,H) C-order, og[x*H+y] != 0 is obstacle (rrt.py:218) */
    int32_t xs[2], xg[2];
    int64_t r2_rewire;      /* smallest integer R with: d2 < r*r  <=>  d2 < R   (rrt.py:180) */
    double r_goal;          /* rrt.py:744 */
    const int32_t *samples; /* (n,2) free-space samples, sample i is used by iteration i (rrt.py:240) */
    const double *unitball; /* (n-ub_offset,2) unit-ball points (rrt.py:582-586), or NULL */
    int32_t ub_offset;      /* iteration that consumes unitball[0] */
    double C[4];            /* rotation_to_world_frame, row-major (rrt.py:601-613) */
    /* ---- outputs (caller allocated, n+1 rows) ---- */
    int32_t *pts;           /* (n+1,2) ; unfilled rows are INT32_MIN (stand-in for the reference's INT64_MIN rows) */
    double *vcost;          /* (n+1)   ; unfilled rows +inf */
    int32_t *parent;        /* (n+1)   ; -1 = none */
    int32_t *nearest_log;   /* optional (n): vnearest of iteration i */
    uint8_t *accept_log;    /* optional (n): 1 if iteration i inserted a node */
    double *cbest_log;      /* optional (n): ellipse cost c of iteration i (NaN when free-sampled) */
    int32_t *jlog;          /* optional (n): j at the top of iteration i */
    int32_t j;              /* number of tree nodes before go2goal */
    int32_t vgoal;
    int32_t found;          /* go2goal connected the goal */
    int32_t i_switch;       /* first iteration sampled from the ellipse, n if none */
    int32_t rows;           /* len(points) after go2goal: n+1 if found else n (rrt.py:320) */
    /* ---- statistics for the algorithmic-byte model ---- */
    int64_t sum_j;          /* sum over iterations of live nodes scanned */
    int64_t sum_cells_nn;   /* Bresenham cells visited nearest -> new */
    int64_t sum_near;       /* sum over accepted iterations of |within| restricted to live rows */
    int64_t sum_cells_cand; /* Bresenham cells visited by choose-parent line-of-sight tests */
    int64_t n_rewired;      /* rewire 0: times the predicate rrt.py:536 was true (always 0); rewire 1: nodes re-parented */
    /* ---- opt-in true RRT* rewire (NOT the reference's behaviour, no reference parity) ---- */
    int32_t rewire;         /* input: 0 = the reference's (vacuous) rewire scan, 1 = "correct" rewire with cost propagation */
    int32_t pad_;
    int64_t n_propagated;   /* rewire 1: descendant costs recomputed */
} orc_plan_t;

/* rrt.py:10-24  r2norm on an integer difference vector: sqrt(x0*x0 + x1*x1). */
static double orc_r2norm_i(int64_t dx, int64_t dy) { return sqrt((double)(dx * dx + dy * dy)); }

/* rrt.py:202-229  all-octant Bresenham, both endpoints inclusive, False at first obstacle.
 * cells (optional) receives the number of grid cells read. */
int orc_collisionfree(const uint8_t *og, int32_t W, int32_t H, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                      int64_t *cells) {
    (void)W;
    int32_t dx = abs(x1 - x0);
    int32_t sx = (x0 < x1) ? 1 : -1;
    int32_t dy = -abs(y1 - y0);
    int32_t sy = (y0 < y1) ? 1 : -1;
    int32_t err = dx + dy;
    int64_t c = 0;
    for (;;) {
        c++;
        if (og[(int64_t)x0 * H + y0] != 0) {
            if (cells) *cells = c;
            return 0;
        } else if (x0 == x1 && y0 == y1) {
            if (cells) *cells = c;
            return 1;
        } else {
            int32_t e2 = 2 * err;
            if (e2 >= dy) {
                err += dy;
                x0 += sx;
            }
            if (e2 <= dx) {
                err += dx;
                y0 += sy;
            }
        }
    }
}

/* Cells of the same walk, for the closed-form check in tests (no grid read). Returns count. */
int32_t orc_bresenham_cells(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t *out_xy, int32_t cap) {
    int32_t dx = abs(x1 - x0), sx = (x0 < x1) ? 1 : -1;
    int32_t dy = -abs(y1 - y0), sy = (y0 < y1) ? 1 : -1;
    int32_t err = dx + dy, c = 0;
    for (;;) {
        if (c < cap) {
            out_xy[2 * c] = x0;
            out_xy[2 * c + 1] = y0;
        }
        c++;
        if (x0 == x1 && y0 == y1) return c;
        int32_t e2 = 2 * err;
        if (e2 >= dy) {
            err += dy;
            x0 += sx;
        }
        if (e2 <= dx) {
            err += dx;
            y0 += sy;
        }
    }
}

/* rrt.py:150-155 + the [0] at rrt.py:422/503/703.  Index of the nearest live row.
 * Ordering by sqrt(float(dx)^2+float(dy)^2) equals ordering by the exact integer d2
 * (radicand < 2^53, sqrt strictly monotone on distinct integers of this size);
 * sentinel rows are never the minimum, so only rows [0,j) are scanned.
 * Ties: lowest index (policy A). */
int32_t orc_nearest(const int32_t *pts, int32_t j, int32_t x, int32_t y) {
    int64_t best = INT64_MAX;
    int32_t bi = 0;
    for (int32_t k = 0; k < j; k++) {
        int64_t dx = (int64_t)pts[2 * k] - x, dy = (int64_t)pts[2 * k + 1] - y;
        int64_t d2 = dx * dx + dy * dy;
        if (d2 < best) {
            best = d2;
            bi = k;
        }
    }
    return bi;
}

/* rrt.py:176-181  ascending indices of live rows with d2 < r*r (strict).  Sentinel rows
 * that the reference would also report carry vcost=inf and never change the result
 * (SURVEY.md 8(a) A5), so only rows [0,j) are scanned.  Returns the count. */
int32_t orc_within(const int32_t *pts, int32_t j, int32_t x, int32_t y, int64_t r2, int32_t *out) {
    int32_t m = 0;
    for (int32_t k = 0; k < j; k++) {
        int64_t dx = (int64_t)pts[2 * k] - x, dy = (int64_t)pts[2 * k + 1] - y;
        if (dx * dx + dy * dy < r2) out[m++] = k;
    }
    return m;
}

/* rrt.py:72-78  default cost: vcosts[v] + r2norm(points[v] - x). */
static double orc_cost(const orc_plan_t *p, int32_t v, int32_t x, int32_t y) {
    return p->vcost[v] + orc_r2norm_i((int64_t)p->pts[2 * v] - x, (int64_t)p->pts[2 * v + 1] - y);
}

/* rrt.py:589-599 sample_ellipse with clamp, given the unit-ball point u and cost c.
 *   xcent = (xstart + xgoal) / 2                                  rrt.py:590
 *   CL    = C @ diag(c/2, sqrt(|c*c - d2|)/2)                     rrt.py:618-625
 *   (x,y) = CL @ u + xcent                                        rrt.py:593
 *   x = int(max(0, min(W-1, x))) ; y likewise with H              rrt.py:597-598
 * np.dot(C, diag) adds exact zeros, so CL[a][b] = C[a][b]*r_b exactly.  The 2x2 @ 2
 * product is evaluated the way the build container's numpy (OpenBLAS dgemv) was
 * observed to evaluate it: fma(CL[r][0], u0, CL[r][1]*u1)  (tests/golden/make_golden.py
 * records the check). */
void orc_sample_ellipse(const double C[4], const int32_t xs[2], const int32_t xg[2], int32_t W, int32_t H, double c,
                        double u0, double u1, int32_t out[2]) {
    double xc0 = ((double)((int64_t)xs[0] + xg[0])) / 2.0, xc1 = ((double)((int64_t)xs[1] + xg[1])) / 2.0;
    int64_t ddx = (int64_t)xs[0] - xg[0], ddy = (int64_t)xs[1] - xg[1];
    double d2 = (double)(ddx * ddx + ddy * ddy);
    double r1 = c / 2.0;
    double r2 = sqrt(fabs(c * c - d2)) / 2.0;
    double CL00 = C[0] * r1, CL01 = C[1] * r2, CL10 = C[2] * r1, CL11 = C[3] * r2;
    double x = fma(CL00, u0, CL01 * u1) + xc0;
    double y = fma(CL10, u0, CL11 * u1) + xc1;
    /* Python: min(W-1, x) -> x if x < W-1 else W-1 ; max(0, v) -> v if v > 0 else 0 ; int() truncates */
    double vx = (x < (double)(W - 1)) ? x : (double)(W - 1);
    vx = (vx > 0.0) ? vx : 0.0;
    double vy = (y < (double)(H - 1)) ? y : (double)(H - 1);
    vy = (vy > 0.0) ? vy : 0.0;
    out[0] = (int32_t)vx;
    out[1] = (int32_t)vy;
}

typedef struct {
    double c;
    int32_t idx;
} orc_ci;
static int orc_ci_cmp(const void *a, const void *b) {
    const orc_ci *p = (const orc_ci *)a, *q = (const orc_ci *)b;
    if (p->c < q->c) return -1;
    if (p->c > q->c) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx);
}

/* rrt.py:284-332 go2goal (tie policy A for the argsort at :317). */
static int orc_go2goal(orc_plan_t *p) {
    const int32_t n = p->n, j = p->j;
    orc_ci *cs = (orc_ci *)malloc(sizeof(orc_ci) * (size_t)(j > 0 ? j : 1));
    for (int32_t i = 0; i < j; i++) { /* rows >= j cost +inf and sort after every live row */
        cs[i].c = orc_cost(p, i, p->xg[0], p->xg[1]);
        cs[i].idx = i;
    }
    qsort(cs, (size_t)j, sizeof(orc_ci), orc_ci_cmp);
    p->found = 0;
    for (int32_t k = 0; k < j; k++) {
        int32_t idx = cs[k].idx;
        if (orc_collisionfree(p->og, p->W, p->H, p->pts[2 * idx], p->pts[2 * idx + 1], p->xg[0], p->xg[1], NULL)) {
            int32_t vgoal = j; /* rrt.py:319 */
            /* rrt.py:320-323: arrays grow to n+1 rows, row n = goal, then row vgoal = goal */
            p->pts[2 * n] = p->xg[0];
            p->pts[2 * n + 1] = p->xg[1];
            p->vcost[n] = cs[k].c;
            p->pts[2 * vgoal] = p->xg[0];
            p->pts[2 * vgoal + 1] = p->xg[1];
            p->vcost[vgoal] = cs[k].c;
            p->parent[vgoal] = idx; /* rrt.py:325 */
            p->vgoal = vgoal;
            p->found = 1;
            p->rows = n + 1;
            break;
        }
    }
    free(cs);
    if (!p->found) {
        p->rows = n;
        if (j < n) return ORC_E_GOAL_UNREACHABLE; /* next argsort entry is a sentinel row: rrt.py:318 faults */
        p->vgoal = 0;                             /* rrt.py:330-331: norm without axis -> scalar -> argmin == 0 */
    }
    return ORC_OK;
}

/* ---- opt-in "correct" rewire (build-defined; contrast rrt.py:531-546) -------------------------------------------------
 * The reference prices a rewire with cost(vn -> xnew) = vcosts[vn] + d, which can never be below vcosts[vn]: its rewire
 * never fires.  With rewire = 1 the textbook step runs instead, defined so that a parallel machine can
 * take all decisions of one insertion at once:
 *   1. decide: for vn in vnear (ascending; the near set taken BEFORE the insertion, like rrt.py:513), with the costs as
 *      they stand right after the insertion:  c = vcosts[vnew] + r2norm(points[vn] - xnew);  vn is re-parented iff
 *      c < vcosts[vn] (strict) and collisionfree(og, points[vn], xnew)  (the call direction of rrt.py:537).
 *   2. apply: parents[vn] = vnew, vcosts[vn] = c for every such vn.
 *   3. propagate: every descendant d of a re-parented node, parents before children: vcosts[d] = vcosts[parents[d]] +
 *      r2norm(points[d] - points[parents[d]]).  (The reference propagates nothing, rrt.py:546.)
 * Costs never decrease along a root path (each is fl(parent cost + non-negative distance)), so an ancestor of vnew can
 * never pass the strict test in 1: the tree stays a tree.  Informed: the best solution vertex is the first minimum of
 * the CURRENT costs over the solution vertices in insertion order (rrt.py:627-633 evaluated on the updated costs). */
typedef struct {
    int32_t *first_child, *next_sib, *prev_sib; /* child lists: prepend on insert, O(1) unlink */
    int32_t *queue;
} orc_kids;

static void orc_kids_link(orc_kids *k, int32_t parent, int32_t child) {
    k->prev_sib[child] = -1;
    k->next_sib[child] = k->first_child[parent];
    if (k->first_child[parent] >= 0) k->prev_sib[k->first_child[parent]] = child;
    k->first_child[parent] = child;
}

static void orc_kids_unlink(orc_kids *k, int32_t parent, int32_t child) {
    const int32_t p = k->prev_sib[child], nx = k->next_sib[child];
    if (p >= 0) k->next_sib[p] = nx; else k->first_child[parent] = nx;
    if (nx >= 0) k->prev_sib[nx] = p;
}

/* The three plan() loops: rrt.py:407-447 (Standard), :487-556 (Star), :678-758 (Informed). */
int orc_plan(orc_plan_t *p) {
    const int32_t n = p->n, W = p->W, H = p->H;
    if (n < 1 || W < 1 || H < 1) return ORC_E_ARG;
    if (p->xs[0] < 0 || p->xs[0] >= W || p->xs[1] < 0 || p->xs[1] >= H) return ORC_E_ARG;
    if (p->xg[0] < 0 || p->xg[0] >= W || p->xg[1] < 0 || p->xg[1] >= H) return ORC_E_ARG;
    /* rrt.py:408-413 */
    for (int32_t k = 0; k <= n; k++) {
        p->pts[2 * k] = p->pts[2 * k + 1] = INT32_MIN;
        p->vcost[k] = INFINITY;
        p->parent[k] = -1;
    }
    p->pts[0] = p->xs[0];
    p->pts[1] = p->xs[1];
    p->vcost[0] = 0.0;
    uint8_t *sampled = (uint8_t *)calloc((size_t)W * H, 1); /* rrt.py:407 `sampled` set */
    int32_t *vnear = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    const int rw = (p->rewire == 1 && p->alg >= 1);
    orc_kids kids = {0, 0, 0, 0};
    int32_t *vsoln = NULL, *rw_v = NULL;
    double *rw_c = NULL;
    if (rw) {
        kids.first_child = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        kids.next_sib = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        kids.prev_sib = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        kids.queue = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        vsoln = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        rw_v = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        rw_c = (double *)malloc(sizeof(double) * (size_t)(n + 1));
        for (int32_t k = 0; k <= n; k++) kids.first_child[k] = kids.next_sib[k] = kids.prev_sib[k] = -1;
    }
    p->n_propagated = 0;
    int32_t i = 0, j = 1;
    /* Informed: vsoln running first-min of vcosts (rrt.py:627-633), costs never change (no rewire) */
    int32_t nsoln = 0, vbest_soln = -1;
    double cmin_soln = INFINITY;
    p->i_switch = n;
    p->sum_j = p->sum_cells_nn = p->sum_near = p->sum_cells_cand = p->n_rewired = 0;
    int status = ORC_OK;

    while (i < n) { /* rrt.py:418 / :498 / :690 */
        int32_t xn[2];
        double clog = NAN;
        if (p->alg == 2 && nsoln > 0) { /* rrt.py:697-701 */
            if (p->i_switch == n) p->i_switch = i;
            if (p->unitball == NULL || i < p->ub_offset) {
                status = ORC_NEED_UNITBALL;
                break;
            }
            double c = cmin_soln + orc_r2norm_i((int64_t)p->xg[0] - p->pts[2 * vbest_soln],
                                                (int64_t)p->xg[1] - p->pts[2 * vbest_soln + 1]); /* rrt.py:699 */
            const double *u = p->unitball + 2 * (size_t)(i - p->ub_offset);
            orc_sample_ellipse(p->C, p->xs, p->xg, W, H, c, u[0], u[1], xn);
            clog = c;
        } else { /* rrt.py:421 / :502 / :696 */
            xn[0] = p->samples[2 * i];
            xn[1] = p->samples[2 * i + 1];
        }
        if (p->cbest_log) p->cbest_log[i] = clog;
        if (p->jlog) p->jlog[i] = j;
        int32_t vnearest = orc_nearest(p->pts, j, xn[0], xn[1]); /* rrt.py:422 */
        p->sum_j += j;
        int64_t cells = 0;
        int nocoll = orc_collisionfree(p->og, W, H, p->pts[2 * vnearest], p->pts[2 * vnearest + 1], xn[0], xn[1],
                                       &cells); /* rrt.py:424 */
        p->sum_cells_nn += cells;
        int acc = nocoll && !sampled[(size_t)xn[0] * H + xn[1]] && j != n; /* rrt.py:425 */
        if (p->nearest_log) p->nearest_log[i] = vnearest;
        if (p->accept_log) p->accept_log[i] = (uint8_t)acc;
        if (acc) {
            sampled[(size_t)xn[0] * H + xn[1]] = 1; /* rrt.py:426 */
            int32_t vbest = vnearest;
            double cbest = orc_cost(p, vnearest, xn[0], xn[1]); /* rrt.py:432 / :512 */
            int32_t m = 0;
            if (p->alg >= 1) {
                m = orc_within(p->pts, j, xn[0], xn[1], p->r2_rewire, vnear); /* rrt.py:513 */
                p->sum_near += m;
                for (int32_t t = 0; t < m; t++) { /* rrt.py:515-521 choose parent */
                    int32_t vn = vnear[t];
                    double cn = orc_cost(p, vn, xn[0], xn[1]);
                    if (cn < cbest) {
                        int64_t cc = 0;
                        if (orc_collisionfree(p->og, W, H, p->pts[2 * vn], p->pts[2 * vn + 1], xn[0], xn[1], &cc)) {
                            vbest = vn;
                            cbest = cn;
                        }
                        p->sum_cells_cand += cc;
                    }
                }
            }
            int32_t vnew = j; /* rrt.py:524-529 */
            p->pts[2 * vnew] = xn[0];
            p->pts[2 * vnew + 1] = xn[1];
            p->vcost[vnew] = cbest;
            p->parent[vnew] = vbest;
            if (p->alg >= 1 && !rw) { /* rrt.py:531-546 rewire scan: predicate never true with the default cost */
                for (int32_t t = 0; t < m; t++) {
                    int32_t vn = vnear[t];
                    double cmaybe = orc_cost(p, vn, xn[0], xn[1]);
                    if (cmaybe < p->vcost[vn]) p->n_rewired++; /* would need the reference's stale-cost rewire */
                }
            }
            int32_t nrw = 0;
            if (rw) { /* the opt-in rewire, see above */
                orc_kids_link(&kids, vbest, vnew);
                for (int32_t t = 0; t < m; t++) { /* 1. decide */
                    int32_t vn = vnear[t];
                    double c = cbest + orc_r2norm_i((int64_t)p->pts[2 * vn] - xn[0], (int64_t)p->pts[2 * vn + 1] - xn[1]);
                    if (c < p->vcost[vn] && orc_collisionfree(p->og, W, H, p->pts[2 * vn], p->pts[2 * vn + 1], xn[0], xn[1], NULL)) {
                        rw_v[nrw] = vn;
                        rw_c[nrw] = c;
                        nrw++;
                    }
                }
                int32_t qh = 0, qt = 0;
                for (int32_t t = 0; t < nrw; t++) { /* 2. apply */
                    int32_t vn = rw_v[t];
                    orc_kids_unlink(&kids, p->parent[vn], vn);
                    orc_kids_link(&kids, vnew, vn);
                    p->parent[vn] = vnew;
                    p->vcost[vn] = rw_c[t];
                    kids.queue[qt++] = vn;
                }
                p->n_rewired += nrw;
                while (qh < qt) { /* 3. propagate, parents before children */
                    int32_t u = kids.queue[qh++];
                    for (int32_t c = kids.first_child[u]; c >= 0; c = kids.next_sib[c]) {
                        p->vcost[c] = p->vcost[u] + orc_r2norm_i((int64_t)p->pts[2 * c] - p->pts[2 * u], (int64_t)p->pts[2 * c + 1] - p->pts[2 * u + 1]);
                        kids.queue[qt++] = c;
                        p->n_propagated++;
                    }
                }
            }
            if (p->alg == 2) { /* rrt.py:744-745 */
                if (orc_r2norm_i((int64_t)xn[0] - p->xg[0], (int64_t)xn[1] - p->xg[1]) < p->r_goal) {
                    if (rw) vsoln[nsoln] = vnew;
                    nsoln++;
                    if (p->vcost[vnew] < cmin_soln) { /* np.argmin: first minimum */
                        cmin_soln = p->vcost[vnew];
                        vbest_soln = vnew;
                    }
                }
                if (rw && nrw > 0 && nsoln > 0) { /* costs moved: rrt.py:627-633 on the current costs, first minimum in insertion order */
                    cmin_soln = INFINITY;
                    for (int32_t t = 0; t < nsoln; t++)
                        if (p->vcost[vsoln[t]] < cmin_soln) {
                            cmin_soln = p->vcost[vsoln[t]];
                            vbest_soln = vsoln[t];
                        }
                }
            }
            j++;
        }
        i++;
    }
    p->j = j;
    free(sampled);
    free(vnear);
    if (rw) {
        free(kids.first_child);
        free(kids.next_sib);
        free(kids.prev_sib);
        free(kids.queue);
        free(vsoln);
        free(rw_v);
        free(rw_c);
    }
    if (status != ORC_OK) return status;
    return orc_go2goal(p);
}
