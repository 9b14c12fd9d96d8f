/*
 * dubins_ref.c -- an INDEPENDENT check of the Dubins planners' decisions (BASELINE.json configs[4]).
 *
 * TEST INFRASTRUCTURE ONLY: nothing under rrtplanner_amd/ may import, link or call this file.
 *
 * Why it exists: the CPU oracle (dubins_oracle.c) and the HIP kernel share include/rrt_dubins.h on purpose -- bit-identical
 * arithmetic is the only way to a bit-identical tree -- so "HIP == oracle" cannot see an error in that header's geometry.  This
 * file does NOT include it.  It is the textbook construction of the six Dubins words (Shkel & Lumelsky's closed forms: LSL, LSR,
 * RSL, RSR, RLR, LRL) written against libm (sin, cos, atan2, acos, fmod), with its own forward integration of a word and its own
 * sweep, and an AUDIT of a finished tree: replaying the sample stream against the tree as the planner built it, it recomputes
 * every decision the planner took with this file's arithmetic and counts where they differ.
 *
 * The planner compares path lengths that agree with libm's only to ~1e-12, so the audit cannot demand equal bits.  It demands:
 *   accept / reject (rrt.py:424-425 with the word's sweep for collisionfree)  equal, except where this file's sweep passes within
 *       AMBIG of a cell boundary (a sample whose cell depends on the last bits), counted as ambiguous
 *   the chosen parent (rrt.py:511-521)  is the first minimum, in index order, of this file's costs over the vertices within
 *       r_rewire that this file's sweep sees -- or its cost is within TOL of that minimum (two candidates closer than the
 *       arithmetic can separate), counted; a chosen parent this file's sweep calls blocked is an error unless ambiguous
 *   the stored cost  equals this file's cost through the chosen parent to TOL
 * There is no reference code for these planners (README.md:12,18-19 only advertises them): this is a second opinion, not parity.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REF_PI 3.14159265358979323846
#define REF_DS 0.5     /* arc-length step of the sweep, cells */
#define REF_AMBIG 1e-7 /* a sweep sample this close to a cell boundary makes the path's decision ambiguous */
#define REF_TOL 1e-9   /* relative + absolute tolerance on costs */

static double mod2pi(double a) {
    double r = fmod(a, 2.0 * REF_PI);
    if (r < 0.0) r += 2.0 * REF_PI;
    return r;
}

typedef struct ref_path_s {
    double seg[3]; /* segment lengths in units of rho */
    int kind[3];   /* +1 left, -1 right, 0 straight */
    double len;    /* rho * sum, cells; HUGE_VAL when no word applies */
    int word;      /* 0 LSL 1 LSR 2 RSL 3 RSR 4 RLR 5 LRL, -1 none */
} ref_path;

static const int KIND[6][3] = {{1, 0, 1}, {1, 0, -1}, {-1, 0, 1}, {-1, 0, -1}, {-1, 1, -1}, {1, -1, 1}};

/* the shortest word from (x0, y0, th0) to (x1, y1, th1), turning radius rho; ties: the first word in the order above.
 * ties / ntie (optional): every OTHER word whose length is within REF_TOL of the shortest -- mirror-symmetric pose pairs have two
 * words of equal length (LSL / RSR, LSR / RSL) whose computed sums differ in the last bits, so which of them an arithmetic calls
 * "the shortest" is its own business, and they are different curves through the grid */
static ref_path ref_shortest_ties(double x0, double y0, double th0, double x1, double y1, double th1, double rho, struct ref_path_s *ties, int *ntie);
static struct ref_path_s ref_shortest(double x0, double y0, double th0, double x1, double y1, double th1, double rho) {
    return ref_shortest_ties(x0, y0, th0, x1, y1, th1, rho, 0, 0);
}
static ref_path ref_shortest_ties(double x0, double y0, double th0, double x1, double y1, double th1, double rho, struct ref_path_s *ties, int *ntie) {
    const double dx = x1 - x0, dy = y1 - y0;
    const double d = sqrt(dx * dx + dy * dy) / rho;
    const double theta = (dx == 0.0 && dy == 0.0) ? 0.0 : mod2pi(atan2(dy, dx));
    const double a = mod2pi(th0 - theta), b = mod2pi(th1 - theta);
    const double sa = sin(a), sb = sin(b), ca = cos(a), cb = cos(b), cab = cos(a - b);
    double t[6], p[6], q[6];
    int ok[6] = {0, 0, 0, 0, 0, 0};
    double tmp;
    tmp = 2.0 + d * d - 2.0 * cab + 2.0 * d * (sa - sb); /* LSL */
    if (tmp >= 0.0) {
        const double phi = atan2(cb - ca, d + sa - sb);
        t[0] = mod2pi(phi - a), p[0] = sqrt(tmp), q[0] = mod2pi(b - phi), ok[0] = 1;
    }
    tmp = -2.0 + d * d + 2.0 * cab + 2.0 * d * (sa + sb); /* LSR */
    if (tmp >= 0.0) {
        p[1] = sqrt(tmp);
        const double phi = atan2(-ca - cb, d + sa + sb) - atan2(-2.0, p[1]);
        t[1] = mod2pi(phi - a), q[1] = mod2pi(phi - mod2pi(b)), ok[1] = 1;
    }
    tmp = -2.0 + d * d + 2.0 * cab - 2.0 * d * (sa + sb); /* RSL */
    if (tmp >= 0.0) {
        p[2] = sqrt(tmp);
        const double phi = atan2(ca + cb, d - sa - sb) - atan2(2.0, p[2]);
        t[2] = mod2pi(a - phi), q[2] = mod2pi(b - phi), ok[2] = 1;
    }
    tmp = 2.0 + d * d - 2.0 * cab + 2.0 * d * (sb - sa); /* RSR */
    if (tmp >= 0.0) {
        const double phi = atan2(ca - cb, d - sa + sb);
        t[3] = mod2pi(a - phi), p[3] = sqrt(tmp), q[3] = mod2pi(phi - b), ok[3] = 1;
    }
    tmp = (6.0 - d * d + 2.0 * cab + 2.0 * d * (sa - sb)) / 8.0; /* RLR */
    if (fabs(tmp) <= 1.0) {
        const double phi = atan2(ca - cb, d - sa + sb);
        p[4] = mod2pi(2.0 * REF_PI - acos(tmp));
        t[4] = mod2pi(a - phi + mod2pi(p[4] / 2.0));
        q[4] = mod2pi(a - b - t[4] + mod2pi(p[4])), ok[4] = 1;
    }
    tmp = (6.0 - d * d + 2.0 * cab + 2.0 * d * (sb - sa)) / 8.0; /* LRL */
    if (fabs(tmp) <= 1.0) {
        const double phi = atan2(ca - cb, d + sa - sb);
        p[5] = mod2pi(2.0 * REF_PI - acos(tmp));
        t[5] = mod2pi(-a - phi + p[5] / 2.0);
        q[5] = mod2pi(mod2pi(b) - a - t[5] + mod2pi(p[5])), ok[5] = 1;
    }
    ref_path best;
    best.len = HUGE_VAL;
    best.word = -1;
    best.seg[0] = best.seg[1] = best.seg[2] = 0.0;
    best.kind[0] = best.kind[1] = best.kind[2] = 0;
    for (int w = 0; w < 6; w++) {
        if (!ok[w]) continue;
        const double sum = t[w] + p[w] + q[w];
        if (sum < best.len) {
            best.len = sum;
            best.word = w;
            best.seg[0] = t[w], best.seg[1] = p[w], best.seg[2] = q[w];
            for (int k = 0; k < 3; k++) best.kind[k] = KIND[w][k];
        }
    }
    if (ties && ntie) {
        *ntie = 0;
        for (int w = 0; w < 6; w++) {
            if (!ok[w] || w == best.word) continue;
            const double sum = t[w] + p[w] + q[w];
            if (sum - best.len <= REF_TOL * (1.0 + best.len)) {
                ref_path *o = &ties[(*ntie)++];
                o->len = sum * rho;
                o->word = w;
                o->seg[0] = t[w], o->seg[1] = p[w], o->seg[2] = q[w];
                for (int k = 0; k < 3; k++) o->kind[k] = KIND[w][k];
            }
        }
    }
    if (best.word >= 0) best.len *= rho;
    return best;
}

/* the pose at arc length s (cells) along the word that starts at (x0, y0, th0): unit-speed integration segment by segment */
static void ref_pose_at(const ref_path *w, double x0, double y0, double th0, double rho, double s, double *ox, double *oy) {
    double x = x0, y = y0, th = th0, left = s / rho;
    for (int k = 0; k < 3; k++) {
        const double tau = left < w->seg[k] ? left : w->seg[k];
        if (w->kind[k] == 0) {
            x += rho * tau * cos(th);
            y += rho * tau * sin(th);
        } else if (w->kind[k] > 0) {
            x += rho * (sin(th + tau) - sin(th));
            y -= rho * (cos(th + tau) - cos(th));
            th += tau;
        } else {
            x -= rho * (sin(th - tau) - sin(th));
            y += rho * (cos(th - tau) - cos(th));
            th -= tau;
        }
        left -= tau;
        if (left <= 0.0) break;
    }
    *ox = x;
    *oy = y;
}

/* the sweep: samples every REF_DS cells from the start, then the end cell; 1 = free.  *ambig is set when a sample lies within
 * REF_AMBIG of a cell boundary (round-half-up at .5) or the sample count itself hangs on the last bits of the length */
static int ref_sweep_free(const uint8_t *og, int W, int H, const ref_path *w, double x0, double y0, double th0, int bx, int by, double rho, int *ambig) {
    *ambig = 0;
    if (w->word < 0) return 0;
    const double q = w->len / REF_DS;
    const int ns = (int)floor(q) + 1;
    if (fabs(q - floor(q + 0.5)) < 1e-6) *ambig = 1;
    for (int k = 0; k < ns; k++) {
        double x, y;
        ref_pose_at(w, x0, y0, th0, rho, (double)k * REF_DS, &x, &y);
        const double fx = x + 0.5 - floor(x + 0.5), fy = y + 0.5 - floor(y + 0.5);
        if (fx < REF_AMBIG || fx > 1.0 - REF_AMBIG || fy < REF_AMBIG || fy > 1.0 - REF_AMBIG) *ambig = 1;
        const int cx = (int)floor(x + 0.5), cy = (int)floor(y + 0.5);
        if (cx < 0 || cx >= W || cy < 0 || cy >= H || og[(int64_t)cx * H + cy] != 0) return 0;
    }
    return og[(int64_t)bx * H + by] == 0;
}

/* the edge (ax, ay, ah) -> (bx, by, bh): its shortest word (returned), whether that word's sweep is free, and whether the answer is
 * AMBIGUOUS: a sample within REF_AMBIG of a cell boundary, or a second word of the same length whose sweep says otherwise */
static ref_path ref_edge(const uint8_t *og, int W, int H, int nh, double rho, int ax, int ay, int ah, int bx, int by, int bh, int *free_, int *ambig) {
    ref_path ties[5];
    int nt = 0;
    const double th0 = 2.0 * REF_PI * (double)ah / (double)nh, th1 = 2.0 * REF_PI * (double)bh / (double)nh;
    const ref_path w = ref_shortest_ties(ax, ay, th0, bx, by, th1, rho, ties, &nt);
    *free_ = ref_sweep_free(og, W, H, &w, ax, ay, th0, bx, by, rho, ambig);
    for (int k = 0; k < nt; k++) {
        int amb2 = 0;
        if (ref_sweep_free(og, W, H, &ties[k], ax, ay, th0, bx, by, rho, &amb2) != *free_) *ambig = 1;
    }
    return w;
}

typedef struct {
    /* inputs: the problem ... */
    int32_t star, n, W, H;
    const uint8_t *og;
    int64_t r2_rewire;
    double rho;
    int32_t nh, pad_;
    const int32_t *samples;  /* (n, 2) */
    const int32_t *headings; /* (n) */
    /* ... and the tree under audit (live rows: j vertices, vertex 0 = the start pose) */
    const int32_t *pts;  /* (rows, 2) */
    const int32_t *head; /* (rows) */
    const double *vcost; /* (rows) */
    const int32_t *parent;
    int32_t j, pad2_;
    /* outputs */
    int64_t n_accepted;        /* samples the tree holds */
    int64_t accept_mismatch;   /* accept / reject differs from this file's, and this file's sweep was not ambiguous */
    int64_t accept_ambiguous;  /* differs, but the sweep passes within REF_AMBIG of a cell boundary */
    int64_t nearest_mismatch;  /* the vertex is not where the stream says (the tree is not a replay of the stream) */
    int64_t parent_is_argmin;  /* chosen parent == first minimum of this file's costs over the visible near-set vertices */
    int64_t parent_within_tol; /* another vertex, with a cost within REF_TOL of that minimum */
    int64_t parent_wrong;      /* neither */
    int64_t parent_blocked;    /* this file's sweep calls the chosen edge blocked (not ambiguous) */
    int64_t parent_blocked_ambiguous;
    int64_t cost_mismatch;     /* stored cost differs from this file's cost through the chosen parent by more than REF_TOL */
    double max_cost_err;       /* largest such difference seen (absolute) */
    int64_t words, sweeps;     /* evaluations this audit made */
    int64_t first_bad_iter;    /* the first iteration counted in accept_mismatch / parent_wrong / parent_blocked / cost_mismatch (-1: none) */
} ref_audit_t;

static double heading_angle(int h, int nh) { return 2.0 * REF_PI * (double)h / (double)nh; }

/* Replays the stream: sample i is accepted iff the tree's next vertex is that pose on a cell not sampled before (the tree's own
 * record of what the planner decided); every decision is then recomputed against the vertices [0, jcur) of the SAME tree. */
int dubref_audit(ref_audit_t *a) {
    const int n = a->n, W = a->W, H = a->H;
    if (n < 1 || a->j < 1 || !(a->rho > 0.0) || a->nh < 1) return -1;
    uint8_t *sampled = (uint8_t *)calloc((size_t)W * H, 1);
    a->n_accepted = a->accept_mismatch = a->accept_ambiguous = a->nearest_mismatch = 0;
    a->parent_is_argmin = a->parent_within_tol = a->parent_wrong = a->parent_blocked = a->parent_blocked_ambiguous = a->cost_mismatch = 0;
    a->max_cost_err = 0.0;
    a->words = a->sweeps = 0;
    a->first_bad_iter = -1;
    int jc = 1;
    for (int i = 0; i < n; i++) {
        const int x = a->samples[2 * i], y = a->samples[2 * i + 1], h = a->headings[i];
        int64_t best = INT64_MAX;
        int vn = 0;
        for (int k = 0; k < jc; k++) {
            const int64_t dx = (int64_t)a->pts[2 * k] - x, dy = (int64_t)a->pts[2 * k + 1] - y;
            const int64_t d2 = dx * dx + dy * dy;
            if (d2 < best) best = d2, vn = k;
        }
        /* what the planner did with this sample, read off its tree */
        const int dev_acc = jc < a->j && !sampled[(size_t)x * H + y] && a->pts[2 * jc] == x && a->pts[2 * jc + 1] == y && a->head[jc] == h;
        const double thx = heading_angle(h, a->nh);
        int amb = 0, free_nn = 0;
        const ref_path pn = ref_edge(a->og, W, H, a->nh, a->rho, a->pts[2 * vn], a->pts[2 * vn + 1], a->head[vn], x, y, h, &free_nn, &amb);
        a->words++, a->sweeps++;
        const int ref_acc = free_nn && !sampled[(size_t)x * H + y] && jc != n;
        if (ref_acc != dev_acc) {
            if (amb) a->accept_ambiguous++;
            else {
                a->accept_mismatch++;
                if (a->first_bad_iter < 0) a->first_bad_iter = i;
            }
        }
        if (!dev_acc) continue;
        a->n_accepted++;
        sampled[(size_t)x * H + y] = 1;
        const int pd = a->parent[jc];
        if (pd < 0 || pd >= jc) {
            a->nearest_mismatch++;
            jc++;
            continue;
        }
        if (!a->star && pd != vn) a->nearest_mismatch++;
        /* this file's cost through the planner's parent, and is that edge free? */
        const ref_path pp = ref_shortest(a->pts[2 * pd], a->pts[2 * pd + 1], heading_angle(a->head[pd], a->nh), x, y, thx, a->rho);
        a->words++;
        const double c_dev = a->vcost[pd] + pp.len;
        const double tol = REF_TOL * (1.0 + fabs(c_dev));
        const double err = fabs(a->vcost[jc] - c_dev);
        if (err > a->max_cost_err) a->max_cost_err = err;
        if (err > tol) {
            a->cost_mismatch++;
            if (a->first_bad_iter < 0) a->first_bad_iter = i;
        }
        if (pd != vn) {
            int amb2 = 0, fr = 0;
            (void)ref_edge(a->og, W, H, a->nh, a->rho, a->pts[2 * pd], a->pts[2 * pd + 1], a->head[pd], x, y, h, &fr, &amb2);
            a->sweeps++;
            if (!fr) {
                if (amb2) a->parent_blocked_ambiguous++;
                else {
                    a->parent_blocked++;
                    if (a->first_bad_iter < 0) a->first_bad_iter = i;
                }
            }
        }
        if (a->star) {
            /* the walk of rrt.py:511-521 with this file's arithmetic: nearest by default, then ascending index, strict < */
            int vbest = vn;
            double cbest = a->vcost[vn] + pn.len;
            int ambiguous_walk = 0;
            if ((double)best >= (double)a->r2_rewire && pd != vn) a->nearest_mismatch++; /* a parent outside the ball can only be the nearest */
            for (int v = 0; v < jc; v++) {
                const int64_t dx = (int64_t)a->pts[2 * v] - x, dy = (int64_t)a->pts[2 * v + 1] - y;
                if (!(dx * dx + dy * dy < a->r2_rewire)) continue;
                /* a word is never shorter than its chord: no evaluation where even the chord cannot beat the best so far */
                if (!(a->vcost[v] + sqrt((double)(dx * dx + dy * dy)) * (1.0 - 1e-9) < cbest + tol)) continue;
                const ref_path pc = ref_shortest(a->pts[2 * v], a->pts[2 * v + 1], heading_angle(a->head[v], a->nh), x, y, thx, a->rho);
                a->words++;
                const double cn = a->vcost[v] + pc.len;
                if (cn < cbest) {
                    int amb3 = 0, fr3 = 0;
                    a->sweeps++;
                    (void)ref_edge(a->og, W, H, a->nh, a->rho, a->pts[2 * v], a->pts[2 * v + 1], a->head[v], x, y, h, &fr3, &amb3);
                    if (fr3) {
                        vbest = v;
                        cbest = cn;
                    }
                    if (amb3) ambiguous_walk = 1;  /* (a candidate whose visibility hangs on the last bits: either walk is defensible) */
                }
            }
            if (vbest == pd) a->parent_is_argmin++;
            else if (fabs(c_dev - cbest) <= tol || ambiguous_walk) a->parent_within_tol++;
            else {
                a->parent_wrong++;
                if (a->first_bad_iter < 0) a->first_bad_iter = i;
            }
        } else {
            a->parent_is_argmin += (pd == vn);
        }
        jc++;
    }
    free(sampled);
    if (jc != a->j) a->nearest_mismatch += (a->j - jc); /* vertices the stream never produced */
    return 0;
}

/* primitives for the tests: the word between two poses, the cells of its sweep */
void dubref_shortest(double x0, double y0, double th0, double x1, double y1, double th1, double rho, double out[5]) {
    const ref_path w = ref_shortest(x0, y0, th0, x1, y1, th1, rho);
    out[0] = w.seg[0], out[1] = w.seg[1], out[2] = w.seg[2], out[3] = w.len, out[4] = (double)w.word;
}
