// rrt_kernels.h -- the tree-expansion kernel: one persistent 1024-thread workgroup per query.
//
// Replaces the loops of rrtplanner/rrt.py:418-437 (RRTStandard), :498-548 (RRTStar),
// :690-748 (RRTStarInformed) and go2goal (:311-332) of the reference.  Per iteration:
//
//   A  every thread scans its stripe of the live node array (LDS-resident chunks first,
//      HBM/L2 chunks beyond) for the nearest node (packed key min) and, for RRT*, appends
//      the nodes within r_rewire to an LDS near-set list         [near :150-155, within :176-181]
//      -- one barrier --
//   B  every wave folds the 16 per-wave minima, then tests the line of sight
//      nearest -> sample with one ballot per 64 cells, and the `sampled` bitmap
//                                                               [collisionfree :202-229, :425]
//   C  RRT*: cost vcost[v] + sqrt(d2) of every near-set entry, workgroup minimum in
//      (cost, index) order, line of sight of the winner, next-best on failure
//                                                               [choose parent :511-521]
//   D  thread 0 appends the node; the rewire scan (:531-546) is evaluated as its
//      predicate only: cost(vn -> xnew) < vcosts[vn] is never true for the default cost
//      (cost adds a non-negative distance), so it changes no state.
//
// Results are bit-identical to the sequential reference under the canonical tie policy
// (lowest index among equal distance / equal cost).
#pragma once

#include "rrt_device.h"

namespace rrtdev {

enum : int32_t { ST_DONE = 0, ST_NEED_UB = 1, ST_UNREACHABLE = -2, ST_RUNNING = 100, ST_IDLE = 101 };

// Per-query descriptor in HBM: inputs, resumable loop state, statistics.
struct QDesc {
    int32_t alg, n;
    int32_t xs[2], xg[2];
    uint32_t r2_rewire, goal_d2;
    double C[4];
    int32_t ub_offset, ub_count;
    int32_t status, i, j, nsoln, vbest_soln, vgoal, found, i_switch;
    double cmin_soln;
    unsigned long long sum_j, sum_cells_nn, sum_near, sum_cells_cand, n_los_cand;
};

struct BatchView {
    QDesc *desc;
    const uint32_t *samples;  // [Q][n_cap]         packed free-space samples
    uint32_t *nodes;          // [Q][node_stride]   packed tree nodes
    double *vcost;            // [Q][node_stride]
    int32_t *parent;          // [Q][node_stride]
    uint32_t *bitmap;         // [Q][bitmap_words]  `sampled` set (rrt.py:407)
    uint2 *spill;             // [Q][n_cap]         near-set overflow / go2goal costs
    const double *unitball;   // [Q][2*n_cap] or null
    int32_t *nearest_log;     // optional logs [Q][n_cap]
    uint8_t *accept_log;
    double *cbest_log;
    int32_t *j_log;
    const uint8_t *og;        // (W,H) x-major occupancy, != 0 is obstacle
    int32_t W, H;
    int32_t n_cap, node_stride, bitmap_words, lds_chunks;
};

struct NearList {
    uint2 *list;        // LDS [CANDCAP]
    uint32_t *count;    // LDS counter
    uint2 *spill;       // HBM overflow
};

// Wave-aggregated append of (idx, d2) for the lanes with `hit`.
__device__ __forceinline__ void near_append(const NearList &nl, bool hit, uint32_t idx, uint32_t d2, int lane) {
    unsigned long long m = __ballot(hit);
    if (m == 0) return;
    uint32_t base = 0;
    int leader = (int)__builtin_ctzll(m);
    if (lane == leader) base = atomicAdd(nl.count, (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (hit) {
        uint32_t pos = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (pos < (uint32_t)CANDCAP)
            nl.list[pos] = make_uint2(idx, d2);
        else
            nl.spill[pos - CANDCAP] = make_uint2(idx, d2);
    }
}

template <bool STAR>
__device__ __forceinline__ void eval4(uint4 v, uint32_t q, uint32_t tag0, uint32_t idx0, uint32_t r2, uint32_t &best,
                                      const NearList &nl, int lane) {
    uint32_t d0 = dist2(v.x, q), d1 = dist2(v.y, q), d2 = dist2(v.z, q), d3 = dist2(v.w, q);
    best = min(best, (d0 << 8) + tag0);
    best = min(best, (d1 << 8) + tag0 + 1);
    best = min(best, (d2 << 8) + tag0 + 2);
    best = min(best, (d3 << 8) + tag0 + 3);
    if (STAR) {
        bool h0 = d0 < r2, h1 = d1 < r2, h2 = d2 < r2, h3 = d3 < r2;
        if (__any(h0 | h1 | h2 | h3)) {
            near_append(nl, h0, idx0, d0, lane);
            near_append(nl, h1, idx0 + 1, d1, lane);
            near_append(nl, h2, idx0 + 2, d2, lane);
            near_append(nl, h3, idx0 + 3, d3, lane);
        }
    }
}

// Workgroup minimum of (c, idx) through 16 LDS slots; every wave ends with the result.
struct CSlot {
    double c;
    uint32_t idx;
    uint32_t pad;
};

__device__ __forceinline__ void block_min_f64_idx(double &c, uint32_t &idx, CSlot *slots, int wave, int lane) {
    wave_min_f64_idx(c, idx);
    if (lane == 0) {
        slots[wave].c = c;
        slots[wave].idx = idx;
    }
    __syncthreads();
    double cc = __longlong_as_double(0x7ff0000000000000ll);
    uint32_t ii = NONE;
    if (lane < NWAVE) {
        cc = slots[lane].c;
        ii = slots[lane].idx;
    }
    wave_min_f64_idx(cc, ii);
    c = cc;
    idx = ii;
}

__global__ __launch_bounds__(TPB) void rrt_expand_kernel(BatchView bv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = (int)blockIdx.x;
    QDesc *D = bv.desc + q;
    const int32_t st0 = D->status;
    if (st0 != ST_RUNNING) return;  // uniform: finished, idle, or waiting for unit-ball data

    // ---- LDS carve (every offset a multiple of 16) ----
    uint32_t *nodes_lds = reinterpret_cast<uint32_t *>(smem);
    size_t off = (size_t)bv.lds_chunks * CHUNK * sizeof(uint32_t);
    uint2 *cand_lds = reinterpret_cast<uint2 *>(smem + off);
    off += 2 * (size_t)CANDCAP * sizeof(uint2);
    CSlot *cslots = reinterpret_cast<CSlot *>(smem + off);  // [2][NWAVE]
    off += 2 * NWAVE * sizeof(CSlot);
    uint2 *nnslots = reinterpret_cast<uint2 *>(smem + off);  // [2][NWAVE]
    off += 2 * NWAVE * sizeof(uint2);
    uint32_t *cand_cnt = reinterpret_cast<uint32_t *>(smem + off);  // [2]

    // ---- per-query views ----
    const int n = D->n, alg = D->alg;
    const bool star = alg >= 1, informed = alg == 2;
    const uint32_t *samples = bv.samples + (size_t)q * bv.n_cap;
    uint32_t *nodes_g = bv.nodes + (size_t)q * bv.node_stride;
    double *vcost = bv.vcost + (size_t)q * bv.node_stride;
    int32_t *parent = bv.parent + (size_t)q * bv.node_stride;
    uint32_t *bitmap = bv.bitmap + (size_t)q * bv.bitmap_words;
    uint2 *spill = bv.spill + (size_t)q * bv.n_cap;
    const double *ub = bv.unitball ? bv.unitball + (size_t)q * 2 * bv.n_cap : nullptr;
    const bool logs = bv.nearest_log != nullptr;
    const uint8_t *og = bv.og;
    const int W = bv.W, H = bv.H;
    const int lds_nodes = bv.lds_chunks * CHUNK;
    const uint32_t r2 = D->r2_rewire, goal_d2 = D->goal_d2;
    const uint32_t xs = pack_xy(D->xs[0], D->xs[1]), xg = pack_xy(D->xg[0], D->xg[1]);
    const int ub_offset = D->ub_offset, ub_count = D->ub_count;

    // ---- resumable state (uniform registers) ----
    int i = D->i, j = D->j;
    int nsoln = D->nsoln, vbest_soln = D->vbest_soln;
    double cmin_soln = D->cmin_soln;
    int i_switch = D->i_switch;
    unsigned long long sum_j = D->sum_j, sum_cells_nn = D->sum_cells_nn, sum_near = D->sum_near,
                       sum_cells_cand = D->sum_cells_cand, n_los_cand = D->n_los_cand;
    int status = ST_RUNNING;

    // Informed: constants of the ellipse transform (rrt.py:590, :621)
    const double xc0 = ((double)(D->xs[0] + D->xg[0])) / 2.0, xc1 = ((double)(D->xs[1] + D->xg[1])) / 2.0;
    const double d2sg = (double)dist2(xs, xg);
    const double C00 = D->C[0], C01 = D->C[1], C10 = D->C[2], C11 = D->C[3];
    double c_ell = 0.0;  // rrt.py:698-699, changes only when the best solution node changes
    if (informed && nsoln > 0) c_ell = cmin_soln + sqrt_u32(dist2(xg, nodes_g[vbest_soln]));

    // ---- prologue: stage the live nodes into LDS (fresh start: node 0 only) ----
    for (int k = t; k < j && k < lds_nodes; k += TPB) nodes_lds[k] = nodes_g[k];
    if (t < 2) cand_cnt[t] = 0;
    __syncthreads();

    uint32_t pend = 0;      // node j-1 when it was inserted by the previous iteration and may
    bool pend_valid = false;  // not be visible in LDS/HBM to the other waves yet

    uint32_t s_next = (i < n) ? samples[i] : 0;

    for (; i < n; ++i) {
        const int par = i & 1;
        // ---------------- sample (rrt.py:421 / :502 / :695-701) ----------------
        uint32_t xq = s_next;
        if (i + 1 < n) s_next = samples[i + 1];
        double clog = __longlong_as_double(0x7ff8000000000000ll);
        if (informed && nsoln > 0) {
            if (i_switch == n) i_switch = i;
            const int ui = i - ub_offset;
            if (ub == nullptr || ui < 0 || ui >= ub_count) {
                status = ST_NEED_UB;
                break;
            }
            const double u0 = ub[2 * ui], u1 = ub[2 * ui + 1];
            const double ra = c_ell / 2.0;
            const double rb = sqrt(fabs(c_ell * c_ell - d2sg)) / 2.0;
            const double CL00 = C00 * ra, CL01 = C01 * rb, CL10 = C10 * ra, CL11 = C11 * rb;
            double x = __builtin_fma(CL00, u0, CL01 * u1) + xc0;
            double y = __builtin_fma(CL10, u0, CL11 * u1) + xc1;
            double vx = (x < (double)(W - 1)) ? x : (double)(W - 1);
            vx = (vx > 0.0) ? vx : 0.0;
            double vy = (y < (double)(H - 1)) ? y : (double)(H - 1);
            vy = (vy > 0.0) ? vy : 0.0;
            xq = pack_xy((int)vx, (int)vy);
            clog = c_ell;
        }
        // `sampled` bitmap word of this cell: issue the load now, use it after the scan.  A bit set
        // by the previous iteration may not be visible yet; that case is the pending node below.
        const uint32_t cell = (uint32_t)ux(xq) * (uint32_t)H + (uint32_t)uy(xq);
        const uint32_t bm_word = bitmap[cell >> 5];

        // ---------------- A: scan the live nodes ----------------
        NearList nl{cand_lds + par * CANDCAP, cand_cnt + par, spill};
        uint32_t best = NONE;
        const int nfull = (j - 1) / CHUNK;  // chunks that hold only committed nodes
        for (int c = 0; c < nfull; ++c) {
            uint4 v = (c < bv.lds_chunks) ? reinterpret_cast<const uint4 *>(nodes_lds)[c * TPB + t]
                                          : reinterpret_cast<const uint4 *>(nodes_g)[c * TPB + t];
            if (star)
                eval4<true>(v, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, nl, lane);
            else
                eval4<false>(v, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, nl, lane);
        }
        {  // tail chunk: masked, the pending node substituted from registers
            const int c = nfull;
            const int idx0 = c * CHUNK + 4 * t;
            if (idx0 < j) {
                uint4 v = (c < bv.lds_chunks) ? reinterpret_cast<const uint4 *>(nodes_lds)[c * TPB + t]
                                              : reinterpret_cast<const uint4 *>(nodes_g)[c * TPB + t];
                uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = idx0 + e;
                    if (idx < j) {
                        uint32_t p = (pend_valid && idx == j - 1) ? pend : pv[e];
                        uint32_t d = dist2(p, xq);
                        best = min(best, (d << 8) + ((uint32_t)c << 2) + (uint32_t)e);
                        if (star && d < r2) {
                            // tail hits are rare: plain per-lane append
                            uint32_t pos = atomicAdd(nl.count, 1u);
                            if (pos < (uint32_t)CANDCAP)
                                nl.list[pos] = make_uint2((uint32_t)idx, d);
                            else
                                nl.spill[pos - CANDCAP] = make_uint2((uint32_t)idx, d);
                        }
                    }
                }
            }
        }
        {
            uint32_t kd = best >> 8, tag = best & 0xffu;
            uint32_t ki = (best == NONE) ? NONE : (tag >> 2) * (uint32_t)CHUNK + 4u * (uint32_t)t + (tag & 3u);
            wave_min_key_idx(kd, ki);
            if (lane == 0) nnslots[par * NWAVE + wave] = make_uint2(kd, ki);
        }
        __syncthreads();  // barrier 1: per-wave minima, near-set list and last iteration's node are visible

        // ---------------- B: nearest, line of sight, duplicate test ----------------
        uint32_t d2n = NONE, vn = NONE;
        if (lane < NWAVE) {
            uint2 s = nnslots[par * NWAVE + lane];
            d2n = s.x;
            vn = s.y;
        }
        wave_min_key_idx(d2n, vn);
        const uint32_t pn = (pend_valid && (int)vn == j - 1) ? pend : ((int)vn < lds_nodes ? nodes_lds[vn] : nodes_g[vn]);
        const double vc_near = vcost[vn];
        int cells = 0;
        const bool nocoll = los_wave(og, H, pn, xq, lane, cells);
        const bool dup = ((bm_word >> (cell & 31)) & 1u) || (pend_valid && pend == xq);
        const bool acc = nocoll && !dup && j != n;  // rrt.py:425
        sum_j += (unsigned long long)j;
        sum_cells_nn += (unsigned long long)cells;
        if (logs && t == 0) {
            bv.nearest_log[(size_t)q * bv.n_cap + i] = (int32_t)vn;
            bv.accept_log[(size_t)q * bv.n_cap + i] = (uint8_t)acc;
            bv.cbest_log[(size_t)q * bv.n_cap + i] = clog;
            bv.j_log[(size_t)q * bv.n_cap + i] = j;
        }
        if (!acc) {
            if (star && t == 0) cand_cnt[par] = 0;  // nobody reads this list
            pend_valid = false;
            continue;
        }

        // ---------------- C: choose parent (rrt.py:511-521) ----------------
        uint32_t vbest = vn;
        double cbest = vc_near + sqrt_u32(d2n);
        if (star) {
            const uint32_t m = cand_cnt[par];
            sum_near += m;
            const double cnear = cbest;
            double floor_c = -1.0;
            uint32_t floor_i = 0;
            int round = 0;
            for (;;) {
                double bc = __longlong_as_double(0x7ff0000000000000ll);
                uint32_t bi = NONE;
                for (uint32_t c = (uint32_t)t; c < m; c += TPB) {
                    const uint2 e = (c < (uint32_t)CANDCAP) ? nl.list[c] : nl.spill[c - CANDCAP];
                    const double cn = vcost[e.x] + sqrt_u32(e.y);
                    if (!(cn < cnear)) continue;  // rrt.py:518, strict
                    if (cn < floor_c || (cn == floor_c && e.x <= floor_i)) continue;  // already refused
                    if (cn < bc || (cn == bc && e.x < bi)) {
                        bc = cn;
                        bi = e.x;
                    }
                }
                block_min_f64_idx(bc, bi, cslots + (round & 1) * NWAVE, wave, lane);
                ++round;
                if (bi == NONE) break;  // nearest stays the parent
                const uint32_t pb = (pend_valid && (int)bi == j - 1) ? pend
                                                                     : ((int)bi < lds_nodes ? nodes_lds[bi] : nodes_g[bi]);
                int cc = 0;
                const bool ok = los_wave(og, H, pb, xq, lane, cc);  // rrt.py:519
                sum_cells_cand += (unsigned long long)cc;
                n_los_cand += 1;
                if (ok) {
                    vbest = bi;
                    cbest = bc;
                    break;
                }
                floor_c = bc;
                floor_i = bi;
            }
            if (t == 0) cand_cnt[par] = 0;  // every thread has consumed the list (barrier inside block_min)
        }

        // ---------------- D: insert (rrt.py:524-529); rewire scan :531-546 is vacuous ----------------
        if (t == 0) {
            nodes_g[j] = xq;
            if (j < lds_nodes) nodes_lds[j] = xq;
            vcost[j] = cbest;
            parent[j] = (int32_t)vbest;
            atomicOr(&bitmap[cell >> 5], 1u << (cell & 31));  // rrt.py:426
        }
        if (informed && dist2(xq, xg) < goal_d2) {  // rrt.py:744-745
            nsoln++;
            if (cbest < cmin_soln) {  // np.argmin keeps the first minimum (rrt.py:632)
                cmin_soln = cbest;
                vbest_soln = j;
                c_ell = cmin_soln + sqrt_u32(dist2(xg, xq));
            }
        }
        pend = xq;
        pend_valid = true;
        j++;
    }

    __syncthreads();  // last insert visible to every wave

    // ---------------- go2goal (rrt.py:311-332), only when the loop has finished ----------------
    int vgoal = 0, found = 0;
    if (status == ST_RUNNING) {
        double *costs = reinterpret_cast<double *>(spill);
        for (int k = t; k < j; k += TPB) costs[k] = vcost[k] + sqrt_u32(dist2(nodes_g[k], xg));  // rrt.py:313-314
        __syncthreads();
        double floor_c = -1.0;
        uint32_t floor_i = 0;
        int round = 0;
        status = ST_DONE;
        for (;;) {  // np.argsort(costs) order (stable), first node with line of sight (rrt.py:317-318)
            double bc = __longlong_as_double(0x7ff0000000000000ll);
            uint32_t bi = NONE;
            for (int k = t; k < j; k += TPB) {
                const double cn = costs[k];
                if (cn < floor_c || (cn == floor_c && (uint32_t)k <= floor_i)) continue;
                if (cn < bc || (cn == bc && (uint32_t)k < bi)) {
                    bc = cn;
                    bi = (uint32_t)k;
                }
            }
            block_min_f64_idx(bc, bi, cslots + (round & 1) * NWAVE, wave, lane);
            ++round;
            if (bi == NONE) break;
            int cc = 0;
            if (los_wave(og, H, nodes_g[bi], xg, lane, cc)) {
                found = 1;
                vgoal = j;  // rrt.py:319
                if (t == 0) {
                    nodes_g[j] = xg;
                    vcost[j] = bc;
                    parent[j] = (int32_t)bi;
                }
                break;
            }
            floor_c = bc;
            floor_i = bi;
        }
        if (!found) {
            if (j < n) status = ST_UNREACHABLE;  // the next argsort entry is a sentinel row (rrt.py:318 faults)
            vgoal = 0;                           // rrt.py:330-331
        }
    }

    if (t == 0) {
        D->status = status;
        D->i = i;
        D->j = j;
        D->nsoln = nsoln;
        D->vbest_soln = vbest_soln;
        D->cmin_soln = cmin_soln;
        D->vgoal = vgoal;
        D->found = found;
        D->i_switch = i_switch;
        D->sum_j = sum_j;
        D->sum_cells_nn = sum_cells_nn;
        D->sum_near = sum_near;
        D->sum_cells_cand = sum_cells_cand;
        D->n_los_cand = n_los_cand;
    }
}

// Arms query state in HBM: clears the `sampled` bitmap, writes node 0 (rrt.py:411-413).
__global__ void rrt_init_kernel(BatchView bv) {
    const int q = (int)blockIdx.y;
    const QDesc *D = bv.desc + q;
    if (D->status != ST_RUNNING || D->i != 0) return;
    uint32_t *bitmap = bv.bitmap + (size_t)q * bv.bitmap_words;
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < bv.bitmap_words; k += (int)(gridDim.x * blockDim.x))
        bitmap[k] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bv.nodes[(size_t)q * bv.node_stride] = pack_xy(D->xs[0], D->xs[1]);
        bv.vcost[(size_t)q * bv.node_stride] = 0.0;
        bv.parent[(size_t)q * bv.node_stride] = -1;
    }
}

}  // namespace rrtdev
