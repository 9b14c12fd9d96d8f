// rrt_block.h -- block-parallel tree expansion: 16 samples per pass of the node array.
//
// The sample stream of RRTStandard / RRTStar does not depend on the tree (rrt.py:240), so a block
// of BS = 16 consecutive samples is evaluated against the tree as it stood at the start of the
// block (the snapshot, nodes [0, j0)), and the few dependencies between samples of one block are
// resolved afterwards in sample order.  The result is bit-identical to the sequential loop.
//
//   A  scan (all 16 waves): every lane loads 4 nodes (16 bytes) once and evaluates them against the
//      16 samples held in scalar registers: coordinates pre-scaled by 16 make
//      v_dot2_i32_i16(d, d, tag) = 256*d2 + tag the packed nearest-neighbour key in ONE instruction.
//      For RRT* the wave-wide "some node of this lane's quad is within r_rewire" ballots go to LDS as
//      64-bit masks (one per sample, wave and 4096-node step).          [near :150-155, within :176-181]
//      -- barrier --
//   B  owner phase: wave k owns sample k.  It folds the 16 per-wave minima (lowest index on ties),
//      tests the line of sight snapshot-nearest -> sample, reads the `sampled` bit, decodes its masks
//      into the snapshot near set, prices it (vcost + sqrt(d2)) and finds the first entry in (cost, index)
//      order with cost < cost-via-nearest and a free line of sight          [rrt.py:424-425, :511-521]
//      -- barrier --
//   C  commit (wave 0), in sample order: if no earlier sample of the block that was inserted is
//      nearer than the snapshot nearest, within r_rewire, or the same cell, the owner's result stands;
//      otherwise the sample is re-resolved against snapshot + inserted block nodes.  Inserts the node
//      (rrt.py:524-529; the rewire scan :531-546 never fires with the default cost).  An Informed block is
//      cut where the ellipse changes (rrt.py:698-700, :744-745).
//      -- barrier --
#pragma once

#include "rrt_kernels.h"

namespace rrtdev {

constexpr int BS = 16;      // samples per block == waves per workgroup
#ifndef RRT_QCAP
#define RRT_QCAP 512
#endif
constexpr int QCAP = RRT_QCAP;  // flagged node quads an owner decodes per round (LDS, 4*QCAP bytes per owner)

// exact sqrt of an integer below 2^24 (0 included): rsq seed + coupled Goldschmidt / Newton steps in
// f64.  tests/test_gpu_parity.py checks every input against the host's correctly rounded sqrt.
__device__ __forceinline__ double sqrt_u24(uint32_t d2) {
    const double x = (double)d2;
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return d2 == 0 ? 0.0 : g;
}

// key = 256*d2 + tag of one node against one sample (both pre-scaled by 16): v_pk_sub_i16 + v_dot2_i32_i16
__device__ __forceinline__ uint32_t key16(uint32_t node_s, uint32_t q_s, uint32_t tag) {
    uint32_t d, r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(node_s), "s"(q_s));
    asm("v_dot2_i32_i16 %0, %1, %1, %2" : "=v"(r) : "v"(d), "s"(tag));
    return r;
}

// Owner's publication for one sample (64 bytes).
struct BRec {
    uint32_t d2s, vs;    // snapshot nearest
    uint32_t los_s;      // line of sight vs -> sample: bit 31 free, low bits cells read
    uint32_t flags;      // bit 0: cell already in `sampled` at the snapshot
    double Vs;           // vcost[vs]
    double pc;           // best passing snapshot near-set entry with cost < cost-via-vs (or inf)
    uint32_t pi;         //   its index (NONE)
    uint32_t pstat;      // owner's candidate line-of-sight tests: count << 20 | cells
    uint32_t nnmask;     // earlier samples of the block strictly nearer than the snapshot nearest
    uint32_t rmask;      // earlier samples within r_rewire
    uint32_t dupmask;    // earlier samples on the same cell
    uint32_t nnear;      // |within| over the snapshot
    uint32_t pad[2];
};
static_assert(sizeof(BRec) == 64, "BRec must be 64 bytes");

// Block state that wave 0 hands to the other waves after the commit.
struct BlkState {
    int32_t i, j, nsoln, vbest_soln, i_switch, status;
    double cmin_soln, c_ell;
};

template <bool STAR>
__device__ __forceinline__ void block_scan_step(u32x4 quad, const uint32_t (&xs16)[BS], uint32_t (&best)[BS], uint32_t tag0,
                                                uint32_t r2key, RRT_LDS unsigned long long *mask_row, int lane) {
    const uint32_t n0 = quad.x << 4, n1 = quad.y << 4, n2 = quad.z << 4, n3 = quad.w << 4;
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        const uint32_t k0 = key16(n0, xs16[k], tag0), k1 = key16(n1, xs16[k], tag0 + 1), k2 = key16(n2, xs16[k], tag0 + 2),
                       k3 = key16(n3, xs16[k], tag0 + 3);
        const uint32_t m4 = min(min(k0, k1), min(k2, k3));
        best[k] = min(best[k], m4);
        if (STAR) {
            // ballot(m4 < r2key) into lane k of (mlo, mhi).  One asm statement: on gfx950 a VALU-written SGPR needs
            // two wait states before another VALU reads it, and hipcc pads nothing around inline asm.
            asm("v_cmp_gt_u32_e32 vcc, %2, %3\n\ts_nop 1\n\tv_writelane_b32 %0, vcc_lo, %4\n\tv_writelane_b32 %1, vcc_hi, %4"
                : "+v"(mlo), "+v"(mhi)
                : "s"(r2key), "v"(m4), "n"(k)
                : "vcc");
        }
    }
    if (STAR && lane < BS) mask_row[lane] = ((unsigned long long)mhi << 32) | mlo;
}

__global__ __launch_bounds__(TPB) void rrt_expand_block_kernel(BatchView bv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [node cache | masks | quad lists]
    __shared__ __attribute__((aligned(16))) u32x2 nnx[BS * NWAVE];        // per sample, per wave: {d2, idx}
    __shared__ __attribute__((aligned(16))) BRec brec[BS];
    __shared__ __attribute__((aligned(16))) BSlot bslots[2 * NWAVE];
    __shared__ __attribute__((aligned(16))) BlkState blk;
    __shared__ uint32_t xq_lds[BS];
    __shared__ double newcost[BS];
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = (int)blockIdx.x;
    QDesc *D = bv.desc + q;
    if (D->status != ST_RUNNING) return;

    // ---- per-query views ----
    const int n = D->n, alg = D->alg;
    const bool star = alg >= 1, informed = alg == 2;
    const uint32_t *samples = bv.samples + (size_t)q * bv.n_cap;
    uint32_t *nodes_g = bv.nodes + (size_t)q * bv.node_stride;
    const u32x4 *nodes_g4 = reinterpret_cast<const u32x4 *>(nodes_g);
    double *vcost = bv.vcost + (size_t)q * bv.node_stride;
    int32_t *parent = bv.parent + (size_t)q * bv.node_stride;
    uint32_t *bitmap = bv.bitmap + (size_t)q * bv.bitmap_words;
    uint2 *spill = bv.spill + (size_t)q * bv.spill_stride;
    const double *ub = bv.unitball ? bv.unitball + (size_t)q * 2 * bv.n_cap : nullptr;
    const bool logs = bv.nearest_log != nullptr;
    const uint8_t *og = bv.og;
    const int W = bv.W, H = bv.H;
    const int lds_chunks = bv.lds_chunks;
    const int lds_nodes = lds_chunks * CHUNK;
    const int nsteps_cap = (bv.n_cap + CHUNK) / CHUNK;  // 4096-node steps the masks are sized for
    const uint32_t r2 = D->r2_rewire, goal_d2 = D->goal_d2;
    const uint32_t r2key = (r2 >= (1u << 23)) ? NONE : (r2 << 8);
    const uint32_t xs = pack_xy(D->xs[0], D->xs[1]), xg = pack_xy(D->xg[0], D->xg[1]);
    const int ub_offset = D->ub_offset, ub_count = D->ub_count;

    // ---- LDS carve ----
    RRT_LDS uint32_t *nodes_lds = (RRT_LDS uint32_t *)smem;
    const RRT_LDS u32x4 *nodes_lds4 = (const RRT_LDS u32x4 *)smem;
    size_t off = (size_t)lds_chunks * CHUNK * sizeof(uint32_t);
    RRT_LDS unsigned long long *masks = (RRT_LDS unsigned long long *)(smem + off);  // [wave][step][sample]
    off += (size_t)NWAVE * nsteps_cap * BS * sizeof(unsigned long long);
    RRT_LDS uint32_t *qlist = (RRT_LDS uint32_t *)(smem + off) + wave * QCAP;  // this wave's decoded quads

    // ---- state ----
    int i = D->i, j = D->j;
    int nsoln = D->nsoln, vbest_soln = D->vbest_soln;
    double cmin_soln = D->cmin_soln;
    int i_switch = D->i_switch;
    int status = ST_RUNNING;
    // statistics live in wave 0
    unsigned long long sum_j = D->sum_j, sum_cells_nn = D->sum_cells_nn, sum_near = D->sum_near,
                       sum_cells_cand = D->sum_cells_cand, n_los_cand = D->n_los_cand;
#ifdef RRT_STAMPS
    unsigned long long cyc[6] = {D->cyc[0], D->cyc[1], D->cyc[2], D->cyc[3], D->cyc[4], D->cyc[5]};
    unsigned long long tstamp = __builtin_amdgcn_s_memtime();
#endif

    const double xc0 = ((double)(D->xs[0] + D->xg[0])) / 2.0, xc1 = ((double)(D->xs[1] + D->xg[1])) / 2.0;
    const double d2sg = (double)dist2(xs, xg);
    const double C00 = D->C[0], C01 = D->C[1], C10 = D->C[2], C11 = D->C[3];
    double c_ell = 0.0;
    if (informed && nsoln > 0) c_ell = cmin_soln + sqrt_u32(dist2(xg, nodes_g[vbest_soln]));

    // ---- prologue: node cache = live nodes, unfilled slots = copy of node 0 (never nearest: equal distance,
    //      higher index; dropped from near sets by the index test) ----
    {
        const uint32_t n0 = nodes_g[0];
        for (int k = t; k < lds_nodes; k += TPB) nodes_lds[k] = (k < j) ? nodes_g[k] : n0;
    }
    __syncthreads();

    auto node_xy = [&](uint32_t v) -> uint32_t { return ((int)v < lds_nodes) ? nodes_lds[v] : nodes_g[v]; };

    // Snapshot near set of sample X (owner's masks `k`): first entry in (cost, index) order with cost < bound and a
    // free line of sight.  Executed by one whole wave.  Returns (pc, pi) or (inf, NONE); adds to the counters.
    auto snapshot_parent = [&](int k, uint32_t X, int j0, int nsteps, double bound, double &pc, uint32_t &pi, uint32_t &nnear,
                               uint32_t &ntests, uint32_t &tcells) {
        pc = f64_inf();
        pi = NONE;
        nnear = 0;
        double lbc = -1.0;
        uint32_t lbi = 0;
        const int nent = NWAVE * nsteps;  // mask words of this sample
        for (;;) {                        // branch-and-bound rounds; one round unless the cheapest entry is blocked
            Top2 tt;
            tt.init();
            uint32_t hits = 0;
            // total flagged quads, processed QCAP at a time
            uint32_t mycnt = 0;
            auto midx = [&](int e) -> size_t { return ((size_t)(e % NWAVE) * nsteps_cap + (size_t)(e / NWAVE)) * BS + (size_t)k; };
            for (int e = lane; e < nent; e += 64) mycnt += (uint32_t)__builtin_popcountll(masks[midx(e)]);
            // exclusive prefix over lanes
            uint32_t incl = mycnt;
            {
                // inclusive scan by DPP row_shr + row_bcast (same ladder as the reductions)
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t excl = incl - mycnt;
            for (uint32_t base = 0; base < total; base += QCAP) {
                // stage 0: write the ids of flagged quads [base, base+QCAP) into the LDS list
                {
                    uint32_t pos = excl;
                    for (int e = lane; e < nent; e += 64) {
                        unsigned long long m = masks[midx(e)];
                        const uint32_t w = (uint32_t)e % NWAVE, s = (uint32_t)e / NWAVE;
                        while (m) {
                            const uint32_t L = (uint32_t)__builtin_ctzll(m);
                            m &= m - 1;
                            if (pos >= base && pos < base + QCAP) qlist[pos - base] = s * TPB + w * 64 + L;
                            ++pos;
                        }
                    }
                }
                const uint32_t cnt = (total - base) < (uint32_t)QCAP ? (total - base) : (uint32_t)QCAP;
                // stage 1+2: evaluate the quads, price the hits
                for (uint32_t p = (uint32_t)lane; p < cnt; p += 64) {
                    const uint32_t qd = qlist[p];
                    u32x4 v;
                    if ((int)(qd / TPB) < lds_chunks)
                        v = nodes_lds4[qd];
                    else
                        v = nodes_g4[qd];
                    const uint32_t idx0 = qd * 4;
                    const uint32_t d0 = dist2(v.x, X), d1 = dist2(v.y, X), d2 = dist2(v.z, X), d3 = dist2(v.w, X);
                    const bool h0 = d0 < r2 && idx0 < (uint32_t)j0, h1 = d1 < r2 && idx0 + 1 < (uint32_t)j0,
                               h2 = d2 < r2 && idx0 + 2 < (uint32_t)j0, h3 = d3 < r2 && idx0 + 3 < (uint32_t)j0;
                    hits += (uint32_t)h0 + (uint32_t)h1 + (uint32_t)h2 + (uint32_t)h3;
                    // gather the four costs together, then price
                    const double v0 = h0 ? vcost[idx0] : 0.0, v1 = h1 ? vcost[idx0 + 1] : 0.0, v2 = h2 ? vcost[idx0 + 2] : 0.0,
                                 v3 = h3 ? vcost[idx0 + 3] : 0.0;
                    if (h0) {
                        const double cn = v0 + sqrt_u24(d0);
                        if (cn < bound && !key_lt(cn, idx0, lbc, lbi)) tt.fold(cn, idx0);
                    }
                    if (h1) {
                        const double cn = v1 + sqrt_u24(d1);
                        if (cn < bound && !key_lt(cn, idx0 + 1, lbc, lbi)) tt.fold(cn, idx0 + 1);
                    }
                    if (h2) {
                        const double cn = v2 + sqrt_u24(d2);
                        if (cn < bound && !key_lt(cn, idx0 + 2, lbc, lbi)) tt.fold(cn, idx0 + 2);
                    }
                    if (h3) {
                        const double cn = v3 + sqrt_u24(d3);
                        if (cn < bound && !key_lt(cn, idx0 + 3, lbc, lbi)) tt.fold(cn, idx0 + 3);
                    }
                }
            }
            nnear = wave_sum_u32(hits);
            tt.wave_reduce();
            // test in key order: the two cheapest are known; a third needs another decode round
            if (tt.i1 == NONE) return;
            int cc = 0;
            bool ok = los_wave(og, H, node_xy(tt.i1), X, lane, cc);  // rrt.py:519
            ntests += 1;
            tcells += (uint32_t)cc;
            if (ok) {
                pc = tt.c1;
                pi = tt.i1;
                return;
            }
            if (tt.i2 == NONE) return;
            ok = los_wave(og, H, node_xy(tt.i2), X, lane, cc);
            ntests += 1;
            tcells += (uint32_t)cc;
            if (ok) {
                pc = tt.c2;
                pi = tt.i2;
                return;
            }
            lbc = tt.c2;
            lbi = tt.i2 + 1;
        }
    };

    while (i < n) {
        const int i0 = i, j0 = j;
        const int nb = (n - i0) < BS ? (n - i0) : BS;
        const bool ell = informed && nsoln > 0;
        // ---------------- sample coordinates of the block (rrt.py:421 / :502 / :695-701) ----------------
        if (ell) {
            if (i_switch == n) i_switch = i0;
            const int u0i = i0 - ub_offset;
            if (ub == nullptr || u0i < 0 || u0i + nb > ub_count) {
                status = ST_NEED_UB;
                break;
            }
        }
        uint32_t xv = 0;  // lane k < nb: sample k
        if (lane < nb) {
            if (ell) {
                const int ui = i0 + lane - ub_offset;
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
                xv = pack_xy((int)vx, (int)vy);
            } else {
                xv = samples[i0 + lane];
            }
        }
        uint32_t X[BS], xs16[BS];
#pragma unroll
        for (int k = 0; k < BS; ++k) {
            X[k] = (uint32_t)__builtin_amdgcn_readlane((int)xv, k);
            if (k >= nb) X[k] = X[0];
            xs16[k] = X[k] << 4;
        }

        // ---------------- A: scan the snapshot for all samples of the block ----------------
        const int nsteps = (j0 + CHUNK - 1) / CHUNK;
        {
            uint32_t best[BS];
#pragma unroll
            for (int k = 0; k < BS; ++k) best[k] = NONE;
            const int nl = nsteps < lds_chunks ? nsteps : lds_chunks;
            if (nl > 0) {
                u32x4 cur = nodes_lds4[t];
                for (int c = 0; c < nl; ++c) {
                    u32x4 nxt = cur;
                    if (c + 1 < nl) nxt = nodes_lds4[(c + 1) * TPB + t];
                    RRT_LDS unsigned long long *row = masks + ((size_t)wave * nsteps_cap + c) * BS;
                    if (star)
                        block_scan_step<true>(cur, xs16, best, (uint32_t)c << 2, r2key, row, lane);
                    else
                        block_scan_step<false>(cur, xs16, best, (uint32_t)c << 2, r2key, row, lane);
                    cur = nxt;
                }
            }
            if (nsteps > nl) {
                u32x4 cur = nodes_g4[nl * TPB + t];
                for (int c = nl; c < nsteps; ++c) {
                    u32x4 nxt = cur;
                    if (c + 1 < nsteps) nxt = nodes_g4[(c + 1) * TPB + t];
                    RRT_LDS unsigned long long *row = masks + ((size_t)wave * nsteps_cap + c) * BS;
                    if (star)
                        block_scan_step<true>(cur, xs16, best, (uint32_t)c << 2, r2key, row, lane);
                    else
                        block_scan_step<false>(cur, xs16, best, (uint32_t)c << 2, r2key, row, lane);
                    cur = nxt;
                }
            }
            // per sample: wave minimum (lowest index among equal distance), gathered into lanes 0..15
            uint32_t gd = NONE, gi = NONE;
#pragma unroll
            for (int k = 0; k < BS; ++k) {
                uint32_t kd = best[k] >> 8;
                const uint32_t tag = best[k] & 0xffu;
                uint32_t ki = (tag >> 2) * (uint32_t)CHUNK + 4u * (uint32_t)t + (tag & 3u);
                wave_min_key_idx(kd, ki);
                if (lane == k) {
                    gd = kd;
                    gi = ki;
                }
            }
            if (lane < BS) {
                u32x2 v = {gd, gi};
                ((RRT_LDS u32x2 *)nnx)[lane * NWAVE + wave] = v;
            }
        }
        if (t < BS) xq_lds[t] = xv;  // lane k: sample k (k < nb)
        STAMP(0);
        __syncthreads();
        STAMP(1);

        // ---------------- B: owner phase, wave k resolves sample k against the snapshot ----------------
        if (wave < nb) {
            const int k = wave;
            const uint32_t Xk = (uint32_t)__builtin_amdgcn_readlane((int)xv, k);  // every wave loaded the same xv
            uint32_t d2s = NONE, vs = NONE;
            if (lane < NWAVE) {
                const u32x2 v = ((RRT_LDS u32x2 *)nnx)[k * NWAVE + lane];
                d2s = v.x;
                vs = v.y;
            }
            wave_min_key_idx(d2s, vs);
            const double Vs = vcost[vs];
            const uint32_t cell = (uint32_t)ux(Xk) * (uint32_t)H + (uint32_t)uy(Xk);
            const uint32_t bm_word = bitmap[cell >> 5];
            int cells = 0;
            const bool free_s = los_wave(og, H, node_xy(vs), Xk, lane, cells);
            // earlier samples of this block that could interact once inserted
            uint32_t xo = (lane < k) ? xq_lds[lane] : Xk;
            const uint32_t dk = dist2(xo, Xk);
            const uint32_t nnmask = (uint32_t)__ballot(lane < k && dk < d2s);
            const uint32_t rmask = (uint32_t)__ballot(lane < k && star && dk < r2);
            const uint32_t dupmask = (uint32_t)__ballot(lane < k && xo == Xk);
            double pc = f64_inf();
            uint32_t pi = NONE, nnear = 0, ntests = 0, tcells = 0;
            if (star) snapshot_parent(k, Xk, j0, nsteps, Vs + sqrt_u32(d2s), pc, pi, nnear, ntests, tcells);
            if (lane == 0) {
                BRec r;
                r.d2s = d2s;
                r.vs = vs;
                r.los_s = (free_s ? 0x80000000u : 0u) | (uint32_t)cells;
                r.flags = (bm_word >> (cell & 31)) & 1u;
                r.Vs = Vs;
                r.pc = pc;
                r.pi = pi;
                r.pstat = (ntests << 20) | (tcells & 0xfffffu);
                r.nnmask = nnmask;
                r.rmask = rmask;
                r.dupmask = dupmask;
                r.nnear = nnear;
                r.pad[0] = r.pad[1] = 0;
                brec[k] = r;
            }
        }
        STAMP(2);
        __syncthreads();
        STAMP(3);

        // ---------------- C: commit in sample order (wave 0) ----------------
        if (wave == 0) {
            uint32_t acc_mask = 0;
            int k = 0;
            for (; k < nb; ++k) {
                const BRec r = brec[k];
                const uint32_t Xk = (uint32_t)__builtin_amdgcn_readlane((int)xv, k);
                const uint32_t cell = (uint32_t)ux(Xk) * (uint32_t)H + (uint32_t)uy(Xk);
                uint32_t vn = r.vs, d2n = r.d2s;
                double Vn = r.Vs;
                bool nocoll = (r.los_s >> 31) != 0;
                uint32_t cells = r.los_s & 0x7fffffffu;
                bool dup = (r.flags & 1u) != 0;
                double pc = r.pc;
                uint32_t pi = r.pi, nnear = r.nnear;
                uint32_t ntests = r.pstat >> 20, tcells = r.pstat & 0xfffffu;
                const uint32_t inter = (r.nnmask | r.rmask | r.dupmask) & acc_mask;
                const uint32_t xo = (lane < BS) ? xq_lds[lane] : Xk;  // lane kk: sample kk
                const uint32_t dk = dist2(xo, Xk);
                const uint32_t lbit = (lane < BS) ? (1u << lane) : 0u;
                if (inter) {
                    // re-resolve against snapshot + inserted block nodes
                    dup = dup || ((r.dupmask & acc_mask) != 0);
                    const uint32_t nm = r.nnmask & acc_mask;
                    bool nn_inblock = false;
                    if (nm) {  // nearest is an inserted block node: smallest distance, earliest sample on ties
                        uint32_t kd = (nm & lbit) ? dk : NONE;
                        uint32_t kk = (uint32_t)lane;
                        wave_min_key_idx(kd, kk);
                        nn_inblock = true;
                        d2n = kd;
                        vn = (uint32_t)j0 + (uint32_t)__builtin_popcount(acc_mask & ((1u << kk) - 1u));
                        Vn = newcost[kk];
                        int cc = 0;
                        nocoll = los_wave(og, H, (uint32_t)__builtin_amdgcn_readlane((int)xo, (int)kk), Xk, lane, cc);  // rrt.py:424
                        cells = (uint32_t)cc;
                    }
                    if (star && nocoll && !dup && j != n) {
                        const double cnear = Vn + sqrt_u32(d2n);
                        if (nn_inblock) {
                            const double cnear_s = r.Vs + sqrt_u32(r.d2s);
                            if (cnear > cnear_s) {  // entries between the two bounds were never priced: redo the snapshot part
                                ntests = 0;
                                tcells = 0;
                                snapshot_parent(k, Xk, j0, nsteps, cnear, pc, pi, nnear, ntests, tcells);
                            } else if (pi != NONE && !(pc < cnear)) {
                                pc = f64_inf();
                                pi = NONE;
                            }
                        }
                        // inserted block nodes within r_rewire, in (cost, index) order, while they beat the snapshot's best
                        uint32_t rm = r.rmask & acc_mask;
                        nnear += (uint32_t)__builtin_popcount(rm);
                        while (rm) {
                            double cn = f64_inf();
                            uint32_t ci = NONE;
                            if (rm & lbit) {
                                cn = newcost[lane] + sqrt_u32(dk);
                                ci = (uint32_t)j0 + (uint32_t)__builtin_popcount(acc_mask & (lbit - 1u));
                                if (!(cn < cnear)) {
                                    cn = f64_inf();
                                    ci = NONE;
                                }
                            }
                            wave_min_f64_idx(cn, ci);
                            if (ci == NONE || !key_lt(cn, ci, pc, pi)) break;
                            // which sample is node ci
                            const uint32_t rank = ci - (uint32_t)j0;
                            uint32_t kk = 0;
                            {
                                uint32_t am = acc_mask;
                                for (uint32_t c = 0; c < rank; ++c) am &= am - 1;
                                kk = (uint32_t)__builtin_ctz(am);
                            }
                            int cc = 0;
                            const bool ok = los_wave(og, H, (uint32_t)__builtin_amdgcn_readlane((int)xo, (int)kk), Xk, lane, cc);  // rrt.py:519
                            ntests += 1;
                            tcells += (uint32_t)cc;
                            if (ok) {
                                pc = cn;
                                pi = ci;
                                break;
                            }
                            rm &= ~(1u << kk);
                        }
                    }
                }
                const bool acc = nocoll && !dup && j != n;  // rrt.py:425
                sum_j += (unsigned long long)j;
                sum_cells_nn += (unsigned long long)cells;
                if (logs && lane == 0) {
                    bv.nearest_log[(size_t)q * bv.n_cap + i0 + k] = (int32_t)vn;
                    bv.accept_log[(size_t)q * bv.n_cap + i0 + k] = (uint8_t)acc;
#ifdef RRT_DEBUG_NNEAR
                    bv.cbest_log[(size_t)q * bv.n_cap + i0 + k] = (double)nnear + (inter ? 0.5 : 0.0);
#else
                    bv.cbest_log[(size_t)q * bv.n_cap + i0 + k] = ell ? c_ell : __longlong_as_double(0x7ff8000000000000ll);
#endif
                    bv.j_log[(size_t)q * bv.n_cap + i0 + k] = j;
                }
                bool cut = false;
                if (acc) {
                    uint32_t vbest = vn;
                    double cbest = Vn + sqrt_u32(d2n);
                    if (star) {
                        sum_near += nnear;
                        n_los_cand += ntests;
                        sum_cells_cand += tcells;
                        if (pi != NONE) {
                            vbest = pi;
                            cbest = pc;
                        }
                    }
                    if (lane == 0) {
                        nodes_g[j] = Xk;
                        if (j < lds_nodes) nodes_lds[j] = Xk;
                        vcost[j] = cbest;
                        parent[j] = (int32_t)vbest;
                        atomicOr(&bitmap[cell >> 5], 1u << (cell & 31));  // rrt.py:426
                        newcost[k] = cbest;
                    }
                    if (informed && dist2(Xk, xg) < goal_d2) {  // rrt.py:744-745
                        const bool first = nsoln == 0;
                        nsoln++;
                        if (cbest < cmin_soln) {
                            cmin_soln = cbest;
                            vbest_soln = j;
                            c_ell = cmin_soln + sqrt_u32(dist2(xg, Xk));
                            cut = true;  // the ellipse changed: later samples of this block are stale (rrt.py:698-700)
                        }
                        if (first) cut = true;  // sampling switches from free space to the ellipse (rrt.py:695)
                    }
                    acc_mask |= 1u << k;
                    j++;
                }
                if (cut) {
                    ++k;
                    break;
                }
            }
            i = i0 + k;
            if (lane == 0) {
                BlkState b;
                b.i = i;
                b.j = j;
                b.nsoln = nsoln;
                b.vbest_soln = vbest_soln;
                b.i_switch = i_switch;
                b.status = status;
                b.cmin_soln = cmin_soln;
                b.c_ell = c_ell;
                blk = b;
            }
        }
        STAMP(4);
        __syncthreads();
        {
            const BlkState b = blk;
            i = b.i;
            j = b.j;
            nsoln = b.nsoln;
            vbest_soln = b.vbest_soln;
            cmin_soln = b.cmin_soln;
            c_ell = b.c_ell;
        }
    }

    __syncthreads();

    // ---------------- go2goal (rrt.py:311-332): same branch and bound as rrt_expand_kernel ----------------
    int vgoal = 0, found = 0;
    if (status == ST_RUNNING) {
        double *costs = reinterpret_cast<double *>(spill);
        for (int k = t; k < j; k += TPB) costs[k] = vcost[k] + sqrt_u32(dist2(nodes_g[k], xg));  // rrt.py:313-314
        __syncthreads();
        status = ST_DONE;
        double pc = f64_inf(), lbc = -1.0;
        uint32_t pi = NONE, lbi = 0;
        int round = 0;
        for (;;) {
            Top2 tt;
            tt.init();
            for (int k = t; k < j; k += TPB) {
                const double cn = costs[k];
                if (!key_lt(cn, (uint32_t)k, lbc, lbi) && key_lt(cn, (uint32_t)k, pc, pi)) tt.fold(cn, (uint32_t)k);
            }
            tt.wave_reduce();
            BSlot bs;
            bs.pc = f64_inf();
            bs.pi = NONE;
            bs.uc = tt.c2;
            bs.ui = tt.i2;
            bs.cells = bs.tested = 0;
            if (tt.i1 != NONE) {
                int cc = 0;
                if (los_wave(og, H, nodes_g[tt.i1], xg, lane, cc)) {  // rrt.py:318
                    bs.pc = tt.c1;
                    bs.pi = tt.i1;
                }
                lbc = tt.c1;
                lbi = tt.i1 + 1;
            }
            if (lane == 0) bslots[(round & 1) * NWAVE + wave] = bs;
            __syncthreads();
            BSlot r;
            r.pc = r.uc = f64_inf();
            r.pi = r.ui = NONE;
            if (lane < NWAVE) r = bslots[(round & 1) * NWAVE + lane];
            ++round;
            double npc = r.pc, uc = r.uc;
            uint32_t npi = r.pi, ui = r.ui;
            wave_min_f64_idx(npc, npi);
            wave_min_f64_idx(uc, ui);
            if (key_lt(npc, npi, pc, pi)) {
                pc = npc;
                pi = npi;
            }
            if (ui == NONE || !key_lt(uc, ui, pc, pi)) break;
        }
        if (pi != NONE) {
            found = 1;
            vgoal = j;  // rrt.py:319
            if (t == 0) {
                nodes_g[j] = xg;
                vcost[j] = pc;
                parent[j] = (int32_t)pi;
            }
        } else {
            if (j < n) status = ST_UNREACHABLE;
            vgoal = 0;  // rrt.py:330-331
        }
        STAMP(5);
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
#ifdef RRT_STAMPS
        for (int k = 0; k < 6; ++k) D->cyc[k] = cyc[k];
#endif
    }
}

}  // namespace rrtdev
