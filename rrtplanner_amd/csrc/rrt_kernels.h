// rrt_kernels.h -- the tree-expansion kernel: one persistent 1024-thread workgroup per query.
//
// Replaces the loops of rrtplanner/rrt.py:418-437 (RRTStandard), :498-548 (RRTStar),
// :690-748 (RRTStarInformed) and go2goal (:311-332) of the reference.  One barrier per iteration:
//
//   A  every thread scans its stripe of the live node array (LDS-resident chunks, then HBM/L2
//      chunks; 16-byte loads, next chunk prefetched) for the nearest node (packed key min); for RRT*
//      each wave appends the nodes within r_rewire to its own LDS near-set list (no atomics)
//                                                               [near :150-155, within :176-181]
//      then, still before the barrier and speculatively, each wave
//        - tests the line of sight  (its local nearest) -> sample       [collisionfree :202-229]
//        - prices its near-set entries, cost = vcost[v] + sqrt(d2), finds its best and second best in
//          (cost, index) order and tests the line of sight of its best  [choose parent :511-521]
//      and publishes all of it in one 64-byte LDS slot
//      -- barrier --
//   B  every wave folds the 16 slots: global nearest (lowest index on ties), its line of sight and the
//      `sampled` bitmap give the accept decision (:425); the cheapest passing wave-best that no untested
//      entry can beat is the parent; only if an untested entry could still win do extra
//      branch-and-bound rounds run (each with a barrier)
//   D  thread 0 appends the node; the rewire scan (:531-546) is evaluated as its predicate only:
//      cost(vn -> xnew) < vcosts[vn] is never true for the default cost (it adds a non-negative
//      distance), so it changes no state.
//
// Results are bit-identical to the sequential reference under the canonical tie policy
// (lowest index among equal distance / equal cost).
#pragma once

#include "rrt_device.h"
#include "rrt_dubins_dev.h"

namespace rrtdev {

enum : int32_t { ST_DONE = 0, ST_NEED_UB = 1, ST_UNREACHABLE = -2, ST_TEAM_FAIL = -3, ST_RUNNING = 100, ST_IDLE = 101 };

// Per-query descriptor in HBM: inputs, resumable loop state, statistics.
struct QDesc {
    int32_t alg, n;
    int32_t xs[2], xg[2];
    uint32_t r2_rewire, goal_d2;
    double C[4];
    int32_t ub_offset, ub_count;
    int32_t cell_shift, ncx, ncy, cell_cap;  // block kernel: near-set record grid of this query (cell = 2^shift pixels)
    int32_t status, i, j, nsoln, vbest_soln, vgoal, found, i_switch;
    double cmin_soln;
    unsigned long long sum_j, sum_cells_nn, sum_near, sum_cells_cand, n_los_cand;
    unsigned long long wcyc[32]; // diagnostic build: per-wave cycles in the block kernel's owner phase [0..15] and its LoS part [16..31]
    unsigned long long cyc[6];  // diagnostic build (-DRRT_STAMPS): wave-0 cycles in scan / pre-barrier / barrier / B+C / D / go2goal
    unsigned long long n_rewired, n_propagated;  // opt-in true rewire (RRT_FLAG_REWIRE): nodes re-parented, descendant costs recomputed
    double rho;          // Dubins planners (alg 3 / 4): turning radius in cells, number of headings, start / goal heading index
    int32_t nh, hs, hg, pad_;
    unsigned long long n_words;  // Dubins planners: dub_shortest() evaluations made (the byte / flop model counts one per near-set entry)
#ifdef RRT_STAMPS
    // diagnostic build, pipelined teams: per worker m = member - 1: [m] polls of the committer's record fetch during which m's records were
    // still missing, [64 + m] blocks in which m was the LAST to arrive, [128 + m] the worker's own cycles in its resolve phase
    unsigned long long dbg2[448];  // (+ [192 + m], [256 + m], [320 + m]: blocks whose resolve phase took m more than 26 k / 32 k / 40 k cycles, [384 + m]: its longest)
    unsigned long long ts[32 * 16];  // wall-clock (10 ns) time stamps of 32 consecutive blocks, 16 events each (rrt_block.h: TSMARK)
#endif
};

struct BatchView {
    QDesc *desc;
    const uint32_t *samples;  // [Q][n_cap]         packed free-space samples
    uint32_t *nodes;          // [Q][node_stride]   packed tree nodes
    double *vcost;            // [Q][node_stride]
    int32_t *parent;          // [Q][node_stride]
    uint32_t *bitmap;         // [Q][bitmap_words]  `sampled` set (rrt.py:407)
    uint2 *spill;             // [Q][spill_stride]  per-wave near-set overflow / go2goal costs
    const double *unitball;   // [Q][2*n_cap] or null
    int32_t *nearest_log;     // optional logs [Q][n_cap]
    uint8_t *accept_log;
    double *cbest_log;
    int32_t *j_log;
    const uint8_t *og;        // (W,H) x-major occupancy, != 0 is obstacle
    int32_t W, H;
    int32_t n_cap, node_stride, bitmap_words, lds_chunks, spill_stride;
    uint4 *cellrec;           // [Q][rec_stride]    block kernel: per-cell arrays of {xy, index, vcost} records
    uint32_t *cellcnt;        // [Q][MAX_CELLS]     fill counts of the cells
    int64_t rec_stride;
    unsigned char *team;      // [Q][TEAM_BYTES]    block kernel with teams: sync words, state, exchanged records
    int32_t Q, team_qpad;     // queries of the batch; block stride between the members of a team (block = member * team_qpad + query)
    int32_t team_fault;       // testing: member 1 of every team leaves at once (the others' hand-offs time out)
    int32_t member0;          // added to the member number a team kernel derives from its block index (1: a launch of the workers only)
    // opt-in true rewire (RRT_FLAG_REWIRE; serial kernel only), null otherwise
    int32_t *kid_first, *kid_next, *kid_prev;  // [Q][node_stride] child lists: first child, next / previous sibling (-1 = none)
    uint32_t *frontier;                        // [Q][2 * node_stride] two propagation frontiers
    int32_t *vsoln;                            // [Q][node_stride] Informed: solution vertices in insertion order
    // Dubins planners (RRT_FLAG_DUBINS; serial kernel only), null otherwise
    uint8_t *heading;               // [Q][node_stride] heading index of every node
    const uint8_t *sample_heading;  // [Q][n_cap]       heading index of sample i
    double *dub_path;               // [Q][NWAVE * WCAP][5] {t, p, q, len, word} of every priced near-set entry of the current iteration
};

constexpr int BLOCK_LIST_CAP = 256;                                  // block kernel, one wave per sample: parked entries per wave kept in LDS
constexpr size_t BLOCK_LIST_LDS_BYTES = (size_t)NWAVE * BLOCK_LIST_CAP * 16;  // 64 KiB
constexpr int MAX_CELLS = 4096;  // cells per query (their fill counts live in LDS: 16 KiB)

// Near set of one wave: every wave keeps the within-radius nodes of its own stripe in its own LDS
// region (WCAP entries {idx, d2}, overflow to its own HBM region) and later prices them itself, so
// the append needs no atomics: the fill count is a wave-uniform register.
struct WaveList {
    RRT_LDS u32x2 *list;  // LDS [WCAP] of this wave
    u32x2 *spill;         // HBM overflow of this wave
};

__device__ __forceinline__ void wl_store(const WaveList &wl, uint32_t pos, uint32_t idx, uint32_t d2) {
    u32x2 v = {idx, d2};
    if (pos < (uint32_t)WCAP) wl.list[pos] = v;
    if (pos >= (uint32_t)WCAP) wl.spill[pos - WCAP] = v;
}

// Append the lanes flagged in h0..h3 (element e of each lane's 4-node load).  Wave-uniform control flow.
__device__ __forceinline__ void wl_append4(const WaveList &wl, uint32_t &wcnt, bool h0, bool h1, bool h2, bool h3,
                                           uint32_t idx0, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, int lane) {
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
    if ((m0 | m1 | m2 | m3) == 0) return;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (h0) wl_store(wl, wcnt + (uint32_t)__builtin_popcountll(m0 & lt), idx0, d0);
    wcnt += (uint32_t)__builtin_popcountll(m0);
    if (h1) wl_store(wl, wcnt + (uint32_t)__builtin_popcountll(m1 & lt), idx0 + 1, d1);
    wcnt += (uint32_t)__builtin_popcountll(m1);
    if (h2) wl_store(wl, wcnt + (uint32_t)__builtin_popcountll(m2 & lt), idx0 + 2, d2);
    wcnt += (uint32_t)__builtin_popcountll(m2);
    if (h3) wl_store(wl, wcnt + (uint32_t)__builtin_popcountll(m3 & lt), idx0 + 3, d3);
    wcnt += (uint32_t)__builtin_popcountll(m3);
}

template <bool STAR>
__device__ __forceinline__ void eval4(u32x4 v, uint32_t q, uint32_t tag0, uint32_t idx0, uint32_t r2, uint32_t &best,
                                      const WaveList &wl, uint32_t &wcnt, int lane) {
    uint32_t d0 = dist2(v.x, q), d1 = dist2(v.y, q), d2 = dist2(v.z, q), d3 = dist2(v.w, q);
    best = min(best, (d0 << 8) + tag0);
    best = min(best, (d1 << 8) + tag0 + 1);
    best = min(best, (d2 << 8) + tag0 + 2);
    best = min(best, (d3 << 8) + tag0 + 3);
    if (STAR) wl_append4(wl, wcnt, d0 < r2, d1 < r2, d2 < r2, d3 < r2, idx0, d0, d1, d2, d3, lane);
}

// One wave's publication for the iteration's barrier (64 bytes).
struct Slot {
    uint32_t d2n, vn;  // local nearest
    uint32_t wcnt;     // near-set entries of this wave
    uint32_t nn_los;   // line of sight local nearest -> sample: bit 31 = free, low bits = cells read
    double vc_nn;      // vcost of the local nearest
    double c1, c2;     // best / second best priced near-set entry of this wave, (cost, index) order
    uint32_t i1, i2;
    uint32_t los1;     // line of sight i1 -> sample: bit 31 = free, low bits = cells read
    uint32_t pad[3];
};
static_assert(sizeof(Slot) == 64, "Slot must be 64 bytes");

// Lane-local running best and second best in (cost, index) order.
struct Top2 {
    double c1, c2;
    uint32_t i1, i2;
    __device__ __forceinline__ void init() {
        c1 = c2 = f64_inf();
        i1 = i2 = NONE;
    }
    __device__ __forceinline__ void fold(double c, uint32_t i) {
        if (key_lt(c, i, c1, i1)) {
            c2 = c1;
            i2 = i1;
            c1 = c;
            i1 = i;
        } else if (key_lt(c, i, c2, i2)) {
            c2 = c;
            i2 = i;
        }
    }
    // wave-wide best and second best of all lanes' entries (uniform result)
    __device__ __forceinline__ void wave_reduce() {
        double bc = c1;
        uint32_t bi = i1;
        wave_min_f64_idx(bc, bi);
        const bool own = (c1 == bc && i1 == bi);
        double sc = own ? c2 : c1;
        uint32_t si = own ? i2 : i1;
        wave_min_f64_idx(sc, si);
        c1 = bc;
        i1 = bi;
        c2 = sc;
        i2 = si;
    }
};

// Workgroup exchange for the (rare) extra branch-and-bound rounds.
struct BSlot {
    double pc, uc;  // passing key of this round (or inf), best still-untested key (or inf)
    uint32_t pi, ui;
    uint32_t cells, tested;
};

// ---------------------------------------------------------------------------------------------------------------
// go2goal (rrt.py:311-332): the goal connects to the first node, in stable (cost, index) order of
// cost = vcost[k] + dist(k, goal), that has line of sight to it.  Nodes are counting-sorted into G2G_NB cost buckets
// (LDS histogram, monotone bucket function), then tested in bucket order, 16 waves x G2G_U nodes per round; the
// search ends once every node of the bucket that holds the cheapest passing node has been tested.
constexpr int G2G_NB = 2048;  // cost buckets (2 x 8 KiB of LDS: fill cursors and bucket ends)
constexpr int G2G_U = 4;      // nodes a wave tests per round

// The nodes considered are kfirst + m * kstep, m < cnt (all of them: 0, 1, j; a team gives each member a stripe and takes the
// minimum of the stripes' answers).
// NT: threads of the calling workgroup (a pipelined team's committer may run as a workgroup of its own with fewer waves).
template <bool DUB = false, int NT = TPB>
__device__ __forceinline__ void go2goal_phase(const uint8_t *og, int H, const uint32_t *nodes_g, const double *vcost, int kfirst, int kstep,
                                              int cnt, uint32_t xg, uint32_t *order, RRT_LDS uint32_t *lds16k, BSlot *bslots, int t, int lane,
                                              int wave, double &pc, uint32_t &pi, const uint8_t *heading = nullptr, int hg = 0, DubCfg dc = DubCfg{}) {
    RRT_LDS uint32_t *cursor = lds16k;          // [G2G_NB]
    RRT_LDS uint32_t *bend = lds16k + G2G_NB;   // [G2G_NB]
    auto cost_of = [&](int k) -> double {  // rrt.py:313-314
        if (DUB) return vcost[k] + dub_between_dev(nodes_g[k], heading[k], xg, hg, dc).len;
        return vcost[k] + sqrt_u32(dist2(nodes_g[k], xg));
    };
    // ---- cost range ----
    double cmin = f64_inf(), cmax = 0.0;
    for (int m = t; m < cnt; m += NT) {
        const double c = cost_of(kfirst + m * kstep);
        cmin = c < cmin ? c : cmin;
        cmax = c > cmax ? c : cmax;
    }
    {
        uint32_t dummy = 0;
        wave_min_f64_idx(cmin, dummy);
        double neg = -cmax;  // max via min on the order-reversed bit pattern is not available for negatives: use bits of cmax directly
        (void)neg;
        // max of non-negative doubles == max of their bit patterns; reduce as min of the complement
        unsigned long long mb = ~(unsigned long long)__double_as_longlong(cmax);
        uint32_t hi = (uint32_t)(mb >> 32), lo = (uint32_t)mb;
        const uint32_t mh = wave_min_u32(hi);
        const uint32_t ml = wave_min_u32(hi == mh ? lo : NONE);
        cmax = __longlong_as_double((long long)~(((unsigned long long)mh << 32) | ml));
        if (lane == 0) {
            bslots[wave].pc = cmin;
            bslots[wave].uc = cmax;
        }
        __syncthreads();
        double a = f64_inf(), b = 0.0;
        for (int w = 0; w < NT / 64; ++w) {
            const double x = bslots[w].pc, y = bslots[w].uc;
            a = x < a ? x : a;
            b = y > b ? y : b;
        }
        cmin = a;
        cmax = b;
        __syncthreads();
    }
    const double scale = (cmax > cmin) ? (double)(G2G_NB - 1) / (cmax - cmin) : 0.0;
    auto bucket_of = [&](double c) -> uint32_t {  // monotone non-decreasing in c
        const double f = (c - cmin) * scale;
        uint32_t b = (uint32_t)f;
        return b > (uint32_t)(G2G_NB - 1) ? (uint32_t)(G2G_NB - 1) : b;
    };
    // ---- histogram, exclusive scan, scatter ----
    for (int b = t; b < 2 * G2G_NB; b += NT) lds16k[b] = 0;
    __syncthreads();
    for (int m = t; m < cnt; m += NT)
        __hip_atomic_fetch_add(&cursor[bucket_of(cost_of(kfirst + m * kstep))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    {
        // thread t owns the PB consecutive buckets PB t .. PB t + PB - 1 (G2G_NB == PB * NT)
        constexpr int PB = G2G_NB / NT;
        static_assert(PB * NT == G2G_NB, "buckets per thread");
        uint32_t cb[PB], own = 0;
#pragma unroll
        for (int e = 0; e < PB; ++e) {
            cb[e] = cursor[PB * t + e];
            own += cb[e];
        }
        uint32_t incl = own;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
        if (lane == 63) bslots[wave].pi = incl;  // wave total
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += bslots[w].pi;
        uint32_t ex = base + incl - own;
#pragma unroll
        for (int e = 0; e < PB; ++e) {
            cursor[PB * t + e] = ex;
            ex += cb[e];
            bend[PB * t + e] = ex;
        }
        __syncthreads();
    }
    for (int m = t; m < cnt; m += NT) {
        const int k = kfirst + m * kstep;
        const uint32_t pos = __hip_atomic_fetch_add(&cursor[bucket_of(cost_of(k))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        order[pos] = (uint32_t)k;
    }
    __syncthreads();
    // ---- test in bucket order ----
    pc = f64_inf();
    pi = NONE;
    uint32_t limit = (uint32_t)cnt;
    int round = 0;
    for (uint32_t pos0 = 0; pos0 < limit; pos0 += (NT / 64) * G2G_U) {
        double bc = f64_inf();
        uint32_t bi = NONE;
#pragma unroll
        for (int u = 0; u < G2G_U; ++u) {
            const uint32_t p = pos0 + (uint32_t)(u * (NT / 64) + wave);
            if (p < limit) {
                const uint32_t k = order[p];
                int cc = 0;
                bool free_k;
                if (DUB) {
                    const dub_path_t pth = dub_between_dev(nodes_g[k], heading[k], xg, hg, dc);
                    free_k = dub_sweep_wave(og, dc, nodes_g[k], heading[k], xg, pth, lane, cc);
                } else {
                    free_k = los_wave(og, H, nodes_g[k], xg, lane, cc);
                }
                if (free_k) {  // rrt.py:318
                    const double c = cost_of((int)k);
                    if (key_lt(c, k, bc, bi)) {
                        bc = c;
                        bi = k;
                    }
                }
            }
        }
        BSlot *sl = bslots + (round & 1) * NWAVE;
        if (lane == 0) {
            sl[wave].pc = bc;
            sl[wave].pi = bi;
        }
        __syncthreads();
        double rc = f64_inf();
        uint32_t ri = NONE;
        if (lane < NT / 64) {
            rc = sl[lane].pc;
            ri = sl[lane].pi;
        }
        wave_min_f64_idx(rc, ri);
        ++round;
        if (key_lt(rc, ri, pc, pi)) {
            pc = rc;
            pi = ri;
            const uint32_t e = bend[bucket_of(pc)];  // every node that could sort before it lies before this position
            limit = e < limit ? e : limit;
        }
    }
}

#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
#define STAMP(k)                                                \
    do {                                                        \
        unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        cyc[k] += now_ - tstamp;                                \
        tstamp = now_;                                          \
    } while (0)
#else
#define STAMP(k) \
    do {         \
    } while (0)
#endif

// RW = true: the opt-in true RRT* rewire of SURVEY.md 8(f) row 4 (NOT the reference's behaviour; oracle/rrt_oracle.c states
// the semantics): after an insertion every near-set entry vn with vcost[vnew] + dist < vcost[vn] and a free line of sight
// vn -> xnew is re-parented to the new node, all decisions taken against the costs right after the insertion; then the costs
// of the re-parented nodes' descendants are recomputed level by level over explicit child lists.
// DUB = true: the Dubins planners (alg 3 Dubins-RRT, 4 Dubins-RRT*; no reference counterpart, include/rrt_dubins.h): every
// node and sample carries a heading, an edge is the shortest Dubins word between the two poses, its cost the word's arc
// length, its collision test the sampled sweep of the word; nearest / within / accept / choose-parent order are unchanged.
// (RRT_SERIAL_DECL_ONLY: a translation unit that only launches the kernel; csrc/kernels_tu.hip instantiates it)
template <bool RW, bool DUB = false>
__global__ __launch_bounds__(TPB) void rrt_expand_kernel(BatchView bv)
#ifdef RRT_SERIAL_DECL_ONLY
    ;
#else
{
    static_assert(!(RW && DUB), "the opt-in rewire is not built for the Dubins planners");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // node cache: lds_chunks * 16 KiB
    __shared__ __attribute__((aligned(16))) u32x2 wlist_lds[NWAVE * WCAP];
    __shared__ __attribute__((aligned(16))) Slot slots[2 * NWAVE];
    __shared__ __attribute__((aligned(16))) BSlot bslots[2 * NWAVE];
    __shared__ uint32_t rw_cnt[2];  // RW: re-parented nodes of this insertion / fill of the next propagation frontier
    // DUB: the near set of the iteration packed over all waves (cost through each entry, its node), the waves' scan results, and
    // what the wave that tests the nearest node found
    constexpr int PKCAP = DUB ? NWAVE * WCAP : 1;
    __shared__ __attribute__((aligned(16))) double pk_cost[PKCAP];
    __shared__ __attribute__((aligned(16))) uint32_t pk_idx[PKCAP];
    __shared__ __attribute__((aligned(16))) u32x4 dub_slot[2 * NWAVE];
    __shared__ struct {
        double cost;
        uint32_t ok, cells;
    } dub_nn;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = (int)blockIdx.x;
    QDesc *D = bv.desc + q;
    const int32_t st0 = D->status;
    if (st0 != ST_RUNNING) return;  // uniform: finished, idle, or waiting for unit-ball data

    RRT_LDS uint32_t *nodes_lds = (RRT_LDS uint32_t *)smem;
    const RRT_LDS u32x4 *nodes_lds4 = (const RRT_LDS u32x4 *)smem;

    // ---- per-query views ----
    const int n = D->n, alg = D->alg;
    const bool star = DUB ? alg == 4 : alg >= 1, informed = !DUB && alg == 2;
    uint8_t *heading = DUB ? bv.heading + (size_t)q * bv.node_stride : nullptr;
    const uint8_t *shead = DUB ? bv.sample_heading + (size_t)q * bv.n_cap : nullptr;
    const DubCfg dc{DUB ? D->rho : 1.0, DUB ? D->nh : 1, bv.W, bv.H};
    double *const dpath = DUB ? bv.dub_path + (size_t)q * (size_t)(NWAVE * WCAP) * 5 : nullptr;
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
    const uint32_t r2 = D->r2_rewire, goal_d2 = D->goal_d2;
    const uint32_t xs = pack_xy(D->xs[0], D->xs[1]), xg = pack_xy(D->xg[0], D->xg[1]);
    const int ub_offset = D->ub_offset, ub_count = D->ub_count;
    // this wave's near-set list: LDS region + HBM overflow region (256 entries per node chunk)
    const WaveList wl{(RRT_LDS u32x2 *)wlist_lds + wave * WCAP,
                      reinterpret_cast<u32x2 *>(spill) + (size_t)wave * (size_t)(bv.spill_stride / NWAVE)};

    // ---- resumable state (uniform registers) ----
    int i = D->i, j = D->j;
    int nsoln = D->nsoln, vbest_soln = D->vbest_soln;
    double cmin_soln = D->cmin_soln;
    int i_switch = D->i_switch;
    unsigned long long sum_j = D->sum_j, sum_cells_nn = D->sum_cells_nn, sum_near = D->sum_near,
                       sum_cells_cand = D->sum_cells_cand, n_los_cand = D->n_los_cand;
    int status = ST_RUNNING;
#ifdef RRT_STAMPS
    unsigned long long cyc[6] = {D->cyc[0], D->cyc[1], D->cyc[2], D->cyc[3], D->cyc[4], D->cyc[5]};
    unsigned long long tstamp = __builtin_amdgcn_s_memtime();
#endif
    // RW: child lists, propagation frontiers, solution vertices
    int32_t *kid_first = RW ? bv.kid_first + (size_t)q * bv.node_stride : nullptr;
    int32_t *kid_next = RW ? bv.kid_next + (size_t)q * bv.node_stride : nullptr;
    int32_t *kid_prev = RW ? bv.kid_prev + (size_t)q * bv.node_stride : nullptr;
    uint32_t *front0 = RW ? bv.frontier + (size_t)q * 2 * bv.node_stride : nullptr;
    uint32_t *front1 = RW ? front0 + bv.node_stride : nullptr;
    int32_t *vsoln = RW ? bv.vsoln + (size_t)q * bv.node_stride : nullptr;
    unsigned long long n_rewired = D->n_rewired, n_propagated = D->n_propagated;

    // Informed: constants of the ellipse transform (rrt.py:590, :621)
    const double xc0 = ((double)(D->xs[0] + D->xg[0])) / 2.0, xc1 = ((double)(D->xs[1] + D->xg[1])) / 2.0;
    const double d2sg = (double)dist2(xs, xg);
    const double C00 = D->C[0], C01 = D->C[1], C10 = D->C[2], C11 = D->C[3];
    double c_ell = 0.0;  // rrt.py:698-699, changes only when the best solution node changes
    if (informed && nsoln > 0) c_ell = cmin_soln + sqrt_u32(dist2(xg, nodes_g[vbest_soln]));

    // ---- prologue: stage the live nodes into LDS (fresh start: node 0 only) ----
    for (int k = t; k < j && k < lds_nodes; k += TPB) nodes_lds[k] = nodes_g[k];
    __syncthreads();

    uint32_t pend = 0;        // node j-1 when it was inserted by the previous iteration and may
    bool pend_valid = false;  // not be visible in LDS/HBM to the other waves yet
    double pend_cost = 0.0;
    int pend_h = 0;

    // coordinates / cost of node v as seen by this iteration
    auto node_xy = [&](uint32_t v) -> uint32_t {
        if (pend_valid && (int)v == j - 1) return pend;
        return ((int)v < lds_nodes) ? nodes_lds[v] : nodes_g[v];
    };
    auto node_h = [&](uint32_t v) -> int { return (pend_valid && (int)v == j - 1) ? pend_h : (int)heading[v]; };
    auto node_cost = [&](uint32_t v) -> double { return (pend_valid && (int)v == j - 1) ? pend_cost : vcost[v]; };

    uint32_t s_next = (i < n) ? samples[i] : 0;

    for (; i < n; ++i) {
        const int par = i & 1;
        // ---------------- sample (rrt.py:421 / :502 / :695-701) ----------------
        uint32_t xq = s_next;
        const int hq = DUB ? (int)shead[i] : 0;
        if (i + 1 < n) s_next = samples[i + 1];
        // an edge v -> sample: its length (the straight distance d2 is known from the scan) and its collision test
        auto edge_len = [&](uint32_t v, uint32_t d2) -> double {
            if (DUB) return dub_between_dev(node_xy(v), node_h(v), xq, hq, dc).len;
            return sqrt_u32(d2);
        };
        auto edge_free = [&](uint32_t v, int &cc) -> bool {
            if (DUB) {
                const uint32_t a = node_xy(v);
                const int ha = node_h(v);
                const dub_path_t pth = dub_between_dev(a, ha, xq, hq, dc);
                return dub_sweep_wave(og, dc, a, ha, xq, pth, lane, cc);
            }
            return los_wave(og, H, node_xy(v), xq, lane, cc);  // rrt.py:424 / :519
        };
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
        // `sampled` bitmap word of this cell: issue the load now, use it after the barrier.  A bit set
        // by the previous iteration may not be visible yet; that case is the pending node below.
        const uint32_t cell = (uint32_t)ux(xq) * (uint32_t)H + (uint32_t)uy(xq);
        const uint32_t bm_word = bitmap[cell >> 5];

        // ---------------- A: scan the live nodes ----------------
        uint32_t best = NONE;
        uint32_t wcnt = 0;                  // this wave's near-set fill (wave-uniform)
        const int nfull = (j - 1) / CHUNK;  // chunks that hold only committed nodes
        const int nl = nfull < lds_chunks ? nfull : lds_chunks;
        if (nl > 0) {
            u32x4 cur = nodes_lds4[t];
            for (int c = 0; c < nl; ++c) {
                u32x4 nxt = cur;
                if (c + 1 < nl) nxt = nodes_lds4[(c + 1) * TPB + t];
                if (star)
                    eval4<true>(cur, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, wl, wcnt, lane);
                else
                    eval4<false>(cur, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, wl, wcnt, lane);
                cur = nxt;
            }
        }
        if (nfull > nl) {
            u32x4 cur = nodes_g4[nl * TPB + t];
            for (int c = nl; c < nfull; ++c) {
                u32x4 nxt = cur;
                if (c + 1 < nfull) nxt = nodes_g4[(c + 1) * TPB + t];
                if (star)
                    eval4<true>(cur, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, wl, wcnt, lane);
                else
                    eval4<false>(cur, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, wl, wcnt, lane);
                cur = nxt;
            }
        }
        {  // tail chunk: masked, the pending node substituted from registers
            const int c = nfull;
            const int idx0 = c * CHUNK + 4 * t;
            uint32_t dd[4] = {NONE, NONE, NONE, NONE};
            if (idx0 < j) {
                u32x4 v;
                if (c < lds_chunks)
                    v = nodes_lds4[c * TPB + t];
                else
                    v = nodes_g4[c * TPB + t];
                uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = idx0 + e;
                    if (idx < j) {
                        uint32_t p = (pend_valid && idx == j - 1) ? pend : pv[e];
                        dd[e] = dist2(p, xq);
                        best = min(best, (dd[e] << 8) + ((uint32_t)c << 2) + (uint32_t)e);
                    }
                }
            }
            if (star)  // dd == NONE for masked elements: never below r2 (r2 <= 2^24)
                wl_append4(wl, wcnt, dd[0] < r2, dd[1] < r2, dd[2] < r2, dd[3] < r2, (uint32_t)idx0, dd[0], dd[1], dd[2], dd[3], lane);
        }
        uint32_t kd = best >> 8, ki;
        {
            const uint32_t tag = best & 0xffu;
            ki = (best == NONE) ? NONE : (tag >> 2) * (uint32_t)CHUNK + 4u * (uint32_t)t + (tag & 3u);
            wave_min_key_idx(kd, ki);
        }
        STAMP(0);

        uint32_t vbest = NONE;
        double cbest = 0.0;
        bool fast_done = false;
        if (DUB) {
            // ---------------- Dubins: a word evaluation costs ~1150 f64 operations, so nothing is evaluated speculatively or twice.
            // The waves' near-set lists are priced as ONE packed list (64 live lanes per wave instead of each wave's dozen),
            // the nearest node's word is evaluated and swept once (by the last wave, meanwhile), and candidates are tested in
            // global (cost, index) order, the two cheapest per round by two waves on two SIMDs. ----------------
            if (lane == 0) dub_slot[par * NWAVE + wave] = u32x4{kd, ki, wcnt, 0u};
            __syncthreads();  // the scan results of all waves; last iteration's node is visible
            u32x4 ds = {NONE, NONE, 0u, 0u};
            if (lane < NWAVE) ds = dub_slot[par * NWAVE + lane];
            uint32_t d2n = ds.x, vn = ds.y;
            wave_min_key_idx(d2n, vn);
            uint32_t incl = ds.z;  // prefix sum of the waves' list lengths over lanes 0..15 (one DPP row)
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
            const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, NWAVE - 1);
            const uint32_t woff = lane < NWAVE ? incl - ds.z : NONE;  // exclusive; lanes past the waves never own an entry
            if (__ballot(lane < NWAVE && ds.z > (uint32_t)WCAP) == 0) {  // (a list that overflowed its LDS part: the general path below)
                fast_done = true;
                for (uint32_t g0 = (uint32_t)wave * 64u; g0 < T; g0 += TPB) {  // (wave-uniform trip count: the bisection reads lanes 0..15 of every wave)
                    const uint32_t g = g0 + (uint32_t)lane;  // entry g of the packed list: owner = the largest wave w with woff[w] <= g
                    uint32_t lo = 0;
#pragma unroll
                    for (uint32_t bit = 8; bit != 0; bit >>= 1) {
                        const uint32_t cand = lo + bit;
                        const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(cand << 2), (int)woff);
                        lo = v <= g ? cand : lo;
                    }
                    const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lo << 2), (int)woff);
                    if (g < T) {
                        const u32x2 e = ((RRT_LDS u32x2 *)wlist_lds)[lo * WCAP + (g - base)];  // {node, d2}
                        pk_idx[g] = e.x;
                        const dub_path_t pth = dub_between_dev(node_xy(e.x), node_h(e.x), xq, hq, dc);
                        pk_cost[g] = node_cost(e.x) + pth.len;
                        double *dp = dpath + (size_t)g * 5;  // kept for the test rounds: a word is evaluated once
                        dp[0] = pth.t;
                        dp[1] = pth.p;
                        dp[2] = pth.q;
                        dp[3] = pth.len;
                        dp[4] = (double)pth.word;
                    }
                }
                STAMP(1);  // (diagnostic build: this wave's share of the pricing)
                if (wave == NWAVE - 1) {  // the nearest node: its word, the cost through it, its sweep (rrt.py:422-424)
                    const uint32_t a = node_xy(vn);
                    const int ha = node_h(vn);
                    const dub_path_t pth = dub_between_dev(a, ha, xq, hq, dc);
                    int cc = 0;
                    const bool ok = dub_sweep_wave(og, dc, a, ha, xq, pth, lane, cc);
                    if (lane == 0) {
                        dub_nn.cost = node_cost(vn) + pth.len;
                        dub_nn.ok = ok ? 1u : 0u;
                        dub_nn.cells = (uint32_t)cc;
                    }
                }
                __syncthreads();
                STAMP(2);  // (diagnostic build: the wait for the nearest node's word and sweep)
                const bool nocoll = dub_nn.ok != 0u;
                const int cells = (int)dub_nn.cells;
                const double cnear = dub_nn.cost;
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
                    pend_valid = false;
                    STAMP(3);
                    continue;
                }
                vbest = vn;
                cbest = cnear;
                if (star) {  // choose parent (rrt.py:511-521): the first entry in (cost, index) order below cnear whose sweep is free
                    sum_near += (unsigned long long)T;
                    double lbc = -1.0;
                    uint32_t lbi = 0;
                    for (int round = 0;; ++round) {
                        Top2 tt;  // this thread's share of the packed list, at or above the lower bound
                        tt.init();
                        for (uint32_t g = (uint32_t)t; g < T; g += TPB) {
                            const double c = pk_cost[g];
                            const uint32_t ix = pk_idx[g];
                            if (c < cnear && !key_lt(c, ix, lbc, lbi)) tt.fold(c, ix);
                        }
                        tt.wave_reduce();
                        BSlot bs;
                        bs.pc = tt.c1;
                        bs.pi = tt.i1;
                        bs.uc = tt.c2;
                        bs.ui = tt.i2;
                        bs.cells = bs.tested = 0;
                        if (lane == 0) bslots[(round & 1) * NWAVE + wave] = bs;
                        __syncthreads();
                        Top2 gt;  // the two cheapest of all: lane w folds wave w's two, then the wave reduces
                        gt.init();
                        if (lane < NWAVE) {
                            const BSlot r = bslots[(round & 1) * NWAVE + lane];
                            if (r.pi != NONE) gt.fold(r.pc, r.pi);
                            if (r.ui != NONE) gt.fold(r.uc, r.ui);
                        }
                        gt.wave_reduce();
                        if (gt.i1 == NONE) break;  // no entry left below cnear: the nearest stays the parent
                        if (wave < 2) {  // waves 0 and 1 (two SIMDs) test the two, both tests in flight side by side
                            const uint32_t cand = wave == 0 ? gt.i1 : gt.i2;
                            uint32_t res = 0;
                            if (cand != NONE) {
                                // its word was evaluated when the list was priced: find the entry, take the path
                                uint32_t gsel = 0;
                                for (uint32_t g0 = 0; g0 < T; g0 += 64) {
                                    const unsigned long long m = __ballot(g0 + (uint32_t)lane < T && pk_idx[g0 + (uint32_t)lane] == cand);
                                    if (m) {
                                        gsel = g0 + (uint32_t)__builtin_ctzll(m);
                                        break;
                                    }
                                }
                                const double *dp = dpath + (size_t)gsel * 5;
                                dub_path_t pth;
                                pth.t = dp[0];
                                pth.p = dp[1];
                                pth.q = dp[2];
                                pth.len = dp[3];
                                pth.word = (int32_t)dp[4];
                                int cc = 0;
                                const bool ok = dub_sweep_wave(og, dc, node_xy(cand), node_h(cand), xq, pth, lane, cc);  // rrt.py:519
                                res = (ok ? 0x80000000u : 0u) | (uint32_t)cc;
                            }
                            if (lane == 0) dub_slot[2 * NWAVE - 2 + wave].w = res;  // (the .w words of the last two slots are free)
                        }
                        __syncthreads();
                        const uint32_t r1 = dub_slot[2 * NWAVE - 2].w, r2_ = dub_slot[2 * NWAVE - 1].w;
                        n_los_cand += 1;
                        sum_cells_cand += (unsigned long long)(r1 & 0x7fffffffu);
                        if (r1 >> 31) {
                            vbest = gt.i1;
                            cbest = gt.c1;
                            break;
                        }
                        if (gt.i2 == NONE) break;
                        n_los_cand += 1;  // (the second test counts only when the first failed, as in the sequential order)
                        sum_cells_cand += (unsigned long long)(r2_ & 0x7fffffffu);
                        if (r2_ >> 31) {
                            vbest = gt.i2;
                            cbest = gt.c2;
                            break;
                        }
                        lbc = gt.c2;
                        lbi = gt.i2 + 1;
                        __syncthreads();  // (the result words are rewritten next round)
                    }
                }
            }
        }
        if (!fast_done) {
        // ---------------- A': speculative work of this wave, overlapping the other waves' scans ----------------
        // (a) local nearest: its cost and its line of sight to the sample
        uint32_t my_nn_los = 0;
        double my_vc_nn = 0.0;
        if (ki != NONE) {
            my_vc_nn = node_cost(ki);
            int cells = 0;
            bool ok;
            if (DUB) {  // one word evaluation serves the cost through the local nearest (not a function of d2) and its sweep
                const uint32_t a = node_xy(ki);
                const int ha = node_h(ki);
                const dub_path_t pth = dub_between_dev(a, ha, xq, hq, dc);
                my_vc_nn += pth.len;
                ok = dub_sweep_wave(og, dc, a, ha, xq, pth, lane, cells);
            } else {
                ok = edge_free(ki, cells);
            }
            my_nn_los = (ok ? 0x80000000u : 0u) | (uint32_t)cells;
        }
        // (b) price this wave's near-set entries, keep them in registers; best and second best
        double ec[WSLOTS];
        uint32_t ei[WSLOTS];
        Top2 top;
        top.init();
        uint32_t my_los1 = 0;
        if (star) {
#pragma unroll
            for (int s = 0; s < WSLOTS; ++s) {
                const uint32_t c = (uint32_t)lane + 64u * (uint32_t)s;
                ec[s] = f64_inf();
                ei[s] = NONE;
                if (c < wcnt && c < (uint32_t)WCAP) {
                    const u32x2 e = wl.list[c];
                    ei[s] = e.x;
                    ec[s] = node_cost(e.x) + edge_len(e.x, e.y);
                    top.fold(ec[s], ei[s]);
                }
            }
            for (uint32_t c = (uint32_t)WCAP + (uint32_t)lane; c < wcnt; c += 64) {  // overflow entries (huge radii)
                const u32x2 e = wl.spill[c - WCAP];
                top.fold(node_cost(e.x) + edge_len(e.x, e.y), e.x);
            }
            top.wave_reduce();
            if (top.i1 != NONE) {
                int cc = 0;
                const bool ok = edge_free(top.i1, cc);  // rrt.py:519, speculative
                my_los1 = (ok ? 0x80000000u : 0u) | (uint32_t)cc;
            }
        }
        if (lane == 0) {
            Slot s;
            s.d2n = kd;
            s.vn = ki;
            s.wcnt = wcnt;
            s.nn_los = my_nn_los;
            s.vc_nn = my_vc_nn;
            s.c1 = top.c1;
            s.c2 = top.c2;
            s.i1 = top.i1;
            s.i2 = top.i2;
            s.los1 = my_los1;
            s.pad[0] = s.pad[1] = s.pad[2] = 0;
            slots[par * NWAVE + wave] = s;
        }
        STAMP(1);
        __syncthreads();  // the iteration's barrier: slots and last iteration's node are visible
        STAMP(2);

        // ---------------- B: fold the 16 slots ----------------
        Slot s;
        s.d2n = NONE;
        s.vn = NONE;
        s.wcnt = 0;
        s.nn_los = 0;
        s.vc_nn = 0.0;
        s.c1 = s.c2 = f64_inf();
        s.i1 = s.i2 = NONE;
        s.los1 = 0;
        if (lane < NWAVE) s = slots[par * NWAVE + lane];
        uint32_t d2n = s.d2n, vn = s.vn;
        wave_min_key_idx(d2n, vn);
        const int lw = (int)__builtin_ctzll(__ballot(s.d2n == d2n && s.vn == vn));  // winner's lane (a wave owns a node exclusively)
        const uint32_t nn_los = (uint32_t)__builtin_amdgcn_readlane((int)s.nn_los, lw);
        const unsigned long long vcb = (unsigned long long)__double_as_longlong(s.vc_nn);
        const double vc_near = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(vcb >> 32), lw) << 32) |
                                                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)vcb, lw)));
        const bool nocoll = (nn_los >> 31) != 0;
        const int cells = (int)(nn_los & 0x7fffffffu);
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
            pend_valid = false;
            STAMP(3);
            continue;
        }

        // ---------------- C: choose parent (rrt.py:511-521) ----------------
        vbest = vn;
        cbest = DUB ? vc_near : vc_near + sqrt_u32(d2n);
        if (star) {
            const double cnear = cbest;
            sum_near += (unsigned long long)wave_sum_u32(s.wcnt);
            // round 0 was run speculatively by every wave before the barrier
            const bool has1 = s.i1 != NONE && s.c1 < cnear;  // rrt.py:518, strict
            double pc = (has1 && (s.los1 >> 31)) ? s.c1 : f64_inf();
            uint32_t pi = (has1 && (s.los1 >> 31)) ? s.i1 : NONE;
            const bool has2 = has1 && s.i2 != NONE && s.c2 < cnear;
            double uc = has2 ? s.c2 : f64_inf();
            uint32_t ui = has2 ? s.i2 : NONE;
            n_los_cand += (unsigned long long)wave_sum_u32(has1 ? 1u : 0u);
            sum_cells_cand += (unsigned long long)wave_sum_u32(has1 ? (s.los1 & 0x7fffffffu) : 0u);
            wave_min_f64_idx(pc, pi);  // B: cheapest passing entry so far
            wave_min_f64_idx(uc, ui);  // U: cheapest entry not tested yet
            // this wave's own lower bound: everything below (lbc, lbi) in its list has been tested
            double lbc = -1.0;
            uint32_t lbi = 0;
            if (top.i1 != NONE && top.c1 < cnear) {
                lbc = top.c1;
                lbi = top.i1 + 1;
            }
            int round = 0;
            while (ui != NONE && key_lt(uc, ui, pc, pi)) {  // an untested entry could still beat B: test more
                Top2 tt;
                tt.init();
#pragma unroll
                for (int k = 0; k < WSLOTS; ++k)
                    if (ei[k] != NONE && ec[k] < cnear && !key_lt(ec[k], ei[k], lbc, lbi) && key_lt(ec[k], ei[k], pc, pi)) tt.fold(ec[k], ei[k]);
                for (uint32_t c = (uint32_t)WCAP + (uint32_t)lane; c < wcnt; c += 64) {
                    const u32x2 e = wl.spill[c - WCAP];
                    const double cn = node_cost(e.x) + edge_len(e.x, e.y);
                    if (cn < cnear && !key_lt(cn, e.x, lbc, lbi) && key_lt(cn, e.x, pc, pi)) tt.fold(cn, e.x);
                }
                tt.wave_reduce();
                BSlot bs;
                bs.pc = f64_inf();
                bs.pi = NONE;
                bs.uc = tt.c2;
                bs.ui = tt.i2;
                bs.cells = 0;
                bs.tested = 0;
                if (tt.i1 != NONE) {
                    int cc = 0;
                    const bool ok = edge_free(tt.i1, cc);
                    if (ok) {
                        bs.pc = tt.c1;
                        bs.pi = tt.i1;
                    }
                    bs.cells = (uint32_t)cc;
                    bs.tested = 1;
                    lbc = tt.c1;
                    lbi = tt.i1 + 1;
                }
                if (lane == 0) bslots[(round & 1) * NWAVE + wave] = bs;
                __syncthreads();
                BSlot r;
                r.pc = r.uc = f64_inf();
                r.pi = r.ui = NONE;
                r.cells = r.tested = 0;
                if (lane < NWAVE) r = bslots[(round & 1) * NWAVE + lane];
                ++round;
                n_los_cand += (unsigned long long)wave_sum_u32(r.tested);
                sum_cells_cand += (unsigned long long)wave_sum_u32(r.cells);
                double npc = r.pc;
                uint32_t npi = r.pi;
                wave_min_f64_idx(npc, npi);
                if (key_lt(npc, npi, pc, pi)) {
                    pc = npc;
                    pi = npi;
                }
                uc = r.uc;
                ui = r.ui;
                wave_min_f64_idx(uc, ui);
            }
            if (pi != NONE) {
                vbest = pi;
                cbest = pc;
            }
        }
        }  // !fast_done
        STAMP(3);

        // ---------------- D: insert (rrt.py:524-529); rewire scan :531-546 is vacuous ----------------
        if (t == 0) {
            nodes_g[j] = xq;
            if (j < lds_nodes) nodes_lds[j] = xq;
            vcost[j] = cbest;
            parent[j] = (int32_t)vbest;
            if (DUB) heading[j] = (uint8_t)hq;
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
        if (RW) {
            // ---------------- E: the opt-in rewire ----------------
            if (t == 0) {
                rw_cnt[0] = 0;
                const int32_t f = kid_first[vbest];  // link the new node under its parent (prepend)
                kid_prev[j] = -1;
                kid_next[j] = f;
                kid_first[j] = -1;
                if (f >= 0) kid_prev[f] = j;
                kid_first[vbest] = j;
                if (informed && dist2(xq, xg) < goal_d2) vsoln[nsoln - 1] = j;
            }
            __syncthreads();  // the insertion is visible, the counter is clear
            // E1 decide: every wave goes through its own near-set entries (the list its scan filled before the insertion)
            if (star) {
                for (uint32_t base = 0; base < wcnt; base += 64) {
                    const uint32_t c = base + (uint32_t)lane;
                    u32x2 e = {NONE, 0u};
                    if (c < wcnt) e = (c < (uint32_t)WCAP) ? wl.list[c] : wl.spill[c - WCAP];
                    double cth = f64_inf();
                    bool cand = false;
                    if (c < wcnt) {
                        cth = cbest + sqrt_u32(e.y);
                        cand = cth < vcost[e.x];
                    }
                    unsigned long long m = __ballot(cand);
                    while (m) {
                        const int l = (int)__builtin_ctzll(m);
                        m &= m - 1;
                        const uint32_t vr = (uint32_t)__builtin_amdgcn_readlane((int)e.x, l);
                        int cc = 0;
                        const uint32_t pv = ((int)vr < lds_nodes) ? nodes_lds[vr] : nodes_g[vr];
                        if (los_wave(og, H, pv, xq, lane, cc) && lane == l) {
                            const uint32_t pos = __hip_atomic_fetch_add(&rw_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            front0[pos] = vr;
                            vcost[vr] = cth;  // each node is in one list only: no other lane reads or writes this cost in E1
                        }
                    }
                }
            }
            __syncthreads();
            uint32_t fcnt = rw_cnt[0];
            if (fcnt > 0) {
                n_rewired += fcnt;
                // E2 apply: re-link the nodes one after the other (two of them may share a parent or be siblings)
                if (t == 0) {
                    for (uint32_t r = 0; r < fcnt; ++r) {
                        const int32_t vr = (int32_t)front0[r];
                        const int32_t po = parent[vr], pr = kid_prev[vr], nx = kid_next[vr];
                        if (pr >= 0) kid_next[pr] = nx; else kid_first[po] = nx;
                        if (nx >= 0) kid_prev[nx] = pr;
                        const int32_t f = kid_first[j];
                        kid_prev[vr] = -1;
                        kid_next[vr] = f;
                        if (f >= 0) kid_prev[f] = vr;
                        kid_first[j] = vr;
                        parent[vr] = j;
                    }
                }
                // E3 propagate, level by level
                uint32_t *cur = front0, *nxt = front1;
                while (fcnt > 0) {
                    if (t == 0) rw_cnt[1] = 0;
                    __syncthreads();  // links / the previous level's costs are visible, the counter is clear
                    for (uint32_t f = (uint32_t)t; f < fcnt; f += TPB) {
                        const uint32_t u = cur[f];
                        const double cu = vcost[u];
                        const uint32_t xu = nodes_g[u];
                        for (int32_t c = kid_first[u]; c >= 0; c = kid_next[c]) {
                            vcost[c] = cu + sqrt_u32(dist2(nodes_g[c], xu));
                            nxt[__hip_atomic_fetch_add(&rw_cnt[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = (uint32_t)c;
                        }
                    }
                    __syncthreads();
                    fcnt = rw_cnt[1];
                    n_propagated += fcnt;
                    uint32_t *tmp = cur;
                    cur = nxt;
                    nxt = tmp;
                }
                if (informed && nsoln > 0) {  // costs moved: first minimum of the current costs over the solution vertices
                    double bc = f64_inf();
                    uint32_t bi = NONE;
                    for (int k = t; k < nsoln; k += TPB) {
                        const double c = vcost[vsoln[k]];
                        if (key_lt(c, (uint32_t)k, bc, bi)) {
                            bc = c;
                            bi = (uint32_t)k;
                        }
                    }
                    wave_min_f64_idx(bc, bi);
                    __syncthreads();  // bslots may still be read by a slower wave
                    if (lane == 0) {
                        bslots[wave].pc = bc;
                        bslots[wave].pi = bi;
                    }
                    __syncthreads();
                    double rc = f64_inf();
                    uint32_t ri = NONE;
                    if (lane < NWAVE) {
                        rc = bslots[lane].pc;
                        ri = bslots[lane].pi;
                    }
                    wave_min_f64_idx(rc, ri);
                    cmin_soln = rc;
                    vbest_soln = vsoln[ri];
                    c_ell = cmin_soln + sqrt_u32(dist2(xg, nodes_g[vbest_soln]));
                    __syncthreads();
                }
            }
            pend_valid = false;  // barriers followed the insertion: node j and its cost are visible to every wave
        } else {
            pend = xq;
            pend_cost = cbest;
            pend_h = hq;
            pend_valid = true;
        }
        j++;
        STAMP(4);
    }

    __syncthreads();  // last insert visible to every wave

    // ---------------- go2goal (rrt.py:311-332), only when the loop has finished ----------------
    // costs[k] = vcost[k] + dist(k, goal) for every live node; the goal connects to the first node in
    // stable (cost, index) order that has line of sight.  Branch and bound: every round each wave tests
    // the cheapest untested node of its stripe; done when the cheapest passing node beats every
    // untested one.
    int vgoal = 0, found = 0;
    if (status == ST_RUNNING) {
        status = ST_DONE;
        double pc;
        uint32_t pi;
        go2goal_phase<DUB>(og, H, nodes_g, vcost, 0, 1, j, xg, reinterpret_cast<uint32_t *>(spill), (RRT_LDS uint32_t *)smem, bslots, t, lane, wave, pc, pi,
                           heading, DUB ? D->hg : 0, dc);
        if (pi != NONE) {
            found = 1;
            vgoal = j;  // rrt.py:319
            if (t == 0) {
                nodes_g[j] = xg;
                vcost[j] = pc;
                parent[j] = (int32_t)pi;
                if (DUB) heading[j] = (uint8_t)D->hg;
            }
        } else {
            if (j < n) status = ST_UNREACHABLE;  // the next argsort entry is a sentinel row (rrt.py:318 faults)
            vgoal = 0;                           // rrt.py:330-331
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
        D->n_rewired = n_rewired;
        D->n_propagated = n_propagated;
#ifdef RRT_STAMPS
        for (int k = 0; k < 6; ++k) D->cyc[k] = cyc[k];
#endif
    }
}
#endif  // RRT_SERIAL_DECL_ONLY

// Arms query state in HBM: clears the `sampled` bitmap, writes node 0 (rrt.py:411-413) and fills the unfilled
// node slots with a copy of node 0 (the block kernel scans whole 4096-node steps; such a slot can never be the
// nearest node -- equal distance, higher index -- and is dropped from near sets by its index).
// (a template only so that every translation unit may see the definition: instantiated where it is launched)
template <int = 0>
__global__ void rrt_init_kernel(BatchView bv) {
    const int q = (int)blockIdx.y;
    const QDesc *D = bv.desc + q;
    if (D->status != ST_RUNNING || D->i != 0) return;
    uint32_t *bitmap = bv.bitmap + (size_t)q * bv.bitmap_words;
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < bv.bitmap_words; k += (int)(gridDim.x * blockDim.x))
        bitmap[k] = 0;
    const uint32_t n0 = pack_xy(D->xs[0], D->xs[1]);
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < bv.node_stride; k += (int)(gridDim.x * blockDim.x))
        bv.nodes[(size_t)q * bv.node_stride + k] = n0;
    if (bv.cellcnt) {  // near-set record grid: empty cells, then node 0 in its cell
        const int c0 = (D->xs[0] >> D->cell_shift) * D->ncy + (D->xs[1] >> D->cell_shift);
        for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < MAX_CELLS; k += (int)(gridDim.x * blockDim.x))
            bv.cellcnt[(size_t)q * MAX_CELLS + k] = (k == c0) ? 1u : 0u;
        if (blockIdx.x == 0 && threadIdx.x == 0)
            bv.cellrec[(size_t)q * (size_t)bv.rec_stride + (size_t)c0 * (size_t)D->cell_cap] = make_uint4(n0, 0u, 0u, 0u);  // vcost 0.0
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bv.vcost[(size_t)q * bv.node_stride] = 0.0;
        bv.parent[(size_t)q * bv.node_stride] = -1;
        if (bv.heading) bv.heading[(size_t)q * bv.node_stride] = (uint8_t)D->hs;
        if (bv.kid_first) {
            bv.kid_first[(size_t)q * bv.node_stride] = -1;
            bv.kid_next[(size_t)q * bv.node_stride] = -1;
            bv.kid_prev[(size_t)q * bv.node_stride] = -1;
        }
    }
}

}  // namespace rrtdev
