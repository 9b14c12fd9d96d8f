// rrt_pipe.h -- RRTStandard / RRTStar (rrt.py:418-437, :498-548) on ONE CU per query without a workgroup barrier in the loop:
// the many-query shape (every query of a batch on its own CU; 128 queries and more per GPU, RRT_FLAG_NOTEAM, or a launch that other
// launches left no room for teams).
//
// The pipeline of rrt_dubins_block.h with the straight edge of the reference.  Waves resolve samples side by side, each against
// the tree AS OF A SNAPSHOT (vertices [0, j_snap): the cell records carry the vertex index, younger ones are skipped); ONE wave
// retires them in order.  Results equal the sequential loop (and the block kernel's, rrt_block.h, which keeps the Informed planner
// and the teams).
//
//   resolve   (15 waves) take the next sample off a ticket counter, at most PP_WIN ahead of retirement.  One stream of the records
//             of the cells the radius ball touches: nearest vertex (near()[0], rrt.py:150-155), |within| (:176-181), and per lane
//             the entry with the smallest single-precision lower bound of cost = vcost + sqrt(d2); the 64 lane minima are priced
//             exactly (f64, sqrt_u24) and tried in (cost, index) order below the cost through the nearest vertex until one has a
//             free line of sight (choose parent, :511-521); only if some lane saw a second entry whose bound is below the best
//             cost so far does the stream run again, collecting such entries in LDS, 64 at a time.  The result goes into a ring
//             in LDS together with four 64-bit masks over the samples still in flight at the snapshot (the sample stream is an
//             input, rrt.py:240): nearer than the nearest vertex found / on the same grid cell / inside the ball / in the same
//             record cell.
//   retire    (1 wave) the one serial chain of the kernel: up to PP_KB ready samples per pass, one lane each; the records' masks
//             against the bits of the samples inserted last decide "resolve again" (a younger vertex is nearer: the retiring wave
//             does it itself, against the exact tree), "reject" (same grid cell, rrt.py:425), and which younger vertices are further
//             candidate parents (priced right there; one that wins sends its sample through the one-at-a-time path with the
//             line-of-sight tests).  Insertions are stored lane-parallel; fill counts and state are published behind their
//             acknowledgement.
//
// Measured (profiles/r03_experiments.md): done one sample at a time -- lane-parallel against a window of the last 64 samples, or by
// scalar mask operations -- the retirement cost 1 000 - 2 000 cycles per sample and decided the run time.
#pragma once

#include "rrt_block.h"

#ifndef RRT_PIPE_STREAM_DEPTH
#define RRT_PIPE_STREAM_DEPTH 4  // steps of a record stream in flight (round 3 measured 1..4 and took 3; with round 4's cheaper steps 4 is
                                 // 5 % faster at 256 queries, 5 and 6 are not: profiles/r04_experiments.md §10)
#endif

namespace rrtdev {

constexpr int PP_BUF = 128;  // collected entries per wave (pass 2): a step appends at most 64, a flush follows as soon as 64 are in
constexpr uint32_t PP_TINY = 64;  // a tree of up to this many vertices is looked at as a whole, one vertex per lane (no cell streams)
#ifndef RRT_PIPE_KB
#define RRT_PIPE_KB 32  // (16 until round 4: 1 - 3 % slower on config 4's shapes)
#endif
constexpr int PP_KB = RRT_PIPE_KB;  // most heads retired in one pass (<= 32)
constexpr int PP_WIN = 64;   // samples in flight ahead of retirement
constexpr unsigned long long PP_STALL_TICKS = 200000000ull;  // 2 s of the 100 MHz wall clock
constexpr int PP_RING = 64;  // ring of deposited samples (>= PP_WIN: the slot of sample s is written again for s + PP_RING, which is
                             // only taken once s has been retired)

// One sample as its wave resolved it against its snapshot of the tree (128 bytes, 32 words).  The four masks are over the samples in
// flight since the snapshot, bit m & 63 for sample m in [snap_i, s): what the retiring wave needs to know about them does not
// depend on whether they were inserted, so the resolving wave works it out (the sample stream is an input, rrt.py:240) and the
// retirement is mask arithmetic, one lane per sample.
struct PpRec {
    uint32_t xq;
    uint32_t nn_idx, nn_d2;  // nearest vertex of the snapshot
    uint32_t flags;          // bit 0: free line of sight from it, bit 1: the sample's cell is already in `sampled`
    uint32_t cells_nn;       // cells of that line of sight read
    uint32_t hits;           // |within| over the snapshot (RRT*)
    uint32_t vb;             // parent
    uint32_t n_los;
    double cb;               // cost through the parent
    uint32_t cells_cand;
    uint32_t snap_i;         // samples retired when it was resolved: it has seen exactly the samples before this one
    uint32_t ready;          // sample number + 1 once deposited
    uint32_t ccnt;           // records in the sample's cell as of the snapshot
    uint32_t pad0[2];
    unsigned long long nnmask;    // samples in flight nearer than the snapshot's nearest vertex
    unsigned long long dupmask;   // ... on the same grid cell
    unsigned long long rmask;     // ... within r_rewire (RRT*)
    unsigned long long cellmask;  // ... in the same cell of the record grid
    uint32_t pad1[8];
};
static_assert(sizeof(PpRec) == 128, "PpRec");
// word numbers of the fields (the retiring wave reads them by word: a pass one record per lane, a lone head one word per lane)
constexpr int PW_XQ = 0, PW_NNIDX = 1, PW_NND2 = 2, PW_FLAGS = 3, PW_CELLSNN = 4, PW_HITS = 5, PW_VB = 6, PW_NLOS = 7, PW_CB = 8, PW_CCAND = 10,
              PW_SNAPI = 11, PW_READY = 12, PW_CCNT = 13, PW_NNMASK = 16, PW_DUPMASK = 18, PW_RMASK = 20, PW_CELLMASK = 22;

struct PpLds {
    alignas(16) uint32_t cellcnt[MAX_CELLS];  // live fill counts of the cells; go2goal's two 8 KiB tables afterwards
    alignas(16) u32x4 buf[NWAVE][PP_BUF];     // pass 2: collected entries {xy, index, vcost}
    alignas(16) PpRec ring[PP_RING];
    uint32_t slots[NWAVE][64];                // the streams' cell starts of a step
    alignas(16) u32x4 win[64];                // position m & 63: the vertex sample m became {xy, index, cost} (the retiring wave's)
    alignas(16) u32x4 winb[PP_KB];            // the same for the heads of the pass being decided, by lane (their positions in `win` still
                                              // hold the samples 64 earlier, which a lower head of the pass may have to look at)
    uint32_t pendcell[64];                    // record cells of the insertions whose fill counts are not published yet
    alignas(16) BSlot bslots[2 * NWAVE];
    alignas(8) unsigned long long state;  // samples retired << 32 | vertices: ONE word, so that a snapshot is consistent
    uint32_t next;                        // ticket counter
    uint32_t pubseq;                      // odd while fill counts and state are being brought up to date (a snapshot reads both)
    uint32_t fail;                        // a wave waited PP_STALL_TICKS without any sample retiring: everybody leaves (never seen; the exit every wave reaches)
    uint32_t simd_of[NWAVE];              // which SIMD each wave runs on
    unsigned long long stat[5];
    unsigned long long dbg[8];  // diagnostic build: [0] samples resolved again, the retiring wave's cycles [1] in passes [2] passes | cycles of heads
                                // retired on their own << 24, [3] waiting for the head [4] publishing, [6] heads retired in passes, [7] publications
};

// conservative single-precision lower bound of vcost + sqrt(d2): below the f64 value by more than every rounding on the way, for
// costs up to ~1e5 cells (the same margins as the block kernel's screens, rrt_block.h)
__device__ __forceinline__ float pp_lower_bound(double V, uint32_t d2) {
    const float s = ((float)V + __builtin_amdgcn_sqrtf((float)d2)) * (1.0f - 1.0e-6f) - 4.0e-3f;
    return s > 0.0f ? s : 0.0f;
}

__global__ __launch_bounds__(TPB) void rrt_pipe_kernel(BatchView bv) {
    __shared__ PpLds L;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = (int)blockIdx.x;
    QDesc *D = bv.desc + q;
    if (D->status != ST_RUNNING) return;

    // ---- per-query views ----
    const int n = D->n;
    const bool star = D->alg >= 1;
    const uint32_t *samples = bv.samples + (size_t)q * bv.n_cap;
    uint32_t *nodes_g = bv.nodes + (size_t)q * bv.node_stride;
    double *vcost = bv.vcost + (size_t)q * bv.node_stride;
    int32_t *parent = bv.parent + (size_t)q * bv.node_stride;
    uint32_t *bitmap = bv.bitmap + (size_t)q * bv.bitmap_words;
    uint2 *spill = bv.spill + (size_t)q * bv.spill_stride;
    const bool logs = bv.nearest_log != nullptr;
    const uint8_t *og = bv.og;
    const int W = bv.W, H = bv.H;
    const uint32_t r2 = D->r2_rewire;
    const uint32_t xg = pack_xy(D->xg[0], D->xg[1]);
    const int cshift = D->cell_shift, ncx = D->ncx, ncy = D->ncy, ccap = D->cell_cap, ncells = ncx * ncy;
    u32x4 *cellrec = reinterpret_cast<u32x4 *>(bv.cellrec) + (size_t)q * (size_t)bv.rec_stride;
    uint32_t *cellcnt_g = bv.cellcnt + (size_t)q * (size_t)MAX_CELLS;
    RRT_LDS uint32_t *cellcnt = (RRT_LDS uint32_t *)L.cellcnt;
    // radius of the first record stream: the rewire radius, but at least two cells (RRTStandard has no near set, and a tiny
    // radius would leave the nearest-vertex search to the doubling below)
    int rad0 = 0;
    uint32_t rr0 = 0;  // its square: the stream deals every vertex nearer than that
    {
        const uint32_t two = (uint32_t)((2 << cshift) * (2 << cshift));
        const uint32_t rr = (star && r2 > two) ? r2 : two;
        rr0 = rr;
        rad0 = (rr >= (1u << 23)) ? 4096 : (int)sqrtf((float)(rr - 1));
        while (rad0 > 0 && (uint32_t)(rad0 * rad0) > rr - 1) --rad0;
        while ((uint32_t)((rad0 + 1) * (rad0 + 1)) <= rr - 1) ++rad0;
    }

    int i = D->i, j = D->j;
    if (t < 5) L.stat[t] = 0ull;  // statistics: added to by whoever retires (under the lock)
    if (t < 8) L.dbg[t] = 0ull;
    if (t < PP_RING) L.ring[t].ready = 0u;
    if (t == 0) {
        L.state = ((unsigned long long)(uint32_t)i << 32) | (uint32_t)j;
        L.next = (uint32_t)i;
        L.pubseq = 0u;
        L.fail = 0u;
    }
#ifdef RRT_STAMPS
    unsigned long long cyc[6] = {D->cyc[0], D->cyc[1], D->cyc[2], D->cyc[3], D->cyc[4], D->cyc[5]};
    unsigned long long tstamp = __builtin_amdgcn_s_memtime();
#endif

#ifdef RRT_STAMPS
    unsigned long long sleep_iters = 0, nslept = 0;
    bool slept_now = false;
    unsigned long long res_t0 = 0, hist_n[5] = {0, 0, 0, 0, 0}, hist_c[5] = {0, 0, 0, 0, 0};
#define DSTAMP(k)                                               \
    do {                                                        \
        unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        cyc[k] += now_ - tstamp;                                \
        tstamp = now_;                                          \
    } while (0)
#else
#define DSTAMP(k) \
    do {          \
    } while (0)
#endif
    for (int k = t; k < ncells; k += TPB) cellcnt[k] = cellcnt_g[k];
    if (lane == 0) L.simd_of[wave] = (__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | ((2 - 1) << 11)) & 3u);  // HW_REG_HW_ID bits [5:4]: SIMD_ID
    __syncthreads();

    const float FINF = __uint_as_float(0x7f800000u);
    auto cell_of = [&](uint32_t X) -> int { return (ux(X) >> cshift) * ncy + (uy(X) >> cshift); };

    // The records of the cells that the box of half-width `rad` around X touches, as ONE packed stream: lane l of a step takes
    // record 64 * step + l of the concatenation of the cells' arrays (exclusive prefix sum of the fill counts over the lanes),
    // 64 cells at a time.  f(record, live) once per step.
    // Records of vertices at or above `jsnap` (inserted after the caller's snapshot) are dealt as dead lanes.
    // `keep_d2`: every vertex at a squared distance up to this must be dealt (cells farther away than that are left out: the corners
    // of the box, a third of its records where the cells are small against the radius).
    volatile RRT_LDS uint32_t *slots = (volatile RRT_LDS uint32_t *)L.slots[wave];  // (lanes talk to each other through it: every access as written)
    auto stream_box = [&](uint32_t X, int rad, uint32_t keep_d2, uint32_t jsnap, auto &&f) {
        // a tree of up to 64 vertices: all of them in one step, from the vertex arrays instead of the cells' (the same answers; a
        // start pose that nothing can be connected to, and the first samples of every run, would otherwise walk ever larger boxes)
        const bool tiny = jsnap <= PP_TINY;
        const int x = ux(X), y = uy(X);
        const int cx0 = (x - rad < 0 ? 0 : x - rad) >> cshift, cx1 = (x + rad > W - 1 ? W - 1 : x + rad) >> cshift;
        const int cy0 = (y - rad < 0 ? 0 : y - rad) >> cshift, cy1 = (y + rad > H - 1 ? H - 1 : y + rad) >> cshift;
        const int ny = cy1 - cy0 + 1, ncr = tiny ? 1 : (cx1 - cx0 + 1) * ny;
        for (int cbase = 0; cbase < ncr; cbase += 64) {
            uint32_t tcnt = 0, toff = 0;
            if (tiny) {
                tcnt = lane == 0 ? jsnap : 0u;  // (one "cell": the vertex arrays)
            } else if (cbase + lane < ncr) {
                const int ci = cbase + lane, ccx = cx0 + ci / ny, ccy = cy0 + ci % ny, cell = ccx * ncy + ccy;
                // squared distance of the sample to the cell's rectangle
                const int xl = ccx << cshift, xh = xl + (1 << cshift) - 1, yl = ccy << cshift, yh = yl + (1 << cshift) - 1;
                const int ddx = x < xl ? xl - x : (x > xh ? x - xh : 0), ddy = y < yl ? yl - y : (y > yh ? y - yh : 0);
                const uint32_t md2 = (uint32_t)(ddx * ddx + ddy * ddy);
                tcnt = md2 <= keep_d2 ? cellcnt[cell] : 0u;
                toff = (uint32_t)cell * (uint32_t)ccap;
            }
            uint32_t incl = tcnt;
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t pre = incl - tcnt;
            // Which cell a record belongs to, without a search and without a loop over the cells: every non-empty cell whose first
            // record falls into this step writes its number into that record's slot (64 words of LDS per wave), the lanes read
            // their slots and a running maximum over the lanes (DPP) carries the number to the records behind it; the lanes in
            // front of the step's first cell start belong to the cell the last step ended in.
            int cur_c = 0;
            auto fetch = [&](uint32_t base) -> u32x4 {
                const uint32_t idx = base + (uint32_t)lane;
                slots[lane] = NONE;
                const uint32_t rel = pre - base;
                __builtin_amdgcn_wave_barrier();
                if (tcnt != 0u && rel < 64u) slots[rel] = (uint32_t)lane;
                __builtin_amdgcn_wave_barrier();
                int cv = (int)slots[lane];  // (NONE = -1)
                cv = max(cv, __builtin_amdgcn_update_dpp(DPP_SMAX_ID, cv, 0x111, 0xf, 0xf, false));
                cv = max(cv, __builtin_amdgcn_update_dpp(DPP_SMAX_ID, cv, 0x112, 0xf, 0xf, false));
                cv = max(cv, __builtin_amdgcn_update_dpp(DPP_SMAX_ID, cv, 0x114, 0xf, 0xf, false));
                cv = max(cv, __builtin_amdgcn_update_dpp(DPP_SMAX_ID, cv, 0x118, 0xf, 0xf, false));
                cv = max(cv, __builtin_amdgcn_update_dpp(DPP_SMAX_ID, cv, 0x142, 0xa, 0xf, false));
                cv = max(cv, __builtin_amdgcn_update_dpp(DPP_SMAX_ID, cv, 0x143, 0xc, 0xf, false));
                cv = cv < 0 ? cur_c : cv;
                cur_c = __builtin_amdgcn_readlane(cv, 63);
                const uint32_t cpre = (uint32_t)__builtin_amdgcn_ds_bpermute(cv << 2, (int)pre);
                const uint32_t coff = (uint32_t)__builtin_amdgcn_ds_bpermute(cv << 2, (int)toff);
                if (tiny) {
                    const uint32_t k = idx < total ? idx : 0u;
                    const unsigned long long cbits = (unsigned long long)__double_as_longlong(vcost[k]);
                    return u32x4{nodes_g[k], k, (uint32_t)cbits, (uint32_t)(cbits >> 32)};
                }
                return cellrec[idx < total ? coff + (idx - cpre) : 0u];  // {xy, index, vcost}
            };
            // RRT_PIPE_STREAM_DEPTH steps in flight: the records of the next steps are requested before this step's are looked at
            constexpr int SD = RRT_PIPE_STREAM_DEPTH;
            u32x4 rq[SD];
#pragma unroll
            for (int k = 0; k < SD; ++k) rq[k] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k + 1 < SD; ++k)
                if ((uint32_t)k * 64u < total) rq[k] = fetch((uint32_t)k * 64u);
            for (uint32_t base = 0; base < total; base += 64u) {
                const uint32_t ahead = base + (uint32_t)(SD - 1) * 64u;
                if (ahead < total) rq[SD - 1] = fetch(ahead);
                f(rq[0], base + (uint32_t)lane < total && rq[0].y < jsnap);
#pragma unroll
                for (int k = 0; k + 1 < SD; ++k) rq[k] = rq[k + 1];
            }
        }
    };

    // Test the priced entries of this wave (lane: has, cost cn through vertex idx at a) that can still become the parent,
    // cheapest first, until one is visible (rrt.py:515-521 ends at the (cost, index)-smallest visible entry below the cost
    // through the nearest vertex; `vb == NONE` while the nearest vertex, which wins every tie, is the parent).
    auto test_priced = [&](bool has, double cn, uint32_t idx, uint32_t a, uint32_t xq, double &cb, uint32_t &vb, uint32_t &nlos, uint32_t &ccells) {
        bool open = has;
        for (;;) {
            const bool better = open && (cn < cb || (cn == cb && vb != NONE && idx < vb));
            double c = better ? cn : f64_inf();
            uint32_t ix = better ? idx : NONE;
            wave_min_f64_idx(c, ix);
            if (ix == NONE) break;
            const unsigned long long m = __ballot(better && idx == ix);
            const int src = (int)__builtin_ctzll(m);
            const uint32_t pa = (uint32_t)__builtin_amdgcn_readlane((int)a, src);
            int cc = 0;
            const bool ok = los_wave(og, H, pa, xq, lane, cc);  // rrt.py:519
            nlos += 1;
            ccells += (uint32_t)cc;
            if (ok) {
                cb = c;
                vb = ix;
                break;  // every other open entry is not below this one
            }
            if (lane == src) open = false;
        }
    };

    // ---- retirement: ONE wave (the last) retires the samples in order and does nothing else: it is the one serial chain of the
    //      kernel.  Up to PP_KB heads that are ready are decided in ONE pass, lane k the head rh + k: a record's masks against the
    //      bits of the samples inserted last (bit m & 63 <-> sample m; for the lower heads of the same pass their acceptance, found
    //      by a fixed-point iteration over ballots) say whether a younger vertex is nearer than the snapshot's nearest (resolve
    //      again: the pass ends in front of that head), sits on the same grid cell (reject), or lies in the ball (one more
    //      candidate parent: priced from L.win, position m & 63 holds the vertex of sample m; a candidate that would win -- one
    //      sample in a thousand -- ends the pass in front of its head, which is then retired on its own with the line-of-sight
    //      tests).  The accepted heads are stored lane-parallel; their stores are acknowledged once per batch (s_waitcnt vmcnt(0)),
    //      then the fill counts and the state name them.
    //      Returns the number of a head that has to be resolved again against the exact tree (the retiring wave does that itself),
    //      -1 when everything is retired (or the run failed). ----
    constexpr int RW = NWAVE - 1;
    const bool retirer = wave == RW;
    int rh = i, rj = j, pub_h = i;          // next sample to retire, vertices (the unpublished ones included), samples published
    unsigned long long insbits = 0ull;      // bit m & 63: sample m (one of the last 64) was inserted
    int npend = 0;                          // insertions stored but not published: their record cells wait in L.pendcell
    unsigned long long a_j = 0, a_cnn = 0, a_near = 0, a_ccand = 0, a_los = 0;  // statistics, summed per lane and folded at the end
    const RRT_LDS uint32_t *ringw = (const RRT_LDS uint32_t *)&L.ring[0];
    volatile RRT_LDS uint32_t *pendcell = (volatile RRT_LDS uint32_t *)L.pendcell;
    RRT_LDS u32x4 *win = (RRT_LDS u32x4 *)L.win;
#ifdef RRT_STAMPS
    unsigned long long rt_mark = 0, rcyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (uniform: scalar registers)
#define RSTAMP(k)                                                  \
    do {                                                           \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        rcyc[k] += now_ - rt_mark;                                 \
        rt_mark = now_;                                            \
    } while (0)
#else
#define RSTAMP(k) \
    do {          \
    } while (0)
#endif
    auto rotl = [](unsigned long long x, int b) -> unsigned long long { return b ? (x << b) | (x >> (64 - b)) : x; };
    auto publish = [&]() {
        if (rh == pub_h) return;
        if (lane == 0) __hip_atomic_store(&L.pubseq, 2u * (uint32_t)pub_h + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (npend > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane < npend) __hip_atomic_fetch_add(&cellcnt[pendcell[lane]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (lane == 0) {
            __hip_atomic_store(&L.state, ((unsigned long long)(uint32_t)rh << 32) | (uint32_t)rj, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&L.pubseq, 2u * (uint32_t)rh, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        npend = 0;
        pub_h = rh;
#ifdef RRT_STAMPS
        rcyc[7] += 1;
#endif
    };
    // rrt.py:524-529 for the lanes with `acc` (their sample s, vertex number v, position `pos` in record cell c); `rank`: how many
    // lanes below this one insert too
    auto store_vertex = [&](bool acc, int s, uint32_t xq, double cb, uint32_t vb, uint32_t v, int c, uint32_t pos, int rank) {
        if (acc) {
            nodes_g[v] = xq;
            vcost[v] = cb;
            parent[v] = (int32_t)vb;
            const uint32_t cellb = (uint32_t)ux(xq) * (uint32_t)H + (uint32_t)uy(xq);
            atomicOr(&bitmap[cellb >> 5], 1u << (cellb & 31));  // rrt.py:426
            const unsigned long long cbits = (unsigned long long)__double_as_longlong(cb);
            const u32x4 rc = u32x4{xq, v, (uint32_t)cbits, (uint32_t)(cbits >> 32)};
            cellrec[(size_t)c * (size_t)ccap + pos] = rc;
            win[s & 63] = rc;  // (what younger samples in flight are checked against)
            pendcell[npend + rank] = (uint32_t)c;
        }
    };
    auto rl = [&](uint32_t v, int k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, k); };
    auto rl64 = [&](uint32_t v, int k) -> unsigned long long { return ((unsigned long long)rl(v, k + 1) << 32) | rl(v, k); };
    // one head on its own (ready): the capacity's end, and a head with a younger vertex that is cheaper than its parent
    auto retire_one = [&]() -> int {
        const int h = rh;
        const uint32_t rec = ringw[(uint32_t)(h & (PP_RING - 1)) * 32u + (uint32_t)(lane & 31)];
        const uint32_t xq = rl(rec, PW_XQ), flags = rl(rec, PW_FLAGS);
        if ((rl64(rec, PW_NNMASK) & insbits) != 0ull) return h;  // a younger vertex is nearer: resolve again
        const bool pre_ok = (flags & 3u) == 1u;  // visible from the nearest vertex, cell not sampled before its snapshot
        const bool dup = (rl64(rec, PW_DUPMASK) & insbits) != 0ull;
        const unsigned long long inball = star ? (rl64(rec, PW_RMASK) & insbits) : 0ull;
        uint32_t vb = rl(rec, PW_VB);
        double cb = __longlong_as_double((long long)rl64(rec, PW_CB));
        uint32_t add_los = 0, add_cells = 0;
        if (pre_ok && !dup && inball != 0ull) {
            // A younger vertex inside the ball is one more candidate parent (rrt.py:515-521 walks it too): the ones below the chosen
            // cost are tested, cheapest first (a younger vertex loses every tie against the snapshot's choice: higher index)
            const bool mine_in = ((inball >> lane) & 1ull) != 0ull;
            const u32x4 e = win[lane];
            const double wcn = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z)) + sqrt_u24(dist2(e.x, xq));
            const bool cnd = mine_in && wcn < cb;
            if (__ballot(cnd) != 0ull) test_priced(cnd, wcn, e.y, e.x, xq, cb, vb, add_los, add_cells);
        }
        const bool acc = pre_ok && !dup && rj != n;  // rrt.py:425
        if (logs && lane == 0) {
            bv.nearest_log[(size_t)q * bv.n_cap + h] = (int32_t)rl(rec, PW_NNIDX);
            bv.accept_log[(size_t)q * bv.n_cap + h] = (uint8_t)acc;
            bv.cbest_log[(size_t)q * bv.n_cap + h] = __longlong_as_double(0x7ff8000000000000ll);
            bv.j_log[(size_t)q * bv.n_cap + h] = rj;
        }
        const unsigned long long hbit = 1ull << (h & 63);
        if (lane == 0) {
            a_j += (unsigned long long)rj;
            a_cnn += (unsigned long long)rl(rec, PW_CELLSNN);
            if (acc && star) {
                a_near += (unsigned long long)rl(rec, PW_HITS) + (unsigned long long)__builtin_popcountll(inball);
                a_los += (unsigned long long)(rl(rec, PW_NLOS) + add_los);
                a_ccand += (unsigned long long)(rl(rec, PW_CCAND) + add_cells);
            }
        }
        const int c = cell_of(xq);
        const uint32_t pos = rl(rec, PW_CCNT) + (uint32_t)__builtin_popcountll(rl64(rec, PW_CELLMASK) & insbits);
        store_vertex(acc && lane == 0, h, xq, cb, vb, (uint32_t)rj, c, pos, 0);
        insbits = acc ? (insbits | hbit) : (insbits & ~hbit);
        npend += acc ? 1 : 0;
        rj += acc ? 1 : 0;
        rh = h + 1;
        return -1;
    };
    auto retire = [&]() -> int {
#ifdef RRT_STAMPS
        rt_mark = __builtin_amdgcn_s_memtime();
#endif
        for (;;) {
#ifndef RRT_PIPE_PUB
#define RRT_PIPE_PUB 16  // samples retired between publications at most (a publication waits for the stores' acknowledgement)
#endif
            if (npend >= RRT_PIPE_PUB || rh - pub_h >= RRT_PIPE_PUB || rh >= n) {
                publish();
                RSTAMP(4);
                if (rh >= n) return -1;
            }
            // ---- the heads that are ready: lane k <-> sample rh + k ----
            const int s = rh + lane;
            const bool mayb = lane < PP_KB && s < n;
            const RRT_LDS uint32_t *rw = ringw + (uint32_t)(s & (PP_RING - 1)) * 32u;
            const uint32_t rdy = mayb ? __hip_atomic_load(rw + PW_READY, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
            const unsigned long long rdm = __ballot(mayb && rdy == (uint32_t)s + 1u);
            const int K = (int)__builtin_ctzll(~rdm);  // (consecutive from the head)
            if (K == 0) {  // nothing to do: what is retired becomes visible; bounded waiting for the head
                publish();
                RSTAMP(4);
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    const uint32_t r0 = __hip_atomic_load(ringw + (uint32_t)(rh & (PP_RING - 1)) * 32u + PW_READY, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (r0 == (uint32_t)rh + 1u) break;
                    if (wall_clock64() - t0 > PP_STALL_TICKS) {  // (never seen; everything retired so far is published above)
                        if (lane == 0) __hip_atomic_store(&L.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        return -1;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                RSTAMP(3);
                continue;
            }
            if (rj + K >= n) {  // the vertex array is about to be full (rrt.py:425, j != n): one head at a time
                const int r1 = retire_one();
                RSTAMP(2);
                if (r1 >= 0) {
                    publish();
                    RSTAMP(4);
                    return r1;
                }
                continue;
            }
            // ---- the records, one per lane ----
            const bool in = lane < K;
            uint32_t xq = 0, flags = 0, vb = 0, ccnt = 0, hits = 0, nlos = 0, ccand = 0, cnn = 0, nnidx = 0;
            unsigned long long nnm = 0, dupm = 0, rmk = 0, cmk = 0;
            double cb = 0.0;
            if (in) {
                xq = rw[PW_XQ];
                flags = rw[PW_FLAGS];
                vb = rw[PW_VB];
                ccnt = rw[PW_CCNT];
                hits = rw[PW_HITS];
                nlos = rw[PW_NLOS];
                ccand = rw[PW_CCAND];
                cnn = rw[PW_CELLSNN];
                if (logs) nnidx = rw[PW_NNIDX];
                nnm = ((unsigned long long)rw[PW_NNMASK + 1] << 32) | rw[PW_NNMASK];
                dupm = ((unsigned long long)rw[PW_DUPMASK + 1] << 32) | rw[PW_DUPMASK];
                rmk = ((unsigned long long)rw[PW_RMASK + 1] << 32) | rw[PW_RMASK];
                cmk = ((unsigned long long)rw[PW_CELLMASK + 1] << 32) | rw[PW_CELLMASK];
                cb = __longlong_as_double((long long)(((unsigned long long)rw[PW_CB + 1] << 32) | rw[PW_CB]));
            }
            const int base = rh & 63;
            const unsigned long long lowk = (1ull << lane) - 1ull;               // the lanes below this one
            const unsigned long long LP = rotl(lowk, base);                       // ... as sample bits: the lower heads of this pass
            // the samples before this pass, as this head sees them (the positions of the HIGHER heads of the pass still hold the
            // samples 64 earlier, which a full window reaches back to)
            const unsigned long long old = insbits & ~LP;
            const bool pre_ok = in && (flags & 3u) == 1u;  // visible from the nearest vertex, cell not sampled before its snapshot
            // which heads are inserted (rrt.py:425): a head is not if an inserted younger sample sits on its cell -- for the lower
            // heads of the pass that is what is being decided, so iterate (lane k is final after k rounds; one or two in practice)
            unsigned long long A_l = __ballot(pre_ok && (dupm & old) == 0ull);
            for (;;) {
                const unsigned long long A_p = rotl(A_l, base);
                const unsigned long long A2 = __ballot(pre_ok && (dupm & (old | (A_p & LP))) == 0ull);
                if (A2 == A_l) break;
                A_l = A2;
            }
            const unsigned long long young = old | (rotl(A_l, base) & LP);  // per lane: the inserted samples this head has to look at
            // a younger vertex nearer than the snapshot's nearest (it loses ties: higher index): the pass ends in front of that head
            const unsigned long long redo_l = __ballot(in && (nnm & young) != 0ull);
            int Kc = redo_l != 0ull ? (int)__builtin_ctzll(redo_l) : K;
            bool redo_cut = redo_l != 0ull;
            const uint32_t vk = (uint32_t)rj + (uint32_t)__builtin_popcountll(A_l & lowk);  // the vertex number of this head (if inserted)
            bool acc = ((A_l >> lane) & 1ull) != 0ull && lane < Kc;
            const int c = cell_of(xq);
            const uint32_t pos = ccnt + (uint32_t)__builtin_popcountll(cmk & young);
            const unsigned long long accl0 = __ballot(acc);
            if (acc) {  // (the higher heads of the pass price it as a candidate parent below)
                const unsigned long long cbits = (unsigned long long)__double_as_longlong(cb);
                L.winb[lane] = u32x4{xq, vk, (uint32_t)cbits, (uint32_t)(cbits >> 32)};
            }
            // younger vertices inside the ball are further candidate parents (rrt.py:515-521 walks them too): one that is cheaper
            // than the chosen parent needs its line of sight tested -- rare; the pass then ends in front of that head
            unsigned long long rc = (acc && star) ? (rmk & young) : 0ull;
            const uint32_t nball = (uint32_t)__builtin_popcountll(rc);
            unsigned long long hitl = 0ull;
            while (__ballot(rc != 0ull) != 0ull) {
                bool hit = false;
                if (rc != 0ull) {
                    const int pp = (int)__builtin_ctzll(rc);
                    rc &= rc - 1ull;
                    const u32x4 e = ((LP >> pp) & 1ull) != 0ull ? L.winb[(pp - base) & 63] : win[pp];  // a lower head of this pass / an older sample
                    const double wcn = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z)) + sqrt_u24(dist2(e.x, xq));
                    hit = wcn < cb;
                }
                hitl |= __ballot(hit);
            }
            bool slow_cut = false;
            if (hitl != 0ull && (int)__builtin_ctzll(hitl) < Kc) {
                Kc = (int)__builtin_ctzll(hitl);
                slow_cut = true;
                redo_cut = false;
            }
            acc = acc && lane < Kc;
            const unsigned long long accl = accl0 & ((1ull << Kc) - 1ull);
            const int nacc = __builtin_popcountll(accl);
            store_vertex(acc, s, xq, cb, vb, vk, c, pos, __builtin_popcountll(accl & lowk));
            if (lane < Kc) {
                a_j += (unsigned long long)vk;
                a_cnn += (unsigned long long)cnn;
                if (logs) {
                    bv.nearest_log[(size_t)q * bv.n_cap + s] = (int32_t)nnidx;
                    bv.accept_log[(size_t)q * bv.n_cap + s] = (uint8_t)acc;
                    bv.cbest_log[(size_t)q * bv.n_cap + s] = __longlong_as_double(0x7ff8000000000000ll);
                    bv.j_log[(size_t)q * bv.n_cap + s] = (int32_t)vk;
                }
            }
            if (acc && star) {
                a_near += (unsigned long long)hits + (unsigned long long)nball;
                a_los += (unsigned long long)nlos;
                a_ccand += (unsigned long long)ccand;
            }
            insbits = (insbits & ~rotl((1ull << Kc) - 1ull, base)) | rotl(accl, base);
            npend += nacc;
            rj += nacc;
            rh += Kc;
#ifdef RRT_STAMPS
            rcyc[6] += (unsigned long long)Kc;
            rcyc[0] += 1;
#endif
            RSTAMP(1);
            if (redo_cut) {
#ifdef RRT_STAMPS
                if (lane == 0) L.dbg[0] += 1;
#endif
                publish();  // the exact tree: everything retired is visible to the snapshot it takes
                RSTAMP(4);
                return rh;
            }
            if (slow_cut) {
                const int r1 = retire_one();
                RSTAMP(2);
                if (r1 >= 0) {
                    publish();
                    RSTAMP(4);
                    return r1;
                }
            }
        }
    };
    if (retirer) __builtin_amdgcn_s_setprio(3);
    // RRT_PIPE_MATES (experiment, profiles/r03_experiments.md): how many resolving waves share the retiring wave's SIMD -- the others
    // that landed there only join the barriers.  While the retirement cost ~2 000 cycles per sample, a SIMD of its own (0) paid for
    // the three waves given up; with the passes it does not (3: every wave works).
#ifndef RRT_PIPE_MATES
#define RRT_PIPE_MATES 3
#endif
    bool idle = false;
    if (RRT_PIPE_MATES < 3 && !retirer && L.simd_of[wave] == L.simd_of[RW]) {
        int before = 0;  // waves of that SIMD with a lower number
        for (int w = 0; w < wave; ++w) before += (L.simd_of[w] == L.simd_of[RW]) ? 1 : 0;
        idle = before >= RRT_PIPE_MATES;
    }

    int wait_done = -1;  // bounded waiting of a resolving wave: the retired count when it began to wait, and when
    unsigned long long wait_t0 = 0;
    for (; !idle;) {
        int s;
        if (retirer) {
            s = retire();
            if (s < 0) break;
        } else {
            uint32_t tk = 0;
            if (lane == 0) tk = __hip_atomic_fetch_add(&L.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            s = __builtin_amdgcn_readfirstlane((int)tk);
            if (s >= n) break;
            bool failed = false;
            for (;;) {  // not more than PP_WIN samples ahead of retirement
                const int done = (int)(__hip_atomic_load(&L.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32);
                if (s - done < PP_WIN) break;
                if (__hip_atomic_load(&L.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
                    failed = true;
                    break;
                }
                if (done != wait_done) {
                    wait_done = done;
                    wait_t0 = wall_clock64();
                } else if (wall_clock64() - wait_t0 > PP_STALL_TICKS) {
                    if (lane == 0) __hip_atomic_store(&L.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    failed = true;
                    break;
                }
#ifdef RRT_STAMPS
                sleep_iters += 1;
                slept_now = true;
#endif
                __builtin_amdgcn_s_sleep(16);
            }
#ifdef RRT_STAMPS
            nslept += slept_now ? 1 : 0;
            slept_now = false;
#endif
            DSTAMP(4);  // (diagnostic build, wave 0) waiting
#ifdef RRT_STAMPS
            res_t0 = tstamp;
#endif
            if (failed) break;
        }
        // =============================== resolve sample s against a snapshot ===============================
        {
            const uint32_t xq = samples[s];
            // the sample before this one whose number is this lane's modulo 64 (the retiring wave keeps the samples in flight by that bit)
            const int m_l = s - 1 - ((s - 1 - lane) & 63);
            const uint32_t xm = samples[m_l < 0 ? 0 : m_l];
            const int own_cell = cell_of(xq);
            // the snapshot: state and the fill count of the sample's own cell, both as of the same publication
            unsigned long long snap;
            uint32_t ccnt_snap;
            for (;;) {
                const uint32_t q1 = __hip_atomic_load(&L.pubseq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                snap = __hip_atomic_load(&L.state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                ccnt_snap = __hip_atomic_load(&cellcnt[own_cell], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t q2 = __hip_atomic_load(&L.pubseq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((q1 & 1u) == 0u && q1 == q2) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const uint32_t snap_i = (uint32_t)(snap >> 32), jsnap = (uint32_t)snap;
            const uint32_t cell = (uint32_t)ux(xq) * (uint32_t)H + (uint32_t)uy(xq);
            const uint32_t bm_word = __hip_atomic_load(bitmap + (cell >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (set by an L2 atomic: read it there)
            // ---- pass 1 of the record stream: nearest record of the box, |within|, per lane the entry with the smallest bound ----
            uint32_t hits = 0;
            uint32_t ld2 = NONE, lidx = NONE, lxy = 0, lvl = 0, lvh = 0;  // this lane's nearest record
            float m1f = FINF, m2f = FINF;                                 // smallest / second smallest bound among this lane's hits
            uint32_t m1idx = NONE, m1xy = 0, m1vl = 0, m1vh = 0;
            stream_box(xq, rad0, rr0 - 1u, jsnap, [&](const u32x4 rc, bool live) {
                const uint32_t d2 = live ? dist2(rc.x, xq) : NONE;
                const bool nearer = d2 < ld2 || (d2 == ld2 && live && rc.y < lidx);
                ld2 = nearer ? d2 : ld2;
                lidx = nearer ? rc.y : lidx;
                lxy = nearer ? rc.x : lxy;
                lvl = nearer ? rc.z : lvl;
                lvh = nearer ? rc.w : lvh;
                if (!star) return;
                const bool hit = d2 < r2;  // within(), rrt.py:176-181 (d2 == NONE for a dead lane: never below r2 <= 2^24)
                hits += hit ? 1u : 0u;
                const double V = __longlong_as_double((long long)(((unsigned long long)rc.w << 32) | rc.z));
                const float lb = hit ? pp_lower_bound(V, d2) : FINF;
                const bool first = lb < m1f || (lb == m1f && hit && rc.y < m1idx);
                m2f = first ? m1f : __builtin_fminf(m2f, lb);
                m1f = first ? lb : m1f;
                m1idx = first ? rc.y : m1idx;
                m1xy = first ? rc.x : m1xy;
                m1vl = first ? rc.z : m1vl;
                m1vh = first ? rc.w : m1vh;
            });
            uint32_t nn_d2 = ld2, nn_idx = lidx;
            wave_min_key_idx(nn_d2, nn_idx);
            // nothing in the box, or something that a vertex outside the box could beat: the box once more at twice the size, then
            // (a sample far from the tree: a region the tree has not reached, or cannot) every vertex in turn -- 4 bytes and six
            // instructions per vertex, where ever larger boxes would deal out every record of the map
            int radn = rad0;
            bool far = jsnap > PP_TINY && (nn_d2 == NONE || nn_d2 > (uint32_t)radn * (uint32_t)radn) && radn < (W > H ? W : H);
            if (far) {
                radn = 2 * radn + 1;
                ld2 = NONE;
                lidx = NONE;
                stream_box(xq, radn, (uint32_t)radn * (uint32_t)radn, jsnap, [&](const u32x4 rc, bool live) {
                    const uint32_t d2 = live ? dist2(rc.x, xq) : NONE;
                    const bool nearer = d2 < ld2 || (d2 == ld2 && live && rc.y < lidx);
                    ld2 = nearer ? d2 : ld2;
                    lidx = nearer ? rc.y : lidx;
                    lxy = nearer ? rc.x : lxy;
                    lvl = nearer ? rc.z : lvl;
                    lvh = nearer ? rc.w : lvh;
                });
                nn_d2 = ld2;
                nn_idx = lidx;
                wave_min_key_idx(nn_d2, nn_idx);
                far = (nn_d2 == NONE || nn_d2 > (uint32_t)radn * (uint32_t)radn) && radn < (W > H ? W : H);
            }
            if (far) {
                ld2 = NONE;
                lidx = NONE;
                for (uint32_t b0 = 0; b0 < jsnap; b0 += 256u) {
                    uint32_t xy4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t k = b0 + 64u * (uint32_t)u + (uint32_t)lane;
                        xy4[u] = nodes_g[k < jsnap ? k : 0u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {  // (a lane meets its vertices in index order: strict < keeps the lowest index)
                        const uint32_t k = b0 + 64u * (uint32_t)u + (uint32_t)lane;
                        const uint32_t d2 = k < jsnap ? dist2(xy4[u], xq) : NONE;
                        const bool nearer = d2 < ld2;
                        ld2 = nearer ? d2 : ld2;
                        lidx = nearer ? k : lidx;
                        lxy = nearer ? xy4[u] : lxy;
                    }
                }
                const unsigned long long cbits = (unsigned long long)__double_as_longlong(vcost[lidx != NONE ? lidx : 0u]);
                lvl = (uint32_t)cbits;
                lvh = (uint32_t)(cbits >> 32);
                nn_d2 = ld2;
                nn_idx = lidx;
                wave_min_key_idx(nn_d2, nn_idx);
            }
            // the nearest vertex's record, uniform
            uint32_t nn_xy, nn_vl, nn_vh;
            {
                const unsigned long long m = __ballot(lidx == nn_idx && ld2 == nn_d2);
                const int src = (int)__builtin_ctzll(m);
                nn_xy = (uint32_t)__shfl((int)lxy, src);
                nn_vl = (uint32_t)__shfl((int)lvl, src);
                nn_vh = (uint32_t)__shfl((int)lvh, src);
            }
            const uint32_t nhits = star ? wave_sum_u32(hits) : 0u;
            DSTAMP(0);  // (diagnostic build, wave 0) the first record stream
            // ---- the lane that prices the nearest vertex: the one whose own entry it is, else one without an entry, else the one
            //      whose entry has the largest bound (that entry is left to pass 2) ----
            int slot;
            {
                const unsigned long long own = __ballot(m1idx == nn_idx);
                const unsigned long long none = __ballot(m1idx == NONE);
                if (own) slot = (int)__builtin_ctzll(own);
                else if (none) slot = (int)__builtin_ctzll(none);
                else {
                    const uint32_t inv = ~__float_as_uint(m1f);  // bounds are non-negative floats: the largest has the smallest complement
                    const uint32_t mx = wave_min_u32(inv);
                    slot = (int)__builtin_ctzll(__ballot(inv == mx));
                }
            }
            float left = m2f;  // this lane's smallest bound among the entries it saw but does not price in pass 1
            if (lane == slot && m1idx != nn_idx && m1idx != NONE) left = __builtin_fminf(left, m1f);
            uint32_t e_idx = (lane == slot) ? nn_idx : m1idx;  // the vertex this lane prices (NONE: none)
            const uint32_t e_xy = (lane == slot) ? nn_xy : m1xy;
            const double e_V = __longlong_as_double((long long)(((unsigned long long)((lane == slot) ? nn_vh : m1vh) << 32) | ((lane == slot) ? nn_vl : m1vl)));
            // ---- one exact price per lane ----
            const double e_cn = e_idx != NONE ? e_V + sqrt_u24(dist2(e_xy, xq)) : f64_inf();
            DSTAMP(1);  // one price per lane
            // ---- nearest vertex: cost through it, its line of sight (rrt.py:422-425) ----
            const double c_nn = __shfl(e_cn, slot);
            int cells = 0;
            const bool nocoll = los_wave(og, H, nn_xy, xq, lane, cells);
            const bool dup = ((bm_word >> (cell & 31)) & 1u) != 0u;
            double cb = c_nn;
            uint32_t vb = NONE, nlos = 0, ccells = 0;
            if (star && nocoll && !dup) {
                // ---- choose parent: the priced entries, then whatever pass 1 left unpriced below the best cost so far ----
                test_priced(e_idx != NONE && lane != slot, e_cn, e_idx, e_xy, xq, cb, vb, nlos, ccells);
                DSTAMP(2);  // the nearest vertex's line of sight, those of the priced entries
                if (__ballot((double)left < cb) != 0ull) {
                    RRT_LDS u32x4 *buf = (RRT_LDS u32x4 *)L.buf[wave];
                    uint32_t nbuf = 0;
                    const uint32_t skip = (lane == slot) ? NONE : m1idx;  // this lane's entry of pass 1 (the stream deals the same records to the same lanes)
                    auto flush = [&]() {
                        u32x4 e = {0u, NONE, 0u, 0u};
                        if ((uint32_t)lane < nbuf) e = buf[lane];
                        const double V = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                        const uint32_t fd2 = dist2(e.x, xq);
                        const bool has = e.y != NONE && (double)pp_lower_bound(V, fd2) < cb;  // (the best cost may have fallen since the entry was collected)
                        const double fcn = has ? V + sqrt_u24(fd2) : f64_inf();
                        test_priced(has, fcn, e.y, e.x, xq, cb, vb, nlos, ccells);
                        // drop the 64 entries just handled
                        u32x4 mv = {0u, NONE, 0u, 0u};
                        const bool tail = (uint32_t)lane + 64u < nbuf;
                        if (tail) mv = buf[lane + 64];
                        if (tail) buf[lane] = mv;
                        nbuf = nbuf > 64u ? nbuf - 64u : 0u;
                    };
                    stream_box(xq, rad0, rr0 - 1u, jsnap, [&](const u32x4 rc, bool live) {
                        const uint32_t d2 = live ? dist2(rc.x, xq) : NONE;
                        const double V = __longlong_as_double((long long)(((unsigned long long)rc.w << 32) | rc.z));
                        const bool take = d2 < r2 && rc.y != nn_idx && rc.y != skip && (double)pp_lower_bound(V, d2) < cb;
                        const unsigned long long tm = __ballot(take);
                        if (tm == 0ull) return;
                        if (take) buf[nbuf + (uint32_t)__builtin_popcountll(tm & ((1ull << lane) - 1ull))] = rc;
                        nbuf += (uint32_t)__builtin_popcountll(tm);
                        if (nbuf >= 64u) flush();
                    });
                    while (nbuf > 0u) flush();
                    DSTAMP(3);  // pass 2: second stream, its prices and lines of sight
                }
            }
            DSTAMP(2);
            {
                // against the samples in flight since the snapshot (bit m & 63 for sample m in [snap_i, s)): rrt.py:150-155 (a nearer
                // vertex), :425 (the same grid cell), :176-181 (inside the ball), and the record cell for the insert position
                const bool fl = m_l >= (int)snap_i;
                const uint32_t dm = dist2(xm, xq);
                const unsigned long long nnmask = __ballot(fl && dm < nn_d2), dupmask = __ballot(fl && xm == xq),
                                         rmask = __ballot(fl && star && dm < r2), cellmask = __ballot(fl && cell_of(xm) == own_cell);
                if (lane == 0) {
                    RRT_LDS u32x4 *slot = (RRT_LDS u32x4 *)&L.ring[s & (PP_RING - 1)];
                    const unsigned long long cbits = (unsigned long long)__double_as_longlong(cb);
                    slot[0] = u32x4{xq, nn_idx, nn_d2, (nocoll ? 1u : 0u) | (dup ? 2u : 0u)};
                    slot[1] = u32x4{(uint32_t)cells, nhits, vb == NONE ? nn_idx : vb, nlos};
                    slot[2] = u32x4{(uint32_t)cbits, (uint32_t)(cbits >> 32), ccells, snap_i};
                    slot[4] = u32x4{(uint32_t)nnmask, (uint32_t)(nnmask >> 32), (uint32_t)dupmask, (uint32_t)(dupmask >> 32)};
                    slot[5] = u32x4{(uint32_t)rmask, (uint32_t)(rmask >> 32), (uint32_t)cellmask, (uint32_t)(cellmask >> 32)};
                    slot[3] = u32x4{0u, ccnt_snap, 0u, 0u};
                    __hip_atomic_store(&L.ring[s & (PP_RING - 1)].ready, (uint32_t)s + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        DSTAMP(5);  // deposit
#ifdef RRT_STAMPS
        {  // how long this sample took its wave (wave 0's, and the retiring wave's resolutions): a slow head holds everybody up
            const unsigned long long dt = tstamp - res_t0;
            const int bk = dt < 20000ull ? 0 : dt < 40000ull ? 1 : dt < 80000ull ? 2 : dt < 160000ull ? 3 : 4;
            hist_n[bk] += 1;
            hist_c[bk] += dt;
        }
#endif
    }
    if (retirer) {
        __builtin_amdgcn_s_setprio(0);
#ifdef RRT_STAMPS
        if (lane == 0)
            for (int k = 1; k < 8; ++k) L.dbg[k] = rcyc[k];
        if (lane == 0) L.dbg[2] = rcyc[0] | (rcyc[2] << 24);
#endif
        atomicAdd(&L.stat[0], a_j);
        atomicAdd(&L.stat[1], a_cnn);
        atomicAdd(&L.stat[2], a_near);
        atomicAdd(&L.stat[3], a_ccand);
        atomicAdd(&L.stat[4], a_los);
    }
    __syncthreads();
    {
        const unsigned long long st = __hip_atomic_load(&L.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        i = (int)(st >> 32);
        j = (int)(uint32_t)st;
    }

    for (int k = t; k < ncells; k += TPB) cellcnt_g[k] = cellcnt[k];
    __syncthreads();
    unsigned long long sum_j = D->sum_j, sum_cells_nn = D->sum_cells_nn, sum_near = D->sum_near, sum_cells_cand = D->sum_cells_cand,
                       n_los_cand = D->n_los_cand;
    sum_j += L.stat[0];
    sum_cells_nn += L.stat[1];
    sum_near += L.stat[2];
    sum_cells_cand += L.stat[3];
    n_los_cand += L.stat[4];
    __syncthreads();  // (go2goal reuses the LDS)

    // ---------------- go2goal (rrt.py:311-332) ----------------
    int status = ST_DONE, vgoal = 0, found = 0;
    if (L.fail != 0u || i < n) {
        status = ST_TEAM_FAIL;  // (reported as RRT_E_HIP; the tree is consistent up to sample i)
    } else {
        double pc;
        uint32_t pi;
        go2goal_phase<false>(og, H, nodes_g, vcost, 0, 1, j, xg, reinterpret_cast<uint32_t *>(spill), (RRT_LDS uint32_t *)L.cellcnt, L.bslots, t, lane, wave, pc, pi);
        if (pi != NONE) {
            found = 1;
            vgoal = j;
            if (t == 0) {
                nodes_g[j] = xg;
                vcost[j] = pc;
                parent[j] = (int32_t)pi;
            }
        } else {
            if (j < n) status = ST_UNREACHABLE;
            vgoal = 0;
        }
    }
    if (t == 0) {
        D->status = status;
        D->i = i;
        D->j = j;
        D->vgoal = vgoal;
        D->found = found;
        D->sum_j = sum_j;
        D->sum_cells_nn = sum_cells_nn;
        D->sum_near = sum_near;
        D->sum_cells_cand = sum_cells_cand;
        D->n_los_cand = n_los_cand;
#ifdef RRT_STAMPS
        for (int k = 0; k < 6; ++k) D->cyc[k] = cyc[k];
        for (int k = 0; k < 8; ++k) D->wcyc[k] = L.dbg[k];
        for (int k = 0; k < 5; ++k) {
            D->wcyc[8 + k] = hist_n[k];
            D->wcyc[13 + k] = hist_c[k];
        }
        D->wcyc[18] = nslept;
        D->wcyc[19] = sleep_iters;
#endif
    }
}

#undef DSTAMP

}  // namespace rrtdev
