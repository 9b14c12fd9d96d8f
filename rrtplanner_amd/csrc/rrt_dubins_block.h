// rrt_dubins_block.h -- Dubins-RRT / Dubins-RRT* (BASELINE.json configs[4]; no reference counterpart: include/rrt_dubins.h,
// oracle/dubins_oracle.c) on one CU per query, 16 samples per round.
//
// The loop is the reference's (rrt.py:418-437 / :498-548) with the straight edge replaced by the shortest Dubins word: nearest
// vertex on (x, y), sweep of the word nearest -> sample, accept test, choose parent over the radius ball, insert.  What is
// expensive here is the word itself (~1150 dependent f64 operations), so the kernel is built around evaluating few of them:
//
//   pipeline  the sample stream does not depend on the tree, so the 16 waves of the workgroup each take the next sample off a
//             ticket counter and resolve it on their own against the tree AS OF A SNAPSHOT (vertices [0, j_snap): the records
//             carry the vertex index, younger ones are skipped), deposit the result in a ring in LDS and go on with the next
//             ticket, at most DB_WIN samples ahead of retirement.  Samples RETIRE in order (whichever wave finds the head of
//             the ring ready takes a lock and retires as far as it can): a sample is checked against the samples inserted since
//             its snapshot -- one lane each -- and inserted or rejected if none of them can have influenced it; otherwise the
//             retiring wave resolves it again, now against the exact tree (it is the head: nothing is in flight before it).
//             No workgroup barrier in the loop: a sample that needs a second pass delays retirement, not the other waves.
//             Results equal the sequential loop.
//   nearest   every vertex also lives as a 16-byte record {xy, index, vcost} in the array of its cell of a uniform cell grid
//             (the RRT* kernels' layout, rrt_block.h).  A wave streams the records of the cells its sample's radius ball touches
//             as one packed stream; the nearest record of the box is near()[0] of the whole tree whenever it is no farther
//             than the box radius (everything outside the box is), else the box is doubled.
//   chord bound  a Dubins word is never shorter than the chord between its end points, so vcost[v] + |v - x| bounds the cost
//             through v from below (oracle/dubins_oracle.c counts violations of the slackened bound in its study mode: none).
//             The walk of rrt.py:515-521 ends at the (cost, index)-smallest visible entry below the cost through the nearest
//             vertex; an entry whose bound is not below the best visible cost found so far can neither beat nor tie it (the
//             bounds used are strictly below the computed costs), so it is never priced.  On BASELINE configs[4] 2.6 % of the
//             near-set entries are priced (4 per accepted sample instead of 157).
//   pass 1    the first stream leaves in every lane the entry with the smallest bound among the records that lane saw; those
//             64 entries -- one of them replaced by the nearest vertex -- are priced in ONE word evaluation per lane, the
//             nearest vertex's word is swept (accept test), and the priced entries below the cost through it are swept in
//             (cost, index) order until one is visible.
//   pass 2    only if some lane saw a second entry whose bound is below the best cost so far: the stream runs again, entries
//             with a bound below the (tightening) best cost are collected in LDS and priced 64 at a time.
#pragma once

#include "rrt_kernels.h"

#ifndef RRT_DUB_STREAM_DEPTH
#define RRT_DUB_STREAM_DEPTH 2  // steps of a record stream in flight (measured 1..4: profiles/r03_experiments.md)
#endif

namespace rrtdev {

constexpr int DB_BUF = 128;  // collected entries per wave (pass 2): a step appends at most 64, a flush follows as soon as 64 are in
constexpr uint32_t DB_TINY = 64;  // a tree of up to this many vertices is looked at as a whole, one vertex per lane (no cell streams)
constexpr int DB_WIN = 64;   // samples in flight ahead of retirement
constexpr unsigned long long DB_STALL_TICKS = 200000000ull;  // 2 s of the 100 MHz wall clock
constexpr int DB_RING = 128; // ring of deposited / retired samples (>= 2 * DB_WIN: a retiring sample looks back at most DB_WIN - 1,
                             // the youngest sample in flight is at most DB_WIN - 1 ahead of the head)

// One sample as its wave resolved it against its snapshot of the tree (64 bytes); after retirement flags bit 2 says whether it
// was inserted and cb is its vertex cost (what younger samples in flight are checked against).
struct DbRec {
    uint32_t xq, hq;
    uint32_t nn_idx, nn_d2;  // nearest vertex of the snapshot
    uint32_t flags;          // bit 0: its word's sweep is free, bit 1: the sample's cell is already in `sampled`, bit 2: inserted (retired)
    uint32_t cells_nn;       // samples of that sweep read
    uint32_t hits;           // |within| over the snapshot (RRT*)
    uint32_t vb;             // parent
    double cb;               // cost through the parent
    uint32_t n_los, cells_cand, nwords;
    uint32_t snap_i;         // samples retired when it was resolved: it has seen exactly the samples before this one
    uint32_t ready;          // sample number + 1 once deposited (a slot is reused every DB_RING samples)
    uint32_t vidx;           // after retirement: its vertex
};
static_assert(sizeof(DbRec) == 64, "DbRec");

struct DbLds {
    alignas(16) uint32_t cellcnt[MAX_CELLS];  // live fill counts of the cells; go2goal's two 8 KiB tables afterwards
    alignas(16) u32x4 buf[NWAVE][DB_BUF];     // pass 2: collected entries {xy, index, vcost}
    alignas(16) DbRec ring[DB_RING];
    alignas(16) BSlot bslots[2 * NWAVE];
    alignas(8) unsigned long long state;  // samples retired << 32 | vertices: ONE word, so that a snapshot is consistent
    uint32_t next, lock;                  // ticket counter, retirement lock
    uint32_t fail;                        // a wave waited DB_STALL_TICKS without any sample retiring: everybody leaves (never seen; the exit every wave reaches)
    unsigned long long stat[6];
    alignas(16) double htab[3][256];  // angle, sine, cosine of the discrete headings (DubCfg::htab)
    uint32_t slots[NWAVE][64];        // the streams' cell starts of a step
    unsigned long long dbg[8];  // diagnostic build: [0] samples resolved again, [1] retirements that priced younger vertices, [2] those vertices
};

// conservative single-precision lower bound of vcost + chord (see the head comment): below the f64 value by more than every
// rounding on the way, for costs up to ~1e5 cells
__device__ __forceinline__ float db_lower_bound(double V, uint32_t d2) {
    const float s = ((float)V + __builtin_amdgcn_sqrtf((float)d2)) * (1.0f - 1.0e-6f) - 4.0e-3f;
    return s > 0.0f ? s : 0.0f;
}

__global__ __launch_bounds__(TPB) void rrt_dubins_block_kernel(BatchView bv) {
    __shared__ DbLds L;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int q = (int)blockIdx.x;
    QDesc *D = bv.desc + q;
    if (D->status != ST_RUNNING) return;

    // ---- per-query views ----
    const int n = D->n;
    const bool star = D->alg == 4;
    uint8_t *heading = bv.heading + (size_t)q * bv.node_stride;
    const uint8_t *shead = bv.sample_heading + (size_t)q * bv.n_cap;
    for (int h = t; h < D->nh && h < 256; h += TPB) {  // (visible after the barrier behind the fill counts' copy below)
        const double th = dub_heading(h, D->nh);
        L.htab[0][h] = th;
        dub_sincos(th, &L.htab[1][h], &L.htab[2][h]);
    }
    DubCfg dc{D->rho, D->nh, bv.W, bv.H};
    dc.htab = (const RRT_LDS double *)&L.htab[0][0];
    dc.inv_rho = 1.0 / D->rho;
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
    // radius of the first record stream: the rewire radius, but at least two cells (Dubins-RRT has no near set, and a tiny
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
    if (t < 6) L.stat[t] = 0ull;  // statistics: added to by whoever retires (under the lock)
    if (t < 8) L.dbg[t] = 0ull;
    if (t < DB_RING) L.ring[t].ready = 0u;
    if (t == 0) {
        L.state = ((unsigned long long)(uint32_t)i << 32) | (uint32_t)j;
        L.next = (uint32_t)i;
        L.lock = 0u;
        L.fail = 0u;
    }
#ifdef RRT_STAMPS
    unsigned long long cyc[6] = {D->cyc[0], D->cyc[1], D->cyc[2], D->cyc[3], D->cyc[4], D->cyc[5]};
    unsigned long long tstamp = __builtin_amdgcn_s_memtime();
#endif

#ifdef RRT_STAMPS
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
    __syncthreads();

    const float FINF = __uint_as_float(0x7f800000u);
    auto cell_of = [&](uint32_t X) -> int { return (ux(X) >> cshift) * ncy + (uy(X) >> cshift); };

    // The records of the cells that the box of half-width `rad` around X touches, as ONE packed stream: lane l of a step takes
    // record 64 * step + l of the concatenation of the cells' arrays (exclusive prefix sum of the fill counts over the lanes),
    // 64 cells at a time.  f(record, live) once per step.  The stream of rrt_pipe.h (round 4; until then this kernel found a
    // record's cell by a bisection of eight dependent ds_bpermute per step and dealt the box's corners too):
    // `keep_d2`: every vertex at a squared distance up to this must be dealt; cells farther away than that are left out.
    // Records of vertices at or above `jsnap` (inserted after the caller's snapshot) are dealt as dead lanes.
    volatile RRT_LDS uint32_t *slots = (volatile RRT_LDS uint32_t *)L.slots[wave];  // (lanes talk to each other through it: every access as written)
    auto stream_box = [&](uint32_t X, int rad, uint32_t keep_d2, uint32_t jsnap, auto &&f) {
        // a tree of up to 64 vertices: all of them in one step, from the vertex arrays instead of the cells' (the same answers; a
        // start pose that nothing can be connected to, and the first samples of every run, would otherwise walk ever larger boxes)
        const bool tiny = jsnap <= DB_TINY;
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
            // Which cell a record belongs to, without a search: every non-empty cell whose first record falls into this step writes
            // its number into that record's slot (64 words of LDS per wave), the lanes read their slots and a running maximum over
            // the lanes (DPP) carries the number to the records behind it; the lanes in front of the step's first cell start belong
            // to the cell the last step ended in.
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
            // RRT_DUB_STREAM_DEPTH steps in flight: the records of the next steps are requested before this step's are looked at
            constexpr int SD = RRT_DUB_STREAM_DEPTH;
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

    // a word (uniform after the call): the five values of lane `src`
    auto bcast_path = [&](const dub_path_t &p, int src) -> dub_path_t {
        dub_path_t o;
        o.t = __shfl(p.t, src);
        o.p = __shfl(p.p, src);
        o.q = __shfl(p.q, src);
        o.len = __shfl(p.len, src);
        o.word = __shfl(p.word, src);
        return o;
    };

    // Sweep the priced entries of this wave (lane: has, cost cn through vertex idx at a / ha by word pth) that can still become
    // the parent, cheapest first, until one is visible (rrt.py:515-521 ends at the (cost, index)-smallest visible entry below the
    // cost through the nearest vertex; `vb == NONE` while the nearest vertex, which wins every tie, is the parent).
    auto test_priced = [&](bool has, double cn, uint32_t idx, uint32_t a, int ha, const dub_path_t &pth, uint32_t xq, double &cb, uint32_t &vb,
                           uint32_t &nlos, uint32_t &ccells) {
        bool open = has;
        for (;;) {
            const bool better = open && (cn < cb || (cn == cb && vb != NONE && idx < vb));
            double c = better ? cn : f64_inf();
            uint32_t ix = better ? idx : NONE;
            wave_min_f64_idx(c, ix);
            if (ix == NONE) break;
            const unsigned long long m = __ballot(better && idx == ix);
            const int src = (int)__builtin_ctzll(m);
            const dub_path_t p = bcast_path(pth, src);
            const uint32_t pa = (uint32_t)__shfl((int)a, src);
            const int pha = __shfl(ha, src);
            int cc = 0;
            const bool ok = dub_sweep_wave(og, dc, pa, pha, xq, p, lane, cc);  // rrt.py:519
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

    // ---- retirement (under the lock): as far as the head of the ring is ready.  Returns the sample number of a head that has to
    //      be resolved again against the exact tree (its deposit is withdrawn; the caller does it next), else -1. ----
    auto try_retire = [&]() -> int {
        for (;;) {
            uint32_t got = 0;
            if (lane == 0) got = __hip_atomic_compare_exchange_strong(&L.lock, &got, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
            if (__builtin_amdgcn_readfirstlane((int)got) == 0) return -1;  // somebody else is retiring; it looks at the head again before it leaves
#ifndef RRT_NO_RETIRE_PRIO
            __builtin_amdgcn_s_setprio(3);  // the retirement is the one serial chain of the kernel: ahead of the three waves that share this SIMD
#endif
            int redo = -1;
#ifdef RRT_STAMPS
            const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
#endif
            const unsigned long long st0 = __hip_atomic_load(&L.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int h = (int)(st0 >> 32), jh = (int)(uint32_t)st0;  // local while the lock is held; published in batches
            const int h0 = h;
            // Insertions of this batch whose fill counts are not published yet: lane k holds the cell of the k-th.  The vertices'
            // stores are acknowledged ONCE per batch (s_waitcnt vmcnt(0)), then the counts and the state name them.
            uint32_t pend_cell = NONE;
            int npend = 0;
            auto publish = [&]() {
                if (h == h0 && npend == 0) return;
                if (npend > 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane < npend) __hip_atomic_fetch_add(&cellcnt[pend_cell], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (lane == 0)
                    __hip_atomic_store(&L.state, ((unsigned long long)(uint32_t)h << 32) | (uint32_t)jh, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                pend_cell = NONE;
                npend = 0;
            };
            for (;;) {
                if (h >= n) break;
                DbRec *slot = &L.ring[h & (DB_RING - 1)];
                if (__hip_atomic_load(&slot->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (uint32_t)h + 1u) break;
                DbRec r = *slot;
                // against the samples inserted since its snapshot: lane l <-> sample snap_i + l (fewer than DB_WIN of them)
                const int m = (int)r.snap_i + lane;
                bool ins = false;
                uint32_t xm = 0, hm = 0, vm = NONE;
                double cm = 0.0;
                if (m < h) {
                    const DbRec *e = &L.ring[m & (DB_RING - 1)];
                    ins = (e->flags & 4u) != 0u;
                    xm = e->xq;
                    hm = e->hq;
                    vm = e->vidx;
                    cm = e->cb;
                }
                const uint32_t d2 = dist2(xm, r.xq);
                const bool pre_ok = (r.flags & 3u) == 1u;  // visible from the nearest vertex, cell not sampled before its snapshot
                if (__ballot(ins && d2 < r.nn_d2) != 0ull) {  // a younger vertex is nearer (it loses ties: higher index): resolve again
                    if (lane == 0) __hip_atomic_store(&slot->ready, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    redo = h;
#ifdef RRT_STAMPS
                    if (lane == 0) L.dbg[0] += 1;
#endif
                    break;
                }
                const unsigned long long same = __ballot(ins && xm == r.xq);
                const unsigned long long inball = __ballot(ins && star && d2 < r2);
                uint32_t add_los = 0, add_cells = 0, add_words = 0;
                if (pre_ok && same == 0ull) {
                    // A younger vertex inside the ball is one more candidate parent (rrt.py:515-521 walks it too) unless its chord
                    // bound is not below the chosen cost: price those, one word per lane, and sweep the ones below it (a younger
                    // vertex loses every tie: higher index than the snapshot's choice)
                    const bool cnd = ins && star && d2 < r2 && cm + sqrt_u32(d2) * (1.0 - 1.0e-9) < r.cb;
                    if (__ballot(cnd) != 0ull) {
                        dub_path_t wp;
                        wp.t = wp.p = wp.q = 0.0;
                        wp.len = f64_inf();
                        wp.word = DUB_NONE;
                        double wcn = f64_inf();
                        if (cnd) {
                            wp = dub_between_dev(xm, (int)hm, r.xq, (int)r.hq, dc);
                            wcn = cm + wp.len;
                        }
                        add_words = (uint32_t)__builtin_popcountll(__ballot(cnd));
#ifdef RRT_STAMPS
                        if (lane == 0) {
                            L.dbg[1] += 1;
                            L.dbg[2] += add_words;
                        }
#endif
                        double cb = r.cb;
                        uint32_t vb = r.vb;
                        test_priced(cnd, wcn, vm, xm, (int)hm, wp, r.xq, cb, vb, add_los, add_cells);
                        r.cb = cb;
                        r.vb = vb;
                    }
                }
                const bool acc = pre_ok && same == 0ull && jh != n;  // rrt.py:425
                int c = 0;
                uint32_t pos = 0;
                if (acc) {
                    c = cell_of(r.xq);
                    pos = cellcnt[c] + (uint32_t)__builtin_popcountll(__ballot(pend_cell == (uint32_t)c));
                    if (lane == npend) pend_cell = (uint32_t)c;
                }
                if (lane == 0) {
                    L.stat[0] += (unsigned long long)jh;
                    L.stat[1] += (unsigned long long)r.cells_nn;
                    L.stat[5] += (unsigned long long)(r.nwords + add_words);
                    if (logs) {
                        bv.nearest_log[(size_t)q * bv.n_cap + h] = (int32_t)r.nn_idx;
                        bv.accept_log[(size_t)q * bv.n_cap + h] = (uint8_t)acc;
                        bv.cbest_log[(size_t)q * bv.n_cap + h] = __longlong_as_double(0x7ff8000000000000ll);
                        bv.j_log[(size_t)q * bv.n_cap + h] = jh;
                    }
                    if (acc) {  // rrt.py:524-529
                        if (star) {
                            L.stat[2] += (unsigned long long)r.hits + (unsigned long long)__builtin_popcountll(inball);
                            L.stat[4] += (unsigned long long)(r.n_los + add_los);
                            L.stat[3] += (unsigned long long)(r.cells_cand + add_cells);
                        }
                        nodes_g[jh] = r.xq;
                        heading[jh] = (uint8_t)r.hq;
                        vcost[jh] = r.cb;
                        parent[jh] = (int32_t)r.vb;
                        const uint32_t cellb = (uint32_t)ux(r.xq) * (uint32_t)H + (uint32_t)uy(r.xq);
                        atomicOr(&bitmap[cellb >> 5], 1u << (cellb & 31));  // rrt.py:426
                        const unsigned long long cbits = (unsigned long long)__double_as_longlong(r.cb);
                        cellrec[(size_t)c * (size_t)ccap + pos] = u32x4{r.xq, (uint32_t)jh, (uint32_t)cbits, (uint32_t)(cbits >> 32)};
                    }
                    // what younger samples in flight are checked against (read under the lock only)
                    slot->flags = (r.flags & 3u) | (acc ? 4u : 0u);
                    slot->cb = r.cb;
                    slot->vidx = (uint32_t)jh;
                }
                npend += acc ? 1 : 0;
                jh += acc ? 1 : 0;
                h += 1;
                if (npend == 16) publish();
            }
            publish();
#ifdef RRT_STAMPS
            if (lane == 0) {
                L.dbg[5] += __builtin_amdgcn_s_memtime() - tl0;
                L.dbg[6] += 1;
                L.dbg[7] += (unsigned long long)(h - h0);
            }
#endif
            if (lane == 0) __hip_atomic_store(&L.lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef RRT_NO_RETIRE_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            if (redo >= 0) return redo;
            // a deposit that arrived while the lock was held found it taken and left: look at the head once more
            if (h >= n || __hip_atomic_load(&L.ring[h & (DB_RING - 1)].ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (uint32_t)h + 1u) return -1;
        }
    };

    int mine = -1, redo = -1;  // a ticket taken but not resolved yet; a head to resolve again (first)
    int wait_done = -1;        // bounded waiting: the retired count when this wave began to wait, and when
    unsigned long long wait_t0 = 0;
    for (;;) {
        int s;
        if (redo >= 0) {
            s = redo;
            redo = -1;
        } else {
            if (mine < 0) {
                uint32_t tk = 0;
                if (lane == 0) tk = __hip_atomic_fetch_add(&L.next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                mine = __builtin_amdgcn_readfirstlane((int)tk);
                if (mine > n) mine = n;  // (the counter runs on while the waves drain)
            }
            const int done = (int)(__hip_atomic_load(&L.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32);
            if (mine >= n && done >= n) break;
            if (mine >= n || mine - done >= DB_WIN) {  // nothing left to take, or too far ahead of retirement: help retiring, wait
                if (__hip_atomic_load(&L.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;
                if (done != wait_done) {
                    wait_done = done;
                    wait_t0 = wall_clock64();
                } else if (wall_clock64() - wait_t0 > DB_STALL_TICKS) {
                    if (lane == 0) __hip_atomic_store(&L.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                redo = try_retire();
                if (redo < 0) __builtin_amdgcn_s_sleep(16);
                DSTAMP(4);  // (diagnostic build, wave 0) waiting
                continue;
            }
            s = mine;
            mine = -1;
        }
        // =============================== resolve sample s against a snapshot ===============================
        {
            const unsigned long long snap = __hip_atomic_load(&L.state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t snap_i = (uint32_t)(snap >> 32), jsnap = (uint32_t)snap;
            const uint32_t xq = samples[s];
            const int hq = (int)shead[s];
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
                const float lb = hit ? db_lower_bound(V, d2) : FINF;
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
            int radn = rad0;
            // nothing in the box, or something that a vertex outside the box could beat: double the box (nearest only)
            while (jsnap > DB_TINY && (nn_d2 == NONE || nn_d2 > (uint32_t)radn * (uint32_t)radn) && radn < (W > H ? W : H)) {
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
            // ---- one word per lane ----
            int e_h = 0;
            dub_path_t e_p;
            e_p.t = e_p.p = e_p.q = 0.0;
            e_p.len = f64_inf();
            e_p.word = DUB_NONE;
            double e_cn = f64_inf();
            if (e_idx != NONE) {
                e_h = (int)heading[e_idx];
                e_p = dub_between_dev(e_xy, e_h, xq, hq, dc);
                e_cn = e_V + e_p.len;
            }
            uint32_t nwords = (uint32_t)__builtin_popcountll(__ballot(e_idx != NONE));
            DSTAMP(1);  // one word per lane
            // ---- nearest vertex: cost through it, its sweep (rrt.py:422-425) ----
            const dub_path_t p_nn = bcast_path(e_p, slot);
            const int h_nn = __shfl(e_h, slot);
            const double c_nn = __shfl(e_cn, slot);
            int cells = 0;
            const bool nocoll = dub_sweep_wave(og, dc, nn_xy, h_nn, xq, p_nn, lane, cells);
            const bool dup = ((bm_word >> (cell & 31)) & 1u) != 0u;
            double cb = c_nn;
            uint32_t vb = NONE, nlos = 0, ccells = 0;
            if (star && nocoll && !dup) {
                // ---- choose parent: the priced entries, then whatever pass 1 left unpriced below the best cost so far ----
                test_priced(e_idx != NONE && lane != slot, e_cn, e_idx, e_xy, e_h, e_p, xq, cb, vb, nlos, ccells);
                DSTAMP(2);  // the nearest vertex's sweep, the sweeps of the priced entries
                if (__ballot((double)left < cb) != 0ull) {
                    RRT_LDS u32x4 *buf = (RRT_LDS u32x4 *)L.buf[wave];
                    uint32_t nbuf = 0;
                    const uint32_t skip = (lane == slot) ? NONE : m1idx;  // this lane's entry of pass 1 (the stream deals the same records to the same lanes)
                    auto flush = [&]() {
                        u32x4 e = {0u, NONE, 0u, 0u};
                        if ((uint32_t)lane < nbuf) e = buf[lane];
                        const double V = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                        bool has = e.y != NONE && (double)db_lower_bound(V, dist2(e.x, xq)) < cb;  // (the best cost may have fallen since the entry was collected)
                        int fh = 0;
                        dub_path_t fp;
                        fp.t = fp.p = fp.q = 0.0;
                        fp.len = f64_inf();
                        fp.word = DUB_NONE;
                        double fcn = f64_inf();
                        if (has) {
                            fh = (int)heading[e.y];
                            fp = dub_between_dev(e.x, fh, xq, hq, dc);
                            fcn = V + fp.len;
                        }
                        nwords += (uint32_t)__builtin_popcountll(__ballot(has));
                        test_priced(has, fcn, e.y, e.x, fh, fp, xq, cb, vb, nlos, ccells);
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
                        const bool take = d2 < r2 && rc.y != nn_idx && rc.y != skip && (double)db_lower_bound(V, d2) < cb;
                        const unsigned long long tm = __ballot(take);
                        if (tm == 0ull) return;
                        if (take) buf[nbuf + (uint32_t)__builtin_popcountll(tm & ((1ull << lane) - 1ull))] = rc;
                        nbuf += (uint32_t)__builtin_popcountll(tm);
                        if (nbuf >= 64u) flush();
                    });
                    while (nbuf > 0u) flush();
                    DSTAMP(3);  // pass 2: second stream, its word evaluations and sweeps
                }
            }
            DSTAMP(2);
            if (lane == 0) {
                DbRec *slot = &L.ring[s & (DB_RING - 1)];
                DbRec r;
                r.xq = xq;
                r.hq = (uint32_t)hq;
                r.nn_idx = nn_idx;
                r.nn_d2 = nn_d2;
                r.flags = (nocoll ? 1u : 0u) | (dup ? 2u : 0u);
                r.cells_nn = (uint32_t)cells;
                r.hits = nhits;
                r.vb = vb == NONE ? nn_idx : vb;
                r.cb = cb;
                r.n_los = nlos;
                r.cells_cand = ccells;
                r.nwords = nwords;
                r.snap_i = snap_i;
                r.ready = 0u;
                r.vidx = NONE;
                *slot = r;
                __hip_atomic_store(&slot->ready, (uint32_t)s + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        redo = try_retire();
        DSTAMP(5);  // deposit + retirement
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
                       n_los_cand = D->n_los_cand, n_words = D->n_words;
    sum_j += L.stat[0];
    sum_cells_nn += L.stat[1];
    sum_near += L.stat[2];
    sum_cells_cand += L.stat[3];
    n_los_cand += L.stat[4];
    n_words += L.stat[5];
    __syncthreads();  // (go2goal reuses the LDS)

    // ---------------- go2goal (rrt.py:311-332) ----------------
    int status = ST_DONE, vgoal = 0, found = 0;
    if (L.fail != 0u || i < n) {
        status = ST_TEAM_FAIL;  // (reported as RRT_E_HIP; the tree is consistent up to sample i)
    } else {
        double pc;
        uint32_t pi;
        go2goal_phase<true>(og, H, nodes_g, vcost, 0, 1, j, xg, reinterpret_cast<uint32_t *>(spill), (RRT_LDS uint32_t *)L.cellcnt, L.bslots, t, lane, wave, pc, pi,
                            heading, D->hg, dc);
        if (pi != NONE) {
            found = 1;
            vgoal = j;
            if (t == 0) {
                nodes_g[j] = xg;
                vcost[j] = pc;
                parent[j] = (int32_t)pi;
                heading[j] = (uint8_t)D->hg;
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
        D->n_words = n_words;
#ifdef RRT_STAMPS
        for (int k = 0; k < 6; ++k) D->cyc[k] = cyc[k];
        for (int k = 0; k < 8; ++k) D->wcyc[k] = L.dbg[k];
#endif
    }
}

#undef DSTAMP

}  // namespace rrtdev
