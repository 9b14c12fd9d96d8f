// rrt_block.h -- block-parallel tree expansion: up to 64 samples per pass of the node array, on 1 to 64 CUs per query.
//
// The sample stream of RRTStandard / RRTStar does not depend on the tree (rrt.py:240), so a (super-)block of consecutive
// samples is evaluated against the tree as it stood at the start of the block (the snapshot, nodes [0, j0)), and the few
// dependencies between samples of one block are resolved afterwards in sample order.  The result is bit-identical to the
// sequential loop.  A team of G workgroups (one per CU) shares the block: member g takes BSM samples (see "teams" below).
//
//   A  scan (all 16 waves of every member; RRT* with r_rewire < 16, and RRTStandard on teams of 8 and more): every lane loads 4 nodes (16 bytes) once
//      and evaluates them against the member's samples held in scalar registers: coordinates pre-scaled by 16 make
//      v_dot2_i32_i16(d, d, tag) = 256*d2 + tag the packed nearest-neighbour key in ONE instruction (brute force over the whole
//      live tree, near :150-155).
//      -- barrier --
//   B  owner phase: one wave (or a group of 2..16 waves) per sample.
//      RRT*: the radius-ball near set (within :176-181) comes from a uniform cell grid over the map: every tree node also
//      lives as a 16-byte record {xy, index, vcost} in the array of its cell, so the owners stream the records of the cells
//      the ball touches as one packed stream (64 live records per step, coalesced 16-byte loads), price them
//      (vcost + sqrt(d2)) and find the first entry in (cost, index) order with cost < cost-via-nearest and a free line of sight
//      (choose parent :511-521).  The nearest node comes out of the same stream -- a hit lies inside the ball, everything outside
//      the streamed cells is farther -- so phase A does not run for these queries; a sample whose ball is empty gets its nearest
//      from one wave's own pass over the node cache.  vcost never changes after the insert (the rewire predicate :536 is never
//      true), so the copy in the record stays valid.
//      Then: the line of sight nearest -> sample and the `sampled` bit (rrt.py:424-425).
//      -- barrier; members g > 0 hand their records to member 0 --
//   C  commit (wave 0 of member 0, one lane per sample): samples whose result cannot be changed by an earlier sample of the
//      same block (none inserted that is nearer than the snapshot nearest, on the same cell, or a candidate parent that
//      would be tried before the snapshot's choice) commit together, lane-parallel; a sample that can is re-resolved against
//      snapshot + inserted block nodes on its own, in order.  An Informed block is cut where the ellipse changes
//      (rrt.py:698-700, :744-745).
//      -- member 0 publishes the new state; barrier --
#pragma once

#include "rrt_kernels.h"

namespace rrtdev {

constexpr int BS = 16;  // most samples one workgroup resolves per pass (see BSM below)

// ---- teams: G workgroups (CUs) on one query ---------------------------------------------------------------------
// A single query is a chain of inserts, but the expensive parts of a block -- scan and owner phase -- only read the
// snapshot.  A team of G workgroups therefore takes a super-block of up to 64 samples: member g scans and resolves BSM of
// them on its own CU (own LDS copy of the node cache and of the cell fill counts; BSM = 16 with one wave per sample for
// G <= 4, else 64 / G with a group of 16 / BSM waves per sample), hands its records to member 0, whose wave 0 commits all
// samples in order (one lane per sample) and publishes the new state; every member then appends the new nodes to its LDS
// copies.  Two hand-offs per super-block:
//   records   member g>0 -> member 0: owners leave their record in LDS; after the workgroup barrier ONE wave writes the
//             member's records write-through (agent-scope stores), drains, and stores the member's arrival flag; wave 0 of
//             member 0 polls the flags and reads the records with agent-scope loads (they bypass its L1)
//   commit    member 0 -> members g>0: plain stores (nodes, costs, parents, cell records, bitmap), agent release,
//             s_waitcnt, flag; a member polls the flag with one wave, runs ONE agent acquire (drops its L1), waits for it,
//             joins the workgroup barrier, and only then the workgroup loads.
// Teams can be pipelined (template parameter PIPE below; instantiated for 2, 3, 4, 8, 16, 32 and 64 workers): one more workgroup
// that only commits, the workers two blocks ahead of it.  All members must be resident together (the launch keeps teams x members <= CUs); every spin is bounded
// by a wall-clock limit that fails the query (status ST_TEAM_FAIL, the host then continues it with one CU) instead of
// hanging the device.
constexpr int TEAM_MAX = 64;
#ifndef RRT_PIPE_LAG
#define RRT_PIPE_LAG 2  // blocks a pipelined team's workers run ahead of the commit (a batch without Informed queries)
#endif
#ifndef RRT_PIPE_LAG_SMALL
#define RRT_PIPE_LAG_SMALL 1  // ... where one wave resolves a sample (teams of up to 4 workers).  Such a team's period is its workers'
                              // throughput, not the ring's latency: a second block in flight buys nothing and costs the commit its masks
                              // and re-resolutions (measured, profiles/r04_experiments.md §9: config 4's share 6.45 -> 6.16 ms, teams of 8 and more
                              // workers 4 - 10 % slower with one block)
#endif
#ifndef RRT_PIPE_LAG_INF
#define RRT_PIPE_LAG_INF 2  // ... when the batch may hold Informed queries (a commit that moves the ellipse voids the blocks in flight)
#endif
#ifndef RRT_NPMAX
#define RRT_NPMAX 2  // most blocks in flight a record can carry interaction masks for (>= both RRT_PIPE_LAG values)
#endif
constexpr int NPMAX = RRT_NPMAX;
// Owners on big teams (a group of waves per sample) also test the lines of sight from every sample IN FLIGHT within r_rewire to
// their sample -- positions come from the sample stream, not from the tree -- and hand the answers over as a list of up to PL_MAX
// 16-bit entries, so that the committer settles "an inserted sample in flight is a cheaper parent" (nine in ten of the samples a
// commit has to look at again) lane-parallel, without a line-of-sight test of its own.  Entry: bits 0-5 sample, 6-7 set (0 = oldest
// previous block ... NP = this block), 8-14 cells the test read, 15 free; order: oldest block first, sample order = node order.
#ifndef RRT_PL_MAX
#define RRT_PL_MAX 12
#endif
constexpr int PL_MAX = RRT_PL_MAX, PL_WORDS = (RRT_PL_MAX + 3) / 4;
constexpr int BREC_WORDS = 10 + 3 * NPMAX + PL_WORDS;  // 8-byte words of an owner's record (BRec below): 152 bytes for two blocks in flight
// per query: [go | fail | state (NPMAX + 1 slots of 64 bytes) | records (NPMAX + 1 slots of 64) | arrival flags (65 x 128) | go2goal answers (65 x 16)]
constexpr int TEAM_OFF_GO = 128, TEAM_OFF_FAIL = 256, TEAM_OFF_STATE = 384, TEAM_OFF_REC = 1024;
constexpr int TEAM_OFF_ARRIVE = (TEAM_OFF_REC + (NPMAX + 1) * 64 * BREC_WORDS * 8 + 127) / 128 * 128;
constexpr int TEAM_OFF_RES = TEAM_OFF_ARRIVE + 65 * 128;
constexpr int TEAM_BYTES = (TEAM_OFF_RES + 65 * 16 + 1023) / 1024 * 1024;
static_assert(TEAM_OFF_STATE + (NPMAX + 1) * 64 <= TEAM_OFF_REC, "state slots");
constexpr unsigned long long TEAM_TIMEOUT_TICKS = 50000000ull;  // 0.5 s of the 100 MHz wall clock

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) u64 gu64;
#define RRT_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ u64 lowmask64(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }

// Element idx of a per-query array (all of them are far below 4 GiB): base + zero-extended 32-bit byte offset, which the
// compiler turns into a scalar base with a 32-bit vector offset instead of keeping a 64-bit copy of the base in vector registers.
template <class T>
__device__ __forceinline__ T &at32(T *base, uint32_t idx) {
    return *reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(const_cast<typename std::remove_const<T>::type *>(base)) +
                                  (size_t)(uint32_t)(idx * (uint32_t)sizeof(T)));
}

// Loads of tree data that another workgroup of the team has written in this launch (nodes, costs, cell records, the bitmap):
// agent-scope loads (sc1: past this CU's L1, served by L2 / memory) where COH, so that taking a commit needs no L1 invalidate;
// plain loads for a query that runs on one CU.
template <bool COH>
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) {
    if (COH) return __hip_atomic_load((gu32 *)const_cast<uint32_t *>(p), RRT_RLX_AGENT);
    return *p;
}
template <bool COH>
__device__ __forceinline__ double ld_f64(const double *p) {
    if (COH) return __longlong_as_double((long long)__hip_atomic_load((gu64 *)const_cast<double *>(p), RRT_RLX_AGENT));
    return *p;
}
template <bool COH>
__device__ __forceinline__ u32x4 ld_rec(const u32x4 *p) {  // (two 8-byte halves: a record is complete before its go flag is raised)
    if (COH) {
        const gu64 *q = (const gu64 *)p;
        const u64 a = __hip_atomic_load(q, RRT_RLX_AGENT), b = __hip_atomic_load(q + 1, RRT_RLX_AGENT);
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    }
    return *p;
}

// 16 bytes at agent scope (what the compiler emits for an agent-scope atomic load, at four times the width the atomic builtins
// reach).  The caller issues all its loads, then fence_b128s(), then uses the values.
__device__ __forceinline__ u32x4 ld_b128_agent(const void *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void fence_b128s(u32x4 (&v)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));  // (every use of v[k] stays behind the wait)
}

// A value every lane of the wave holds alike (read from LDS or memory): moved to scalar registers, so that it does not
// take a vector register per lane for as long as it lives.
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int unis32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); }
__device__ __forceinline__ double unif64(double v) { return __longlong_as_double((long long)uni64((u64)__double_as_longlong(v))); }

// One wave polls the arrival flags of `count` members starting with member `first` (lane l: member first + l, flags 128 bytes
// apart) until all have reached `target`.
__device__ __forceinline__ bool team_wait_all(gu32 *flags, int first, int count, uint32_t target, gu32 *fail, int lane) {
    const u64 t0 = wall_clock64();
    for (;;) {
        const bool mine = lane < count ? __hip_atomic_load(flags + 32 * (first + lane), RRT_RLX_AGENT) >= target : true;
        if (__all(mine)) return true;
        if (__hip_atomic_load(fail, RRT_RLX_AGENT) != 0u) return false;
        if (wall_clock64() - t0 > TEAM_TIMEOUT_TICKS) {
            __hip_atomic_store(fail, 1u, RRT_RLX_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// One wave polls one word until it reaches `target`; false on timeout or when another member has failed.
__device__ __forceinline__ bool team_wait(gu32 *word, uint32_t target, gu32 *fail) {
    const u64 t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load(word, RRT_RLX_AGENT) >= target) return true;
        if (__hip_atomic_load(fail, RRT_RLX_AGENT) != 0u) return false;
        if (wall_clock64() - t0 > TEAM_TIMEOUT_TICKS) {
            __hip_atomic_store(fail, 1u, RRT_RLX_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

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

// key = 256*d2 + tag of one node against one sample (both pre-scaled by 16): v_pk_sub_i16 + v_dot2_i32_i16.
// Inline asm: hipcc pads no hazards around it; both instructions only read SALU-written scalars and plain VGPRs.
__device__ __forceinline__ uint32_t key16(uint32_t node_s, uint32_t q_s, uint32_t tag) {
    uint32_t d, r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(node_s), "s"(q_s));
    asm("v_dot2_i32_i16 %0, %1, %1, %2" : "=v"(r) : "v"(d), "s"(tag));
    return r;
}

// Owner's publication for one sample (128 bytes = 16 words of 8 bytes).
struct BRec {
    uint32_t d2s, vs;   // snapshot nearest
    uint32_t los_s;     // line of sight vs -> sample: bit 31 free, low bits cells read
    uint32_t flags;     // bit 0: cell already in `sampled` at the snapshot; bit 1 (committer's copy): re-resolved, nnear counts the block's nodes;
                        // bits 8-15: samples in flight within r_rewire (0 where no list is kept; more than PL_MAX: the list is void)
    double Vs;          // vcost[vs]
    double cbest;       // cost of the sample through its snapshot-resolved parent
    uint32_t vbest;     // snapshot-resolved parent (vs, or the best passing near-set entry)
    uint32_t pstat;     // owner's candidate line-of-sight tests: count << 20 | cells
    uint32_t nnear;     // |within| over the snapshot
    float amin;         // no parent below the bound: a lower bound of the cheapest near-set entry at or above it (0: unknown)
    double pc;          // cost of the best passing near-set entry (inf: none, parent is vs)
    u64 nnmask;         // earlier samples of the (super-)block strictly nearer than the snapshot nearest
    u64 rmask;          // earlier samples within r_rewire
    u64 dupmask;        // earlier samples on the same cell
    u64 pnn[NPMAX], pr[NPMAX], pdup[NPMAX];  // pipelined teams: the same three masks against the samples of the previous super-block [0]
                                 // and of the one before it [1] (workers two blocks ahead of the commit)
    u64 plist[PL_WORDS];         // big pipelined teams: lines of sight from the samples in flight within r_rewire (see PL_MAX)
};
static_assert(sizeof(BRec) == 8 * BREC_WORDS, "BRec size");
union BRecWords {
    BRec r;
    u64 w[BREC_WORDS];
};

// Block state that wave 0 hands to the other waves after the commit.
struct BlkState {
    int32_t i, j, nsoln, vbest_soln, pad0, pad1;
    double cmin_soln, c_ell;
};

template <int BSM>
__device__ __forceinline__ void block_scan_step(u32x4 quad, const uint32_t (&xs16)[BSM], uint32_t (&best)[BSM], uint32_t tag0) {
    const uint32_t n0 = quad.x << 4, n1 = quad.y << 4, n2 = quad.z << 4, n3 = quad.w << 4;
#pragma unroll
    for (int k = 0; k < BSM; ++k) {
        const uint32_t k0 = key16(n0, xs16[k], tag0), k1 = key16(n1, xs16[k], tag0 + 1), k2 = key16(n2, xs16[k], tag0 + 2),
                       k3 = key16(n3, xs16[k], tag0 + 3);
        best[k] = min(min(best[k], k0), min(min(k1, k2), k3));
    }
}

// Owner phase of a sample handled by a group of waves (team members with fewer samples than waves): what each wave
// found in its share of the cells, and what the group's leader decided.
struct GSlot {
    double c1, c2;        // after stream_cells: the share's two cheapest; after consume_list: c1 = its cheapest passing entry
    uint32_t i1, i2;
    uint32_t hits, nlist; // after stream_cells: |within| of the share, parked entries; after consume_list: nlist = open entries
    uint32_t nn_d2, nn_idx;  // after stream_cells: the nearest of the share's hits (NONE: no hit)
    uint32_t pad[2];      // (consume: pad[0] = the share's amin)
    uint32_t x1, x2;      // coordinates of the two cheapest (the records hold them: no look-up in front of their lines of sight)
    uint32_t nn_xy, nn_vlo, nn_vhi;  // coordinates and vcost of that nearest hit (its record's)
};
struct GCtl {
    double lbc;
    uint32_t lbi, consume;  // consume: the two cheapest are blocked, every wave tests its own parked entries above (lbc, lbi)
    double bound;           // the cost through the nearest node (the group's waves learn it from the leader)
};

union BlkWords {
    BlkState b;
    u64 w[5];
};

struct ParRound {  // pipelined committer: the samples re-resolved side by side, one wave each
    u64 acc_opt, aprev[NPMAX];
    uint32_t list[NWAVE];
    uint32_t accs[64];  // per sample: inserted by its re-resolution?
    uint32_t count;
    int32_t jp0[NPMAX];
};

// How far the workers run ahead of the commit: two blocks (every record carries masks against the samples of BOTH blocks in
// flight; a commit of an Informed query that ends early or moves the ellipse voids both).
template <bool PIPE, bool INF, int BSM>
struct PipeShape {
    static constexpr int LAG = PIPE ? (INF ? RRT_PIPE_LAG_INF : (BSM >= 16 ? RRT_PIPE_LAG_SMALL : RRT_PIPE_LAG)) : 0;
    static_assert(LAG <= NPMAX, "a record carries masks for NPMAX blocks in flight");
    static constexpr int NP = LAG > 0 ? LAG : 1;      // previous blocks a record / the commit looks at (array extents)
    static constexpr int NSLOT = PIPE ? LAG + 1 : 1;  // record and state buffers in the hand-off area: by block number modulo NSLOT
};

// The static LDS of the block kernel (one object per workgroup, whatever its role).
template <int G, int BSM, bool PIPE, bool INF>
struct BlockLds {
    static constexpr int SB = BSM * G, NP = PipeShape<PIPE, INF, BSM>::NP;
    alignas(16) u32x2 nnx[(BSM <= BS ? BSM : 1) * NWAVE];     // per own sample, per wave: {d2, idx} (phase A; not with more than 16 samples per member)
    alignas(16) BRec brec[PIPE ? 2 : 1][SB];  // (a pipelined committer: this block's records and the next one's)
    uint32_t xq_next[PIPE ? 2 : 1][PIPE ? 64 : 1];  // pipelined committer: the next block's samples and whether its records are in,
    uint32_t pre_state[2];                          // by the half of brec they belong to (fetched while the block before them commits)
    alignas(16) BSlot bslots[2 * NWAVE];
    alignas(16) BlkState blk;
    alignas(16) unsigned long long statred[SB * 5];
    uint32_t xq_lds[SB];
    double newcost[SB];
    alignas(16) ParRound par;
    uint32_t xqp_lds[PIPE ? NP : 1][PIPE ? 64 : 1];  // pipelined teams: the samples of the previous super-block(s) ...
    double prevcost[PIPE ? NP : 1][PIPE ? 64 : 1];   // ... and (committer) the exact costs of the nodes they inserted
    uint32_t help_n[NWAVE], help_x[NWAVE];  // single-wave owners: open candidates of a blocked sample that all waves test together, its coordinates
    uint32_t help_any;
    uint32_t tick;  // BSM > 16: the next sample of this member's share that a wave takes when it is through with its own
    uint32_t slots[BSM >= 16 ? NWAVE : 1][64];  // single-wave owners: the cell starts of a step of the near-set stream
    alignas(16) GSlot gslot[NWAVE];
    alignas(16) GCtl gctl[BSM];
    alignas(8) uint16_t plist[(PIPE && BSM < 16) ? BSM : 1][PL_MAX];  // a group's list of in-flight lines of sight, filled by its waves
    uint32_t oq_cnt[(PIPE && BSM < 16) ? BSM : 1], oq_ovf;
    uint32_t qhist[16], qstage2;  // a big queue (BSM == 1): open candidates per sixteenth of the cost range; "nothing passed among the cheap ones"  // worker groups: fill of a blocked sample's queue of open candidates, "a queue overflowed"
#ifdef RRT_STAMPS
    unsigned long long dbg[16];  // pipelined teams: phase cycles of wave 0 of the committer and of worker 1
#endif
};

// What a workgroup does: everything (teams without a pipeline), or one of the two halves of a pipelined team.  The body is
// instantiated per role so that neither half carries the other's state through its loop.
constexpr int ROLE_ALL = 0, ROLE_COMMIT = 1, ROLE_WORK = 2;

// PIPE (teams of 8 and more): G workers plus one workgroup that only commits, and a pipeline of super-blocks -- while block s
// is committed the workers already resolve block s + 1 (and s + 2: PipeShape::LAG) against the tree as it stood BEFORE block s;
// every record then also carries the three interaction masks against the samples of each block in flight, and the commit of a
// block treats the nodes those blocks inserted like inserted samples of its own block (their acceptance and costs are exact
// by then).
// INF: the batch may hold Informed queries (alg 2); without it everything the ellipse needs is compiled out.

// CW: waves of THIS workgroup (16; round 3 measured a committer of 8 waves as a kernel of its own, compiled for 256 vector registers:
// no gain, profiles/r03_experiments.md -- the parameter stays, the kernels went).
template <int G, int BSM, bool PIPE, bool INF, int ROLE, int CW = NWAVE>
__device__ __forceinline__ void rrt_block_body(BatchView bv, BlockLds<G, BSM, PIPE, INF> &L) {
    static_assert(CW == NWAVE || ROLE == ROLE_COMMIT, "only a committer runs with fewer waves");
    constexpr int NTG = CW * 64;  // threads of this workgroup
    static_assert(G >= 1 && G <= TEAM_MAX && BSM >= 1 && BSM * G <= 64, "team size");
    // BSM > 16 (a pipelined team of 2 workers: 32 samples per member, 64 per super-block): one wave per sample as with 16, and a
    // wave that is through takes the next of the member's samples off a counter; the hand-overs of a block are then shared by
    // twice the samples.  (The owner phase is bound by the CU's vector issue, not by its slowest sample: 21 samples per member of
    // a team of three took as long per sample as 16.)  No phase A: the launch only takes the variant when every query's
    // near-set radius spans a cell (grid_nn).
    constexpr bool WIDE = BSM > BS;
    static_assert(!PIPE || G > 1, "a pipeline needs a team");
    static_assert(PIPE == (ROLE != ROLE_ALL), "roles are the halves of a pipelined team");
    constexpr int LAG = PipeShape<PIPE, INF, BSM>::LAG, NP = PipeShape<PIPE, INF, BSM>::NP, NSLOT = PipeShape<PIPE, INF, BSM>::NSLOT;
    constexpr int SB = BSM * G;  // samples per (super-)block: one lane of the committing wave each; BSM per member
    // steps (of 64 records) of the near-set stream a wave has in flight: two where one wave streams a sample's whole ball, one
    // where a group of waves shares it (a wave of a group rarely has a second step, and the team kernels sit at the register cap:
    // measured, profiles/r02_experiments.md)
#ifndef RRT_CG16
#define RRT_CG16 2
#endif
    constexpr int CG = (BSM >= 16) ? (ROLE == ROLE_COMMIT ? 2 : RRT_CG16) : 1;
#ifndef RRT_BLOCK_SCATTER
#define RRT_BLOCK_SCATTER 1
#endif
    constexpr bool SCATTER = RRT_BLOCK_SCATTER && BSM >= 16;  // (one wave streams a sample's whole ball: every call runs part 0 of 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [node cache | cell fill counts]
    auto &nnx = L.nnx;
    auto &brec = L.brec;
    auto &xq_next = L.xq_next;
    auto &pre_state = L.pre_state;
    auto &bslots = L.bslots;
    auto &blk = L.blk;
    auto &statred = L.statred;
    auto &xq_lds = L.xq_lds;
    auto &newcost = L.newcost;
    auto &par = L.par;
    auto &xqp_lds = L.xqp_lds;
    auto &prevcost = L.prevcost;
    auto &gslot = L.gslot;
    auto &gctl = L.gctl;
    constexpr int WPS = BSM >= 16 ? 1 : NWAVE / BSM;  // waves per sample in the owner phase
    constexpr bool FASTL = PIPE && WPS > 1;  // records carry the lines of sight from the samples in flight (PL_MAX): big pipelined teams
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    int q = (int)blockIdx.x, g_ = 0;  // query, team member
    if (G > 1) {
        q = (int)blockIdx.x % bv.team_qpad;  // members of one team are 8k blocks apart: dealt to the same XCD (speed only)
        g_ = (int)blockIdx.x / bv.team_qpad + bv.member0;  // (member0 = 1: the workers' own launch, the committer runs elsewhere)
        if (q >= bv.Q) return;
        if (bv.team_fault && g_ == 1) return;
    }
    const int g = ROLE == ROLE_COMMIT ? 0 : g_;
    if (ROLE == ROLE_WORK) __builtin_assume(g > 0);
    const bool worker = !PIPE || ROLE == ROLE_WORK;  // scans and resolves samples (a pipelined team's member 0 only commits)
    const int wg = PIPE ? (ROLE == ROLE_COMMIT ? 0 : g - 1) : g;  // which BSM samples of a super-block this workgroup owns (a committer: none)
    QDesc *D = bv.desc + q;
    if (D->status != ST_RUNNING) return;
    unsigned char *tb = (G > 1) ? bv.team + (size_t)q * TEAM_BYTES : nullptr;
    gu32 *const t_arrive = (gu32 *)(tb + TEAM_OFF_ARRIVE), *const t_go = (gu32 *)(tb + TEAM_OFF_GO), *const t_fail = (gu32 *)(tb + TEAM_OFF_FAIL);
    gu64 *const t_state = (gu64 *)(tb + TEAM_OFF_STATE), *const t_rec = (gu64 *)(tb + TEAM_OFF_REC);  // PIPE: NSLOT of each, by block number
    uint32_t epoch = 0;  // super-blocks of this launch so far
    // pre-scan (team members, RRTStandard / RRTStar): while member 0 commits block s, a member already scans the snapshot of
    // block s for the samples of block s + 1; after the commit only the steps that hold the new nodes are scanned again
    int pre_i = -1, pre_j = 0;  // the iteration the pre-scan is for, the node count it covered
    constexpr int BSA = WIDE ? 1 : BSM;  // samples of phase A (none where a member takes more than 16)
    uint32_t pre_xv = 0, pre_best[BSA];
#pragma unroll
    for (int k = 0; k < BSA; ++k) pre_best[k] = NONE;
    bool team_failed = false;
    u64 A_prev[NP];  // pipelined committer: the samples of the previous block(s) that were inserted, and the node count before each
    int jp0[NP];
#pragma unroll
    for (int p2 = 0; p2 < NP; ++p2) {
        A_prev[p2] = 0;
        jp0[p2] = 0;
    }

    // ---- per-query views ----
    const int n = D->n, alg = D->alg;
    const bool star = alg >= 1, informed = INF && alg == 2;
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
    // cell grid of the near-set records
    const int cshift = D->cell_shift, ncy = D->ncy, ccap = D->cell_cap, ncells = D->ncx * D->ncy;
    u32x4 *cellrec = reinterpret_cast<u32x4 *>(bv.cellrec) + (size_t)q * (size_t)bv.rec_stride;
    uint32_t *cellcnt_g = bv.cellcnt + (size_t)q * (size_t)MAX_CELLS;
    // The radius of the record stream: r_rewire for RRT* (the near set, within :176-181); for RRTStandard, which has no near set,
    // two cells -- there the stream only serves the nearest-neighbour search (every tree keeps its nodes in the cell records).
    // (Only where one wave resolves a sample, i.e. teams of up to 4 workers and single CUs: the many-query shapes.  A single
    // RRTStandard query on a big team is bound by its committer, which would only pay for the records: measured 5.08 -> 5.27 ms.)
    const bool cells_on = star || BSM >= 16;
    const uint32_t r2h = star ? r2 : (cells_on ? (uint32_t)((2 << cshift) * (2 << cshift)) : 0u);
    int rad = 0;  // largest |dx| with dx*dx < r2h
    if (r2h > 0) {
        rad = (r2h >= (1u << 23)) ? 4096 : (int)sqrtf((float)(r2h - 1));
        while (rad > 0 && (uint32_t)(rad * rad) > r2h - 1) --rad;
        while ((uint32_t)((rad + 1) * (rad + 1)) <= r2h - 1) ++rad;
    }

    // ---- LDS carve ----
    RRT_LDS uint32_t *nodes_lds = (RRT_LDS uint32_t *)smem;
    const RRT_LDS u32x4 *nodes_lds4 = (const RRT_LDS u32x4 *)smem;
    RRT_LDS uint32_t *cellcnt = (RRT_LDS uint32_t *)(smem + (size_t)lds_chunks * CHUNK * sizeof(uint32_t));

    // ---- state ----
    int i = D->i, j = D->j;
    int nsoln = D->nsoln, vbest_soln = D->vbest_soln;
    double cmin_soln = D->cmin_soln;
    int i_switch = D->i_switch;
    int status = ST_RUNNING;
    // statistics: per-lane accumulators of wave 0 in LDS (statred), folded once at the end
#ifdef RRT_STAMPS
    unsigned long long cyc[6] = {D->cyc[0], D->cyc[1], D->cyc[2], D->cyc[3], D->cyc[4], D->cyc[5]};
    unsigned long long tstamp = __builtin_amdgcn_s_memtime();
    unsigned long long wcyc_acc = 0, wcyc_los = 0;
    auto &dbg = L.dbg;
    if (t < 16) dbg[t] = 0;
    unsigned long long dbgt = __builtin_amdgcn_s_memtime();
#ifdef RRT_STAMPS_LIGHT  // only the stamps named by this bit mask (a stamp costs ~200 cycles: sixteen per block shift the balance of the ring)
#define DBGT(k)                                                      \
    do {                                                             \
        if (((RRT_STAMPS_LIGHT) >> (k)) & 1) {                       \
            unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
            if (t == 0) dbg[k] += now_ - dbgt;                       \
            dbgt = now_;                                             \
        }                                                            \
    } while (0)
#else
#define DBGT(k)                                                  \
    do {                                                         \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        if (t == 0) dbg[k] += now_ - dbgt;                       \
        dbgt = now_;                                             \
    } while (0)
#endif
#else
#define DBGT(k) \
    do {        \
    } while (0)
#endif

#if defined(RRT_STAMPS)
    // wall-clock stamp of event ev of block ep (32 blocks of the run from RRT_TS_BASE on), by one lane
#ifndef RRT_TS_BASE
#define RRT_TS_BASE 300
#endif
#define TSMARK(ep, ev)                                                                                          \
    do {                                                                                                        \
        if ((int)(ep) >= RRT_TS_BASE && (int)(ep) < RRT_TS_BASE + 32 && lane == 0) D->ts[((int)(ep) - RRT_TS_BASE) * 16 + (ev)] = wall_clock64(); \
    } while (0)
#else
#define TSMARK(ep, ev) \
    do {               \
    } while (0)
#endif
#if defined(RRT_STAMPS) && defined(RRT_STAMPS_OWNER)
    unsigned long long wst_ = 0;
#define WST0() (wst_ = __builtin_amdgcn_s_memtime())
#define WST(k)                                                     \
    do {                                                           \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        if (PIPE && g == 1 && t == 0) dbg[k] += now_ - wst_;       \
        wst_ = now_;                                               \
    } while (0)
#else
#define WST0() \
    do {       \
    } while (0)
#define WST(k) \
    do {       \
    } while (0)
#endif
    const double xc0 = ((double)(D->xs[0] + D->xg[0])) / 2.0, xc1 = ((double)(D->xs[1] + D->xg[1])) / 2.0;
    const double d2sg = (double)dist2(xs, xg);
    const double C00 = D->C[0], C01 = D->C[1], C10 = D->C[2], C11 = D->C[3];
    double c_ell = 0.0;
    if (informed && nsoln > 0) c_ell = cmin_soln + sqrt_u24(dist2(xg, nodes_g[vbest_soln]));

    // ---- prologue: node cache = live nodes, unfilled slots = copy of node 0 (never nearest: equal distance,
    //      higher index); cell fill counts from HBM ----
    {
        const uint32_t n0 = nodes_g[0];
        for (int k = t; k < lds_nodes; k += NTG) nodes_lds[k] = (k < j) ? nodes_g[k] : n0;
        for (int k = t; k < ncells; k += NTG) cellcnt[k] = cellcnt_g[k];
        if (t < SB * 5) statred[t] = 0;
        if (t < NWAVE) L.help_n[t] = 0;
        if (t == 0) L.help_any = 0;
        if (t < (int)(sizeof(L.oq_cnt) / sizeof(uint32_t))) L.oq_cnt[t] = 0;
        if (t == 0) L.oq_ovf = 0;
        if (t < 16) L.qhist[t] = 0;
        if (t == 0) L.qstage2 = 0;
    }
    __syncthreads();

    // A team reads whatever its committer stores (nodes, costs, cell records, bitmap) with agent-scope loads: where the nearest node
    // comes from the record stream (grid_nn, below) no plain load of such bytes is left, and taking a commit costs no L1 invalidate.
    constexpr bool COH = G > 1;
    auto node_xy = [&](uint32_t v) -> uint32_t { return ((int)v < lds_nodes) ? nodes_lds[v] : ld_u32<COH>(nodes_g + v); };
    auto cell_of = [&](uint32_t X) -> int { return (ux(X) >> cshift) * ncy + (uy(X) >> cshift); };

#include "rrt_block_nearset.inc"
    // ---- nearest node from the record stream (owners of RRT* samples) ---------------------------------------------------
    // The owner streams the cells of the ball ONCE, without a bound (the bound is the cost through the nearest node, which the
    // same stream finds): top two of all hits, all hits parked, and the nearest hit = near()[0] of the whole snapshot whenever
    // the ball holds a node.  The brute-force scan of the node array (phase A) then never runs; a sample whose ball is empty --
    // the first samples of a run, pockets the tree has not reached -- gets its nearest from one wave's own pass over the nodes.
    // Used when the radius spans at least a cell; smaller radii keep phase A (the ball is empty too often).
    const bool grid_nn = WIDE || rad >= 16;  // (RRTStandard: always, its stream radius is two cells of at least 16 pixels)
    if (WIDE && rad < 16) {  // (the launch does not take these variants for such a query; should it: leave at once, the one-CU kernel continues)
        if (t == 0 && g_ == 0) D->status = ST_TEAM_FAIL;
        return;
    }
    // what snapshot_parent does behind its stream, for a stream that ran without the bound
    // amin (only meaningful when no parent is found): a lower bound of the cheapest entry at or above the bound -- the stream ran
    // without the bound, so its two cheapest entries are the two cheapest of the whole ball.  The committer needs it when a node
    // of a block in flight turns out to be the sample's nearest: the bound moves up, and only an entry between the old bound and
    // the new one makes it search the ball again.
    auto lower_f32 = [](double c) -> float { return __builtin_fmaxf((float)c * (1.0f - 1.0e-6f) - 4.0e-3f, 0.0f); };
    // defer: the blocked-candidate list is only priced here (open = its open entries, left in the list); the caller has them tested
    auto finish_parent = [&](uint32_t X, int j0, double bound, Top2 tt, uint32_t nlist, double &pc, uint32_t &pi, uint32_t &ntests, uint32_t &tcells,
                             float &amin, bool defer, uint32_t &open) {
        pc = f64_inf();
        pi = NONE;
        open = 0;
        amin = tt.i1 == NONE ? FINF : lower_f32(tt.c1);
        if (tt.i1 == NONE || !(tt.c1 < bound)) return;  // rrt.py:518, strict
        amin = tt.i2 == NONE ? FINF : lower_f32(tt.c2);  // (the cheapest entry is below the bound)
        if (tt.i2 != NONE && !(tt.c2 < bound)) tt.i2 = NONE;
        bool ok1, ok2;
        int cc1, cc2;
        los_wave2(og, H, node_xy(tt.i1), tt.i2 != NONE ? node_xy(tt.i2) : X, tt.i2 != NONE, X, lane, ok1, cc1, ok2, cc2);
        ntests += 1;
        tcells += (uint32_t)cc1;
        if (ok1) {
            pc = tt.c1;
            pi = tt.i1;
            return;
        }
        if (tt.i2 == NONE) return;
        ntests += 1;
        tcells += (uint32_t)cc2;
        if (ok2) {
            pc = tt.c2;
            pi = tt.i2;
            return;
        }
        amin = 0.0f;  // both are below the bound: unknown unless the parked list tells
        if (nlist > clist_cap) {  // the list overflowed: stream again, bounded, above the two that are blocked
            uint32_t nn2 = 0;
            snapshot_parent(X, j0, false, bound, pc, pi, nn2, ntests, tcells, tt.c2, tt.i2 + 1);
            return;
        }
        uint32_t nval = 0;
        if (defer) {
            consume_price(X, bound, tt.c2, tt.i2 + 1, nlist, nval, amin);
            open = nval;
            return;
        }
        consume_list(X, bound, tt.c2, tt.i2 + 1, nlist, pc, pi, nval, amin);
        count_tests(nval, pc, pi, ntests, tcells);
    };
    // near()[0] (rrt.py:150-155) of one sample over the snapshot [0, j0) by ONE wave: 256 nodes per step, 4 per lane
    auto wave_scan_nearest = [&](uint32_t X, int j0, uint32_t &d2s, uint32_t &vs) {
        uint32_t bd = NONE, bi = NONE;
        for (int base = 0; base < j0; base += 256) {
            const int idx0 = base + 4 * lane;
            if (idx0 < j0) {
                u32x4 v;
                if (idx0 < lds_nodes) v = nodes_lds4[idx0 >> 2];
                else v = ld_rec<COH>(nodes_g4 + (idx0 >> 2));
                const uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t d = dist2(pv[e], X);
                    if (idx0 + e < j0 && d < bd) {  // (a lane meets its nodes in index order: strict < keeps the lowest index)
                        bd = d;
                        bi = (uint32_t)(idx0 + e);
                    }
                }
            }
        }
        wave_min_key_idx(bd, bi);
        d2s = bd;
        vs = bi;
    };

    // steps [c0, c1) of the scan: 4096 nodes per step, from the LDS cache or (beyond it) from HBM, next step prefetched
    const uint32_t node0 = xs;  // node 0 = the start
    auto scan_steps = [&](int c0, int c1, int jlim, const uint32_t (&xs16)[BSA], uint32_t (&best)[BSA]) {
        const int nl = c1 < lds_chunks ? c1 : lds_chunks;
        if (c0 < nl) {
            u32x4 cur = nodes_lds4[c0 * TPB + t];
            for (int c = c0; c < nl; ++c) {
                u32x4 nxt = cur;
                if (c + 1 < nl) nxt = nodes_lds4[(c + 1) * TPB + t];
                block_scan_step<BSA>(cur, xs16, best, (uint32_t)c << 2);
                cur = nxt;
            }
        }
        const int g0 = c0 > nl ? c0 : nl;
        if (c1 > g0) {
            u32x4 cur = nodes_g4[g0 * TPB + t];
            for (int c = g0; c < c1; ++c) {
                u32x4 nxt = cur;
                if (c + 1 < c1) nxt = nodes_g4[(c + 1) * TPB + t];
                if (PIPE) {  // the committer may be appending nodes beyond this worker's snapshot right now: read them as node 0
                    const int base = c * CHUNK + 4 * t;
                    if (base + 0 >= jlim) cur.x = node0;
                    if (base + 1 >= jlim) cur.y = node0;
                    if (base + 2 >= jlim) cur.z = node0;
                    if (base + 3 >= jlim) cur.w = node0;
                }
                block_scan_step<BSA>(cur, xs16, best, (uint32_t)c << 2);
                cur = nxt;
            }
        }
    };

    // An Informed query on a pipelined team: the workers resolve block s + 1 as if the commit of block s neither ends early nor
    // changes the ellipse.  When it does (a better solution, 13 times in BASELINE config 3), its state says RESTART: the block
    // in flight is void -- the committer takes its turn without committing anything -- and the workers start over from the
    // true state.  A worker therefore never ends the loop on its own count: it leaves when a state says the run is over.
    constexpr int32_t ST_FLAG_RESTART = 1, ST_FLAG_STOP = 2;
    const bool pipe_inf = PIPE && informed;
    int nprev = 0;            // worker: how many previous blocks exist (their samples are in xqp_lds[0 .. nprev))
    int void_turns = 0;       // committer: the last commit ended early or changed the ellipse: the LAG blocks in flight are void
    bool prefetched = false;  // committer: wave 1 fetched this block's records during the last commit ...
    bool prefetched_smp = false;  // ... and wave 2 its samples (not those of an Informed query: the ellipse may move)
    int bsel = 0;             // committer: which half of brec holds this block's records
    auto publish_state = [&](uint32_t ep, int32_t flags) {  // wave 0 of member 0: state of block `ep`, then its go flag
        if (lane == 0) {
            BlkWords u;
            u.b.i = i;
            u.b.j = j;
            u.b.nsoln = nsoln;
            u.b.vbest_soln = vbest_soln;
            u.b.pad0 = 0;
            u.b.pad1 = flags;
            u.b.cmin_soln = cmin_soln;
            u.b.c_ell = c_ell;
#pragma unroll
            for (int w = 0; w < 5; ++w) __hip_atomic_store(t_state + (size_t)(ep % NSLOT) * 8 + w, u.w[w], RRT_RLX_AGENT);
        }
        // wave 0 made every store of the commit, all of them write-through (nodes, costs, parents, cell records, the bitmap's
        // atomic, the logs' plain stores are host-read only): the flag only must not overtake them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(t_go, ep, RRT_RLX_AGENT);
        // this CU read lines of the arrays just extended through its L1 (the neighbours of the new entries: the scan's plain
        // loads); the stores above bypass it, so drop it like every other member does when it takes the commit
        if (!grid_nn) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    while ((pipe_inf && g > 0) || i < n) {
        const int i0 = i, j0 = j;
        int nb = (n - i0) < SB ? (n - i0) : SB;  // samples of this (super-)block; lane s of every wave: sample s
        const bool ell = informed && nsoln > 0;
        bool void_blk = false;  // worker of a pipelined Informed query: nothing it could resolve this turn
        if (nb <= 0) {
            nb = 0;
            void_blk = true;
        }
        // ---------------- sample coordinates of the block (rrt.py:421 / :502 / :695-701) ----------------
        if (ell) {
            if (i_switch == n) i_switch = i0;
            const int u0i = i0 - ub_offset;
            if (ub == nullptr || u0i < 0 || u0i + nb > ub_count) {
                if (pipe_inf && g > 0) {  // the committer will say so: take the turn, learn the state
                    nb = 0;
                    void_blk = true;
                } else {
                    status = ST_NEED_UB;
                    if (pipe_inf && wave == 0) publish_state(epoch + 1, ST_FLAG_STOP);
                    break;
                }
            }
        }
        ++epoch;
        if (INF && PIPE && g == 0 && void_turns > 0) {  // a block in flight was resolved for a state that no longer holds: a turn without a commit
            if (wave == 0) {
                const bool ok = team_wait_all(t_arrive, 1, G, epoch, t_fail, lane);
                if (ok) publish_state(epoch, 0);
                if (lane == 0) blk.pad0 = ok ? 0 : 1;
            }
            --void_turns;
            prefetched = prefetched_smp = false;  // (whatever was fetched belonged to the void block)
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) {
                A_prev[p2] = 0;
                jp0[p2] = j;
            }
            __syncthreads();
            if (blk.pad0 != 0) {
                team_failed = true;
                break;
            }
            continue;
        }
        if (ROLE == ROLE_COMMIT) DBGT(0);
        if (ROLE == ROLE_COMMIT && wave == 0) TSMARK(epoch, 0);
        uint32_t xv = 0;  // lane s < nb: sample s
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
                if (PIPE && prefetched_smp) xv = xq_next[bsel][lane];  // (a branch of its own: merged with the global load it becomes a flat load)
                else xv = (G > 1 && pre_i == i0) ? pre_xv : at32(samples, (uint32_t)(i0 + lane));
            }
        }
        if (!worker && t < SB) xq_lds[t] = xv;  // a pipelined team's committer: nothing to resolve
#ifdef RRT_STAMPS
        const unsigned long long wres0 = __builtin_amdgcn_s_memtime();
        unsigned long long gs0 = wres0, gs1 = wres0, gs2 = wres0, gs3 = wres0;
        (void)gs0; (void)gs1; (void)gs2; (void)gs3;
#endif
#include "rrt_block_worker.inc"

#include "rrt_block_handover.inc"

#include "rrt_block_commit.inc"
        STAMP(4);
        __syncthreads();
        DBGT(15);
        {
            const BlkState b = blk;
            if (G > 1 && unis32(b.pad0) != 0) {
                team_failed = true;
                break;
            }
            if (INF && PIPE && (unis32(b.pad1) & ST_FLAG_RESTART) != 0) void_turns = LAG;
            prefetched = pre_next;
            prefetched_smp = pre_smp;
            if (PIPE && pre_next) bsel ^= 1;
            i = unis32(b.i);
            j = unis32(b.j);
            nsoln = unis32(b.nsoln);
            vbest_soln = unis32(b.vbest_soln);
            cmin_soln = unif64(b.cmin_soln);
            c_ell = unif64(b.c_ell);
        }
    }

    // ---------------- the end of the run: member 0 finishes the query; a team shares go2goal ----------------
    constexpr uint32_t FINAL = 0x40000000u;   // "epoch" of the final hand-offs, above every block's
    constexpr int NWG = G + (PIPE ? 1 : 0);   // workgroups of the team
    if (team_failed) {
        if (G > 1 && g > 0) return;
        status = ST_TEAM_FAIL;
    }
    if (G > 1 && g > 0) {
        // the members wait for the final tree (a pipelined worker left the loop ahead of the last commits)
        if (wave == 0) {
            const bool ok = team_wait(t_go, FINAL, t_fail);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) blk.pad0 = ok ? 0 : 1;
        }
        __syncthreads();
        if (blk.pad0 != 0) return;
        BlkWords u;
#pragma unroll
        for (int w = 0; w < 5; ++w) u.w[w] = __hip_atomic_load(t_state + (size_t)(FINAL % NSLOT) * 8 + w, RRT_RLX_AGENT);
        if ((u.b.pad1 & ST_FLAG_STOP) != 0) return;  // no goal connection this launch (the host has to supply data, or a failure)
        j = u.b.j;
    } else if (G > 1 && status != ST_TEAM_FAIL) {
        if (wave == 0) publish_state(FINAL, status == ST_RUNNING ? 0 : ST_FLAG_STOP);
    }

    // cell fill counts back to HBM (a resumed launch reloads them); fold wave 0's statistics
    if (g == 0) {
        for (int k = t; k < ncells; k += NTG) cellcnt_g[k] = cellcnt[k];
    }
    __syncthreads();

    // ---------------- go2goal (rrt.py:311-332): the first node in (cost-to-goal, index) order with a free line of sight.  Every
    //                  workgroup of the team answers for the nodes g, g + NWG, g + 2 NWG, ...; member 0 takes the minimum ----------------
    int vgoal = 0, found = 0;
    if (status == ST_RUNNING) {
        status = ST_DONE;
        double pc;
        uint32_t pi;
        const int cnt = j > g ? (j - g + NWG - 1) / NWG : 0;
        go2goal_phase<false, NTG>(og, H, nodes_g, vcost, g, NWG, cnt, xg, reinterpret_cast<uint32_t *>(clist_base), (RRT_LDS uint32_t *)smem, bslots, t, lane,
                                  wave, pc, pi);
        if (G > 1) {
            gu64 *const t_res = (gu64 *)(tb + TEAM_OFF_RES);
            if (g > 0) {
                if (wave == 0) {
                    if (lane == 0) {
                        __hip_atomic_store(t_res + 2 * g, (u64)__double_as_longlong(pc), RRT_RLX_AGENT);
                        __hip_atomic_store(t_res + 2 * g + 1, (u64)pi, RRT_RLX_AGENT);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(t_arrive + 32 * g, FINAL, RRT_RLX_AGENT);
                }
#ifdef RRT_STAMPS
                if (PIPE && t == 0 && g == 1)
#pragma unroll
                    for (int k = 0; k < 16; ++k) D->wcyc[16 + k] = dbg[k];
                if (PIPE && t == 0 && g >= 1 && g <= 64) D->dbg2[128 + g - 1] = dbg[0];
#endif
                return;
            }
            if (wave == 0) {
                const bool ok = team_wait_all(t_arrive, 1, NWG - 1, FINAL, t_fail, lane);
                double c = f64_inf();
                uint32_t ci = NONE;
                if (ok && lane < NWG - 1) {
                    c = __longlong_as_double((long long)__hip_atomic_load(t_res + 2 * (lane + 1), RRT_RLX_AGENT));
                    ci = (uint32_t)__hip_atomic_load(t_res + 2 * (lane + 1) + 1, RRT_RLX_AGENT);
                }
                if (lane == 63 && key_lt(pc, pi, c, ci)) {  // (NWG - 1 <= 64 members in lanes 0..NWG-2; this workgroup's own answer: lane 63 ...
                    c = pc;
                    ci = pi;
                }
                if (NWG - 1 == 64) {  // ... unless all 64 lanes are taken: fold it in after the reduction)
                    wave_min_f64_idx(c, ci);
                    if (key_lt(pc, pi, c, ci)) {
                        c = pc;
                        ci = pi;
                    }
                } else {
                    wave_min_f64_idx(c, ci);
                }
                if (lane == 0) {
                    bslots[0].pc = c;
                    bslots[0].pi = ci;
                    bslots[0].tested = ok ? 0u : 1u;
                }
            }
            __syncthreads();
            pc = bslots[0].pc;
            pi = bslots[0].pi;
            if (bslots[0].tested != 0u) status = ST_TEAM_FAIL;
        }
        if (status == ST_DONE) {
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
        }
        STAMP(5);
    }

    if (t == 0) {
        unsigned long long s[5] = {0, 0, 0, 0, 0};
        for (int l = 0; l < SB; ++l)
            for (int c = 0; c < 5; ++c) s[c] += statred[l * 5 + c];
        D->status = status;
        D->i = i;
        D->j = j;
        D->nsoln = nsoln;
        D->vbest_soln = vbest_soln;
        D->cmin_soln = cmin_soln;
        D->vgoal = vgoal;
        D->found = found;
        D->i_switch = i_switch;
        D->sum_j += s[0];
        D->sum_cells_nn += s[1];
        D->sum_near += s[2];
        D->sum_cells_cand += s[3];
        D->n_los_cand += s[4];
#ifdef RRT_STAMPS
        for (int k = 0; k < 6; ++k) D->cyc[k] = cyc[k];
#endif
    }
#ifdef RRT_STAMPS
    if (lane == 0 && !PIPE) {
        D->wcyc[wave] = wcyc_acc;
        D->wcyc[16 + wave] = wcyc_los;
    }
    if (PIPE && t == 0 && g <= 1)
#pragma unroll
        for (int k = 0; k < 16; ++k) D->wcyc[16 * g + k] = dbg[k];
#endif
}

// (RRT_BLOCK_DECL_ONLY: a translation unit that only launches the kernel; csrc/kernels_tu.hip holds the instantiations, dealt to
//  several translation units so that they compile side by side)
template <int G, int BSM, bool PIPE, bool INF>
__global__ __launch_bounds__(TPB) void rrt_expand_block_kernel(BatchView bv)
#ifdef RRT_BLOCK_DECL_ONLY
    ;
#else
{
    __shared__ BlockLds<G, BSM, PIPE, INF> L;
    if constexpr (PIPE) {
#if defined(RRT_ONLY_ROLE) && RRT_ONLY_ROLE == 1  // (resource analysis of one role; never run)
        rrt_block_body<G, BSM, PIPE, INF, ROLE_COMMIT>(bv, L);
#elif defined(RRT_ONLY_ROLE) && RRT_ONLY_ROLE == 2
        rrt_block_body<G, BSM, PIPE, INF, ROLE_WORK>(bv, L);
#else
        if ((int)blockIdx.x < bv.team_qpad) rrt_block_body<G, BSM, PIPE, INF, ROLE_COMMIT>(bv, L);
        else rrt_block_body<G, BSM, PIPE, INF, ROLE_WORK>(bv, L);
#endif
    } else {
        rrt_block_body<G, BSM, PIPE, INF, ROLE_ALL>(bv, L);
    }
}
#endif

}  // namespace rrtdev
