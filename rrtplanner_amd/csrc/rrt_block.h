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
constexpr int PL_MAX = 12, PL_WORDS = 3;
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
template <bool PIPE, bool INF>
struct PipeShape {
    static constexpr int LAG = PIPE ? (INF ? RRT_PIPE_LAG_INF : RRT_PIPE_LAG) : 0;
    static_assert(LAG <= NPMAX, "a record carries masks for NPMAX blocks in flight");
    static constexpr int NP = LAG > 0 ? LAG : 1;      // previous blocks a record / the commit looks at (array extents)
    static constexpr int NSLOT = PIPE ? LAG + 1 : 1;  // record and state buffers in the hand-off area: by block number modulo NSLOT
};

// The static LDS of the block kernel (one object per workgroup, whatever its role).
template <int G, int BSM, bool PIPE, bool INF>
struct BlockLds {
    static constexpr int SB = BSM * G, NP = PipeShape<PIPE, INF>::NP;
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
    constexpr int LAG = PipeShape<PIPE, INF>::LAG, NP = PipeShape<PIPE, INF>::NP, NSLOT = PipeShape<PIPE, INF>::NSLOT;
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

    // ---- the near set of one sample (within :176-181, choose parent :511-521) ----------------------------------------
    // First entry in (cost, index) order with cost < bound and a free line of sight, over the snapshot nodes [0, j0).
    // Three steps, used by one wave on its own (snapshot_parent) or by the waves of a sample's group (owner phase of
    // teams whose members hold fewer samples than waves):
    //   stream_cells   the 16-byte records of the cells the radius ball touches are streamed once (this wave's share of the
    //                  cells).  Every entry whose vcost alone is below the bound is parked in this wave's HBM list
    //                  {index, d2, vcost}; the two cheapest entries are found on the fly under a screen that tightens to
    //                  the second cheapest so far.
    //   consume_list   only when both are blocked: the parked list is priced and the entries still open are tested.
    //   count_tests    the statistics as the sequential loop counts them.
    // Where one wave resolves a sample (BSM == 16) the first 256 entries of its list live in LDS, the rest in HBM: a sample whose
    // two cheapest candidates are blocked reads its list back several times, and that sample is the one its whole block waits
    // for.  Groups of waves (each with a share of the ball) keep their lists in HBM.
    // GQ: the workers of a big pipelined team (a group of waves per sample).  A sample whose two cheapest candidates are blocked puts its
    // open candidates into ONE queue per group, in the half of brec only a committer uses, and all the group's waves test them eight
    // per wave and memory round trip (see the owner phase).  (Round 4 also kept the waves' parked lists there: no gain, dropped.)
    constexpr bool GQ = PIPE && ROLE == ROLE_WORK && BSM < 16;
    constexpr uint32_t OQCAP = (uint32_t)(SB * BREC_WORDS * 8 / 16 / (BSM < 16 ? BSM : 1));
    constexpr bool GLIST = false;
    constexpr bool LDSLIST = BSM >= 16 || GLIST;
    constexpr uint32_t LCAP = BSM >= 16 ? (uint32_t)BLOCK_LIST_CAP : (uint32_t)(SB * BREC_WORDS * 8 / 16 / NWAVE);
    const uint32_t clist_cap = (uint32_t)(bv.spill_stride / (2 * NWAVE * (G + (PIPE ? 1 : 0))));  // the engine sizes the spill area per team member
    u32x4 *const clist_base = reinterpret_cast<u32x4 *>(spill) + (size_t)(g * NWAVE) * (size_t)clist_cap;  // this member's 16 lists (and go2goal's scratch)
    u32x4 *const clist = clist_base + (size_t)wave * (size_t)clist_cap;
    RRT_LDS u32x4 *const clist_l =
        GLIST ? (RRT_LDS u32x4 *)&brec[PIPE ? 1 : 0][0] + (size_t)wave * LCAP
              : (RRT_LDS u32x4 *)(smem + (size_t)lds_chunks * CHUNK * sizeof(uint32_t) + (size_t)MAX_CELLS * sizeof(uint32_t)) + (size_t)wave * LCAP;
    auto lget = [&](uint32_t p) -> u32x4 {
        if (LDSLIST && p < LCAP) return clist_l[p];
        return clist[p];
    };
    auto lput = [&](uint32_t p, u32x4 e) {
        if (LDSLIST && p < LCAP) clist_l[p] = e;
        else clist[p] = e;
    };
    auto lput_y = [&](uint32_t p, uint32_t y) {
        if (LDSLIST && p < LCAP) clist_l[p].y = y;
        else clist[p].y = y;
    };
    // the list of another wave of this workgroup (the cooperative test of a blocked sample's candidates)
    RRT_LDS u32x4 *const clist_l0 = clist_l - (size_t)wave * LCAP;  // wave 0's
    auto lget_w = [&](int w, uint32_t p) -> u32x4 {
        if (LDSLIST && p < LCAP) return clist_l0[(size_t)w * LCAP + p];
        return (clist_base + (size_t)w * (size_t)clist_cap)[p];
    };
    auto lput_y_w = [&](int w, uint32_t p, uint32_t y) {
        if (LDSLIST && p < LCAP) clist_l0[(size_t)w * LCAP + p].y = y;
        else (clist_base + (size_t)w * (size_t)clist_cap)[p].y = y;
    };
    // single-precision screen: vcost + sqrt(d2) evaluated in f32 is within a few f32 ulps (< 4e-7 relative) of the f64 value
    auto screen_of = [](double c) -> float { return (float)c * (1.0f + 1.0e-6f) + 4.0e-3f; };
    auto hi_of = [](double c) -> uint32_t { return (uint32_t)((unsigned long long)__double_as_longlong(c) >> 32); };
    const float FINF = __uint_as_float(0x7f800000u);

    // nn_d2 / nn_idx: the nearest of the hits (smallest d2, lowest index among equals; NONE / NONE without a hit).  A hit lies
    // within r_rewire, and every node outside the streamed cells is farther than that: if there is a hit, this IS the nearest
    // node of the whole snapshot (near()[0], rrt.py:150-155) -- the brute-force scan is only needed when the ball is empty.
    // Top2::wave_reduce with the entries' coordinates riding along: lane values (tt, fx1, fx2) -> the wave's two cheapest in tt and
    // their coordinates in (ox1, ox2).  (An index names one record, so one lane holds each winner.)
    auto top2_reduce_xy = [&](Top2 &tt, uint32_t fx1, uint32_t fx2, uint32_t &ox1, uint32_t &ox2) {
        double bc = tt.c1;
        uint32_t bi = tt.i1;
        wave_min_f64_idx(bc, bi);
        const bool own = tt.i1 != NONE && tt.c1 == bc && tt.i1 == bi;
        double sc = own ? tt.c2 : tt.c1;
        uint32_t si = own ? tt.i2 : tt.i1;
        const uint32_t sxy = own ? fx2 : fx1;
        const double sc_l = sc;
        const uint32_t si_l = si;
        wave_min_f64_idx(sc, si);
        const u64 m1 = __ballot(own), m2 = __ballot(si_l != NONE && sc_l == sc && si_l == si);
        ox1 = m1 ? (uint32_t)__builtin_amdgcn_readlane((int)fx1, (int)__builtin_ctzll(m1)) : 0u;
        ox2 = m2 ? (uint32_t)__builtin_amdgcn_readlane((int)sxy, (int)__builtin_ctzll(m2)) : 0u;
        tt.c1 = bc;
        tt.i1 = bi;
        tt.c2 = sc;
        tt.i2 = si;
    };
    // sx: what the stream also knows about its answers from the records themselves (wave-uniform after the call): the coordinates of
    // the two cheapest entries, the coordinates and vcost of the nearest hit
    struct StreamExtra {
        uint32_t x1, x2, nn_xy, nn_vlo, nn_vhi;
    } sx = {0u, 0u, 0u, 0u, 0u};
    auto stream_cells = [&](uint32_t X, int j0, bool check_j0, double bound, double lbc, uint32_t lbi, int part, int nparts, Top2 &tt,
                            uint32_t &nnear_part, uint32_t &nlist, uint32_t &nn_d2, uint32_t &nn_idx) {
        uint32_t fx1 = 0, fx2 = 0, lxy = 0, lvlo = 0, lvhi = 0;  // this lane's: coordinates of its two cheapest, its nearest hit's coordinates and vcost
        const int x = ux(X), y = uy(X);
        const int cx0 = (x - rad < 0 ? 0 : x - rad) >> cshift, cx1 = (x + rad > W - 1 ? W - 1 : x + rad) >> cshift;
        const int cy0 = (y - rad < 0 ? 0 : y - rad) >> cshift, cy1 = (y + rad > H - 1 ? H - 1 : y + rad) >> cshift;
        const int ny = cy1 - cy0 + 1, ncr = (cx1 - cx0 + 1) * ny;
        const float boundf = screen_of(bound);
        const uint32_t boundhi = hi_of(bound);  // vcost >= bound  <=>  its high word > boundhi or (== and ...): `<=` keeps a superset
        tt.init();
        float m1 = FINF, m2 = FINF;  // this lane's two cheapest verified entries, rounded up to f32
        uint32_t hits = 0;
        uint32_t ld2 = NONE, lidx = NONE;  // this lane's nearest hit
        nlist = 0;
        float T = boundf;            // wave-uniform screen, tightens to the second cheapest so far
        uint32_t Thi = hi_of((double)boundf);
        // one 16-byte record per lane: hit test, parking, the screens and the exact price (rrt.py:176-181, :515-518)
        auto eval_record = [&](const u32x4 rc, bool &dirty) {
            const uint32_t d2 = dist2(rc.x, X);
            const bool hit = d2 < r2h && (!check_j0 || rc.y < (uint32_t)j0);
            hits += hit ? 1u : 0u;
            if (hit && (d2 < ld2 || (d2 == ld2 && rc.y < lidx))) {
                ld2 = d2;
                lidx = rc.y;
                lxy = rc.x;
                lvlo = rc.z;
                lvhi = rc.w;
            }
            if (!star) return;  // RRTStandard: the stream only names the nearest node
            // park everything whose vcost alone can be below the bound (the high word of a non-negative f64 is monotone)
            const bool park = hit && rc.w <= boundhi;
            const unsigned long long pm = __ballot(park);
            if (pm == 0) return;
            if (park) {
                const uint32_t pos = nlist + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull));
                if (pos < clist_cap) lput(pos, u32x4{rc.y, rc.x, rc.z, rc.w});  // {index, xy, vcost}: whoever tests the entry has its coordinates at hand
            }
            nlist += (uint32_t)__builtin_popcountll(pm);
            // screens, cheapest first; none rejects an entry that belongs to the two cheapest
            const bool pre = park && rc.w <= Thi;
            if (__ballot(pre) == 0) return;
            const double V = __longlong_as_double((long long)(((unsigned long long)rc.w << 32) | rc.z));
            const bool maybe = pre && ((float)V + __builtin_amdgcn_sqrtf((float)d2) < T);
            if (__ballot(maybe) == 0) return;
            if (maybe) {
                const double cn = V + sqrt_u24(d2);
                const bool in = cn < bound && !key_lt(cn, rc.y, lbc, lbi);  // rrt.py:518, strict
                if (in) {
                    if (key_lt(cn, rc.y, tt.c1, tt.i1)) {
                        fx2 = fx1;
                        fx1 = rc.x;
                    } else if (key_lt(cn, rc.y, tt.c2, tt.i2)) {
                        fx2 = rc.x;
                    }
                    tt.fold(cn, rc.y);
                }
                // (m1, m2) <- the two smallest of {m1, m2, cu}, without branches: written as conditional assignments the pair
                // ends up behind a select of addresses and lives in scratch memory, a store and two loads per priced record
                const float cu = in ? screen_of(cn) : FINF;
                const float lo = __builtin_fminf(m1, cu), hi = __builtin_fmaxf(m1, cu);
                m1 = lo;
                m2 = __builtin_fminf(m2, hi);
            }
            dirty = true;
        };
        auto tighten = [&]() {  // the screen follows (an upper bound of) the wave's second cheapest so far
            const float w1 = wave_min_f32_nonneg(m1);
            const float w2 = wave_min_f32_nonneg(m1 == w1 ? m2 : m1);
            T = w2 < boundf ? w2 : boundf;
            Thi = hi_of((double)T);
        };
        // The records of all touched cells as ONE stream: lane l of a step takes record 64 * step + l of the concatenation of
        // the cells' arrays, so a step is 64 live records whatever the fill of the single cells (a cell holds 10 - 30 nodes
        // at these densities: cell by cell three lanes in four would idle).  Exclusive prefix sum of the fill counts over the
        // lanes; a lane finds its cell by bisection over that prefix (ds_bpermute: the prefix stays in registers).  The cells of
        // the ball's bounding box are taken 64 at a time (one slab unless the radius is far beyond the cell size).
        for (int cbase = 0; cbase < ncr; cbase += 64) {
            uint32_t tcnt = 0, toff = 0;  // lane c: fill count and record offset of cell cbase + c
            if (cbase + lane < ncr) {
                const int ci = cbase + lane, ccx = cx0 + ci / ny, ccy = cy0 + ci % ny, cell = ccx * ncy + ccy;
#ifndef RRT_NO_CELL_CULL
                // a cell beyond the radius holds no hit (the corners of the box)
                const int xl = ccx << cshift, xh = xl + (1 << cshift) - 1, yl = ccy << cshift, yh = yl + (1 << cshift) - 1;
                const int ddx = x < xl ? xl - x : (x > xh ? x - xh : 0), ddy = y < yl ? yl - y : (y > yh ? y - yh : 0);
                tcnt = (uint32_t)(ddx * ddx + ddy * ddy) < r2h ? cellcnt[cell] : 0u;
#else
                tcnt = cellcnt[cell];
#endif
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
            const uint32_t pre = incl - tcnt;  // lanes past the last cell hold `total`: never <= a live record number
            int cur_c = 0;  // (SCATTER) the cell the last step ended in
            for (uint32_t base = (uint32_t)part * (64u * CG); base < total; base += (uint32_t)nparts * (64u * CG)) {  // CG steps in flight
                u32x4 rc[CG];
                bool live[CG];
#pragma unroll
                for (int g2 = 0; g2 < CG; ++g2) {
                    const uint32_t idx = base + 64u * (uint32_t)g2 + (uint32_t)lane;
                    uint32_t lo = 0;  // the largest cell c with pre[c] <= idx (an empty cell shares its prefix with its successor)
                    if constexpr (SCATTER) {
                        // one wave walks the whole stream step by step: the cells that begin in this step write their number into the
                        // slot of their first record (64 LDS words per wave), a running maximum over the lanes carries it on; the
                        // lanes in front of the step's first cell start belong to the cell the last step ended in (rrt_pipe.h)
                        volatile RRT_LDS uint32_t *slots = (volatile RRT_LDS uint32_t *)L.slots[wave];
                        slots[lane] = NONE;
                        const uint32_t rel = pre - (base + 64u * (uint32_t)g2);
                        __builtin_amdgcn_wave_barrier();
                        if (tcnt != 0u && rel < 64u) slots[rel] = (uint32_t)lane;
                        __builtin_amdgcn_wave_barrier();
                        int cv = (int)slots[lane];  // (NONE = -1)
                        cv = max(cv, __builtin_amdgcn_update_dpp(-1, cv, 0x111, 0xf, 0xf, false));
                        cv = max(cv, __builtin_amdgcn_update_dpp(-1, cv, 0x112, 0xf, 0xf, false));
                        cv = max(cv, __builtin_amdgcn_update_dpp(-1, cv, 0x114, 0xf, 0xf, false));
                        cv = max(cv, __builtin_amdgcn_update_dpp(-1, cv, 0x118, 0xf, 0xf, false));
                        cv = max(cv, __builtin_amdgcn_update_dpp(-1, cv, 0x142, 0xa, 0xf, false));
                        cv = max(cv, __builtin_amdgcn_update_dpp(-1, cv, 0x143, 0xc, 0xf, false));
                        cv = cv < 0 ? cur_c : cv;
                        cur_c = __builtin_amdgcn_readlane(cv, 63);
                        lo = (uint32_t)cv;
                    } else {
#pragma unroll
                        for (uint32_t bit = 32; bit != 0; bit >>= 1) {
                            const uint32_t cand = lo + bit;
                            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(cand << 2), (int)pre);
                            lo = v <= idx ? cand : lo;
                        }
                    }
                    const uint32_t cpre = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lo << 2), (int)pre);
                    const uint32_t coff = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lo << 2), (int)toff);
                    live[g2] = idx < total;
                    rc[g2] = ld_rec<COH>(cellrec + (live[g2] ? coff + (idx - cpre) : 0u));  // {xy, index, vcost}; (record 0 exists: every cell array has a slot)
                }
                bool dirty = false;
#pragma unroll
                for (int g2 = 0; g2 < CG; ++g2) {
                    if (base + 64u * (uint32_t)g2 >= total) continue;
                    if (!live[g2]) rc[g2].x = 0x7fff7fffu;  // never within the radius
                    eval_record(rc[g2], dirty);
                }
                if (dirty) tighten();
            }
        }
        nnear_part = wave_sum_u32(hits);
        top2_reduce_xy(tt, fx1, fx2, sx.x1, sx.x2);
        nn_d2 = ld2;
        nn_idx = lidx;
        wave_min_key_idx(nn_d2, nn_idx);
        {
            const u64 mn = __ballot(lidx != NONE && ld2 == nn_d2 && lidx == nn_idx);
            const int ln_ = mn ? (int)__builtin_ctzll(mn) : 0;
            sx.nn_xy = (uint32_t)__builtin_amdgcn_readlane((int)lxy, ln_);
            sx.nn_vlo = (uint32_t)__builtin_amdgcn_readlane((int)lvlo, ln_);
            sx.nn_vhi = (uint32_t)__builtin_amdgcn_readlane((int)lvhi, ln_);
        }
    };

    // Price this wave's parked entries once and compact the ones still open (cost < bound, key >= lower bound) to the front
    // of the list as {index, cells, cost}, then test them.  A sample behind a wall has dozens of cheaper-but-blocked
    // candidates; two per memory round trip made it the straggler of its block:
    // every open entry is tested, one line of sight per LANE.
    // (wc, wi) = the cheapest passing entry; every entry with a key up to it has been tested and holds its cell count.
    // amin: a single-precision lower bound of the cheapest parked entry that is NOT below the bound (+inf: none).
    auto consume_price = [&](uint32_t X, double bound, double lbc, uint32_t lbi, uint32_t nlist, uint32_t &nval, float &amin) {
        const float boundf = screen_of(bound);
        nval = 0;
        float am = FINF;
        for (uint32_t p0 = 0; p0 < nlist; p0 += 64) {
            const uint32_t p = p0 + (uint32_t)lane;
            u32x4 e = {NONE, 0u, 0u, 0u};
            if (p < nlist) e = lget(p);  // {index, xy, vcost}
            const double V = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
            const uint32_t ed2 = dist2(e.y, X);
            double cn = f64_inf();
            if (p < nlist) {
                const float cf = (float)V + __builtin_amdgcn_sqrtf((float)ed2);
                bool below = false;  // certainly below the bound (such an entry is open, or one of the tested ones under the lower bound)
                if (cf < boundf) {
                    const double c = V + sqrt_u24(ed2);
                    below = c < bound;
                    if (below && !key_lt(c, e.x, lbc, lbi)) cn = c;
                }
                if (!below) am = __builtin_fminf(am, __builtin_fmaxf(cf * (1.0f - 1.0e-6f) - 4.0e-3f, 0.0f));
            }
            const bool open = cn < bound;
            const unsigned long long om = __ballot(open);
            if (open) {  // positions at or below the ones this iteration has read
                const unsigned long long cb = (unsigned long long)__double_as_longlong(cn);
                // {index, xy, cost}: the test of an entry reads its coordinates from the second word and leaves the cells it read there
                lput(nval + (uint32_t)__builtin_popcountll(om & ((1ull << lane) - 1ull)), u32x4{e.x, e.y, (uint32_t)cb, (uint32_t)(cb >> 32)});
            }
            nval += (uint32_t)__builtin_popcountll(om);
        }
        amin = wave_min_f32_nonneg(am);
#if defined(RRT_STAMPS) && defined(RRT_STAMPS_OWNER)
        if (PIPE && g == 1 && t == 0) {
            L.dbg[9] += nval;
            L.dbg[10] += 1;
            L.dbg[11] += nlist;
        }
#endif
    };
    auto consume_walk = [&](uint32_t X, uint32_t nval, double &wc, uint32_t &wi) {
        wc = f64_inf();
        wi = NONE;
        if (nval == 0) return;  // every entry was tried
#ifndef RRT_WALKB
#define RRT_WALKB 8
#endif
        if (nval <= (uint32_t)RRT_WALKB && rad < 64) {
            // a handful (a wave's share of a group's ball, a committer's second search): four segments per memory round trip,
            // 64 cells of each at once, instead of lane-by-lane walks of up to four round trips
            for (uint32_t b0 = 0; b0 < nval; b0 += 4) {
                const uint32_t pq = b0 + (uint32_t)(lane & 3);
                u32x4 e = {NONE, 0u, 0u, 0x7ff00000u};
                if (pq < nval) e = lget(pq);
                const uint32_t axy = pq < nval ? e.y : X;
                uint32_t a4[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) a4[q4] = (uint32_t)__builtin_amdgcn_readlane((int)axy, q4);
                bool ok4[4];
                int cells4[4];
                los_batch_n<4>(og, H, a4, (int)(nval - b0 < 4u ? nval - b0 : 4u), X, lane, ok4, cells4);
                uint32_t res = 0;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    if (lane == q4) res = (uint32_t)cells4[q4] | (ok4[q4] ? 0u : 0x80000000u);
                double cn = f64_inf();
                uint32_t ci = NONE;
                if (lane < 4 && pq < nval) {
                    lput_y(pq, res & 0x7fffffffu);
                    if ((res >> 31) == 0u) {
                        cn = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                        ci = e.x;
                    }
                }
                wave_min_f64_idx(cn, ci);
                if (ci != NONE && key_lt(cn, ci, wc, wi)) {
                    wc = cn;
                    wi = ci;
                }
            }
            return;
        }
        // one line of sight PER LANE, 64 entries at a time, every open entry tested (each lane walks its own segment, 16 cell
        // loads in flight); the answer is the cheapest passing entry.  (Ranking up to 16 open entries and testing them in
        // (cost, index) order, one wave per line of sight and 8 in flight, was measured slower on every bench workload:
        // profiles/r02_experiments.md.)
        // A short list spreads every segment over 2 or 4 lanes (16-cell pieces dealt round-robin), so that up to 16 / 32 open
        // entries are through after one / two memory round trips instead of four.
        for (uint32_t p0 = 0; p0 < nval;) {
            const uint32_t rem = nval - p0;
            // log2 of the lanes per entry (single CUs only: on teams the same split measured slower, profiles/r02_experiments.md)
            const int sh = G > 1 ? 0 : (rem <= 16 ? 2 : (rem <= 32 ? 1 : 0));
            const int S = 1 << sh, part = lane & (S - 1);
            const uint32_t p = p0 + ((uint32_t)lane >> sh);
            const bool have = p < nval;
            u32x4 e = {NONE, 0u, 0u, 0x7ff00000u};
            if (have) e = lget(p);
            const uint32_t axy = have ? e.y : X;
            const rrt_line_t ln = rrt_line_setup(ux(axy), uy(axy), ux(X), uy(X));
            const int L = ln.major;
            constexpr int WU = 16, NOHIT = 0x7fffffff;
            int fb = NOHIT;  // first blocked cell among this lane's pieces
            for (int it = 0;; ++it) {
                const unsigned long long hitm = __ballot(fb != NOHIT);
                const bool grp_hit = ((hitm >> (lane & ~(S - 1))) & ((1ull << S) - 1ull)) != 0;
                if (!__any(!grp_hit && it * S * WU <= L)) break;  // (after a full round every cell below (it S WU) has been read)
                const int k0 = (it * S + part) * WU;
                uint8_t v[WU];
#pragma unroll
                for (int u = 0; u < WU; ++u) {  // unconditional loads (clamped to the segment's last cell)
                    const int kk = (k0 + u) < L ? (k0 + u) : L;
                    int x, y;
                    rrt_line_cell(&ln, kk, &x, &y);
                    v[u] = og[(uint32_t)(x * H + y)];
                }
#pragma unroll
                for (int u = 0; u < WU; ++u)
                    if (fb == NOHIT && k0 + u <= L && v[u] != 0) fb = k0 + u;
            }
            if (sh >= 1) {
                const int o = __builtin_amdgcn_update_dpp(NOHIT, fb, 0xB1, 0xf, 0xf, false);  // lane ^ 1
                fb = o < fb ? o : fb;
            }
            if (sh >= 2) {
                const int o = __builtin_amdgcn_update_dpp(NOHIT, fb, 0x4E, 0xf, 0xf, false);  // lane ^ 2
                fb = o < fb ? o : fb;
            }
            const bool blocked = fb != NOHIT;
            const int cells = blocked ? fb + 1 : L + 1;
            if (have && part == 0) lput_y(p, (uint32_t)cells);  // cells read by this test, for count_tests
            double cn = f64_inf();
            uint32_t ci = NONE;
            if (have && part == 0 && !blocked) {
                cn = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                ci = e.x;
            }
            p0 += 64u >> sh;
            wave_min_f64_idx(cn, ci);
            if (ci != NONE && key_lt(cn, ci, wc, wi)) {
                wc = cn;
                wi = ci;
            }
        }
    };

    // One line of sight PER LANE (a -> X for the lanes that `have` one), 16 cell loads in flight per lane and pass: the walk consume_walk
    // makes over a long list, for entries a caller holds in its lanes.  cells as in los_wave.
    auto los_per_lane = [&](bool have, uint32_t axy, uint32_t X, bool &blocked, int &cells) {
        const rrt_line_t ln = rrt_line_setup(ux(have ? axy : X), uy(have ? axy : X), ux(X), uy(X));
        const int L = ln.major;
        constexpr int WU = 16, NOHIT = 0x7fffffff;
        int fb = NOHIT;
        for (int it = 0;; ++it) {
            if (!__any(have && fb == NOHIT && it * WU <= L)) break;
            const int k0 = it * WU;
            uint8_t v[WU];
#pragma unroll
            for (int u = 0; u < WU; ++u) {
                const int kk = (k0 + u) < L ? (k0 + u) : L;
                int x, y;
                rrt_line_cell(&ln, kk, &x, &y);
                v[u] = og[(uint32_t)(x * H + y)];
            }
#pragma unroll
            for (int u = 0; u < WU; ++u)
                if (fb == NOHIT && k0 + u <= L && v[u] != 0) fb = k0 + u;
        }
        blocked = fb != NOHIT;
        cells = blocked ? fb + 1 : L + 1;
    };

    auto consume_list = [&](uint32_t X, double bound, double lbc, uint32_t lbi, uint32_t nlist, double &wc, uint32_t &wi, uint32_t &nval,
                            float &amin) {
        consume_price(X, bound, lbc, lbi, nlist, nval, amin);
        consume_walk(X, nval, wc, wi);
    };

    // The tests the sequential loop makes over a consumed list: up to and including the first passing entry (wc, wi), or all.
    auto count_tests = [&](uint32_t nval, double wc, uint32_t wi, uint32_t &ntests, uint32_t &tcells) {
        uint32_t nt = 0, tcl = 0;
        for (uint32_t p = (uint32_t)lane; p < nval; p += 64) {
            const u32x4 e = lget(p);
            const double cn = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
            if (wi == NONE || !key_lt(wc, wi, cn, e.x)) {
                nt += 1;
                tcl += e.y;
            }
        }
        ntests += wave_sum_u32(nt);
        tcells += wave_sum_u32(tcl);
    };

    // One whole wave on its own.  Returns (pc, pi) or (inf, NONE); nnear = |within| over nodes [0, j0).
    auto snapshot_parent = [&](uint32_t X, int j0, bool check_j0, double bound, double &pc, uint32_t &pi, uint32_t &nnear,
                               uint32_t &ntests, uint32_t &tcells, double lbc = -1.0, uint32_t lbi = 0) {
        pc = f64_inf();
        pi = NONE;
        nnear = 0;
        if (r2 == 0) return;
        for (;;) {
            Top2 tt;
            uint32_t nlist = 0;
            uint32_t nd2, nidx;
            stream_cells(X, j0, check_j0, bound, lbc, lbi, 0, 1, tt, nnear, nlist, nd2, nidx);
            if (tt.i1 == NONE) return;
            // the two cheapest, both lines of sight in flight together (rrt.py:519); the second counts only if needed
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
            lbc = tt.c2;
            lbi = tt.i2 + 1;
            if (nlist > clist_cap) continue;  // the list overflowed: stream the cells again above the new lower bound
            uint32_t nval = 0;
            float am_;
            consume_list(X, bound, lbc, lbi, nlist, pc, pi, nval, am_);
            count_tests(nval, pc, pi, ntests, tcells);
            return;
        }
    };

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
        if (worker && !void_blk) {
        uint32_t xs16[BSA];
#pragma unroll
        for (int k = 0; k < BSA; ++k) {  // this member's samples [BSM wg, BSM (wg + 1))
            const int sk = wg * BSM + k;
            uint32_t X = (uint32_t)__builtin_amdgcn_readlane((int)xv, sk);
            if (sk >= nb) X = (uint32_t)__builtin_amdgcn_readlane((int)xv, 0);
            xs16[k] = X << 4;
        }

        // ---------------- A: scan the snapshot for all samples of the block (not when the owners take the nearest node from
        //                  the record stream) ----------------
        const int nsteps = (j0 + CHUNK - 1) / CHUNK;
        if (!grid_nn) {
            uint32_t best[BSA];
            int c_first = 0;
            if (G > 1 && pre_i == i0) {  // steps [0, pre_j) were scanned while member 0 committed; unfilled slots held node 0
                c_first = pre_j / CHUNK;
#pragma unroll
                for (int k = 0; k < BSA; ++k) best[k] = pre_best[k];
            } else {
#pragma unroll
                for (int k = 0; k < BSA; ++k) best[k] = NONE;
            }
            scan_steps(c_first, nsteps, j0, xs16, best);
            // per sample: wave minimum of d2, lowest index among the lanes that hold it (a lane's best key already
            // carries its lowest such index); gathered into lanes 0..15
            uint32_t gd = NONE, gi = NONE;
#pragma unroll
            for (int k = 0; k < BSA; ++k) {
                const uint32_t key = best[k];
                const uint32_t d2m = wave_min_u32(key) >> 8;
                const uint32_t tag = key & 0xffu;
                uint32_t ki = (tag >> 2) * (uint32_t)CHUNK + 4u * (uint32_t)t + (tag & 3u);
                const unsigned long long tie = __ballot((key >> 8) == d2m);
                if (__builtin_popcountll(tie) == 1)
                    ki = (uint32_t)__builtin_amdgcn_readlane((int)ki, (int)__builtin_ctzll(tie));
                else
                    ki = wave_min_u32((key >> 8) == d2m ? ki : NONE);
                if (lane == k) {
                    gd = d2m;
                    gi = ki;
                }
            }
            if (lane < BSA) {
                u32x2 v = {gd, gi};
                ((RRT_LDS u32x2 *)nnx)[lane * NWAVE + wave] = v;
            }
        }
        if (t < SB) xq_lds[t] = xv;  // lane s: sample s (s < nb)
        if (WIDE && t == 0) L.tick = NWAVE;
        STAMP(0);
        __syncthreads();
        STAMP(1);

        // ---------------- B: owner phase, wave k resolves sample k against the snapshot ----------------
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
        const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
        uint32_t my_open = 0, my_ntests = 0, my_tcells = 0;  // single-wave owner of a blocked sample: see below
        int my_sidx = -1;  // the blocked sample whose candidates wait in this wave's list
        auto take_ticket = [&]() -> int {
            uint32_t tk = 0;
            if (lane == 0) tk = __hip_atomic_fetch_add(&L.tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return __builtin_amdgcn_readfirstlane((int)tk);
        };
        for (int pass = 0;; ++pass) {  // (once, except WIDE with samples left behind blocked ones)
        my_open = 0;
        if constexpr (WPS == 1) {
        // A sample whose two cheapest candidates are blocked leaves its open candidates in its list; behind the barrier ALL waves
        // of the workgroup test them, four lines of sight per wave and memory round trip (one wave on its own walks them lane by
        // lane, four round trips, while the other fifteen wait for it: it was a third of a single-wave owner's block).
        const bool coop = grid_nn && rad < 64;  // (every candidate segment shorter than 64 cells)
        // wave k resolves sample k of the member's share; with more than 16 samples per member (WIDE) whoever is through takes the
        // next one off L.tick -- except a wave that holds a blocked sample: its list has to stay as it is until all waves have
        // tested the candidates (behind the barrier), after which the waves come back for whatever is left of the share
        int k = wave;
        if (WIDE && pass > 0) k = take_ticket();
        for (; k < BSM;) {
            const int sidx = wg * BSM + k;
            if (sidx >= nb) break;
            const uint32_t Xk = (uint32_t)__builtin_amdgcn_readlane((int)xv, sidx);  // every wave holds the same xv
            uint32_t d2s = NONE, vs = NONE;
            Top2 tt0;
            tt0.init();
            uint32_t nlist0 = 0, nnear0 = 0;
            WST0();
            if (grid_nn) {  // the record stream first: it names the nearest node unless the ball is empty
                uint32_t nd2, nidx;
                stream_cells(Xk, j0, false, f64_inf(), -1.0, 0u, 0, 1, tt0, nnear0, nlist0, nd2, nidx);
                WST(4);
                if (nnear0 != 0) {
                    d2s = nd2;
                    vs = nidx;
                } else {
                    wave_scan_nearest(Xk, j0, d2s, vs);
                }
            } else {
                if (lane < NWAVE) {
                    const u32x2 v = ((RRT_LDS u32x2 *)nnx)[k * NWAVE + lane];
                    d2s = v.x;
                    vs = v.y;
                }
                wave_min_key_idx(d2s, vs);
            }
            const double Vs = ld_f64<COH>(&at32(vcost, vs));
            const uint32_t cell = (uint32_t)ux(Xk) * (uint32_t)H + (uint32_t)uy(Xk);
            const uint32_t bm_word = ld_u32<COH>(&at32(bitmap, cell >> 5));
            const uint32_t vsxy = node_xy(vs);
            const LosPending lp = los_issue(og, H, vsxy, Xk, lane);  // finished behind the near-set stream
            // earlier samples of this block that could interact once inserted
            const uint32_t xo = (lane < sidx) ? xq_lds[lane] : Xk;
            const uint32_t dk = dist2(xo, Xk);
            const u64 nnmask = __ballot(lane < sidx && dk < d2s);
            const u64 rmask = __ballot(lane < sidx && star && dk < r2);
            const u64 dupmask = __ballot(lane < sidx && xo == Xk);
            u64 pnn[NPMAX] = {}, pr[NPMAX] = {}, pdup[NPMAX] = {};
            if (PIPE) {  // ... and every sample of the blocks in flight, which are being committed meanwhile
#pragma unroll
                for (int p2 = 0; p2 < NP; ++p2) {
                    if (p2 < nprev) {
                        const uint32_t xop = xqp_lds[p2][lane];
                        const uint32_t dp = dist2(xop, Xk);
                        pnn[p2] = __ballot(dp < d2s);
                        pr[p2] = __ballot(star && dp < r2);
                        pdup[p2] = __ballot(xop == Xk);
                    }
                }
            }
            double pc = f64_inf();
            uint32_t pi = NONE, nnear = 0, ntests = 0, tcells = 0;
            const double cnear_s = Vs + sqrt_u24(d2s);
            WST(5);
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
            const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
#endif
            float amin = 0.0f;
            if (grid_nn) {
                nnear = star ? nnear0 : 0u;
                amin = FINF;  // (an empty ball)
                if (star && nnear0 != 0) finish_parent(Xk, j0, cnear_s, tt0, nlist0, pc, pi, ntests, tcells, amin, coop, my_open);
                my_ntests = ntests;
                my_tcells = tcells;
                if (my_open != 0) my_sidx = sidx;
                if (my_open != 0 && lane == 0) {
                    L.help_n[wave] = my_open;
                    L.help_x[wave] = Xk;
                    L.help_any = 1u;
                }
            } else if (star) {
                snapshot_parent(Xk, j0, false, cnear_s, pc, pi, nnear, ntests, tcells);  // no block node is in the cells yet
            }
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
            wcyc_los += __builtin_amdgcn_s_memtime() - tl0;
#endif
            WST(6);
            int cells = 0;
            const bool free_s = los_finish(lp, og, H, vsxy, Xk, lane, cells);
            WST(7);
            if (lane == 0) {
                BRec r;
                r.d2s = d2s;
                r.vs = vs;
                r.los_s = (free_s ? 0x80000000u : 0u) | (uint32_t)cells;
                r.flags = (bm_word >> (cell & 31)) & 1u;
                r.Vs = Vs;
                r.cbest = (pi != NONE) ? pc : cnear_s;
                r.vbest = (pi != NONE) ? pi : vs;
                r.pstat = (ntests << 20) | (tcells & 0xfffffu);
                r.nnmask = nnmask;
                r.rmask = rmask;
                r.dupmask = dupmask;
                r.nnear = nnear;
                r.amin = amin;
                r.pc = pc;
#pragma unroll
                for (int p2 = 0; p2 < NPMAX; ++p2) {
                    r.pnn[p2] = pnn[p2];
                    r.pr[p2] = pr[p2];
                    r.pdup[p2] = pdup[p2];
                }
#pragma unroll
                for (int w = 0; w < PL_WORDS; ++w) r.plist[w] = 0;  // (no list: flags bits 8-15 are zero)
                brec[0][sidx] = r;
            }
            if (!WIDE || my_open != 0) break;
            k = take_ticket();
        }
        } else {
            // ---- a group of WPS waves per sample: every wave streams its share of the cells; the group's first wave (leader)
            //      combines, tests lines of sight and writes the record; blocked-candidate lists are tested by all WPS waves ----
            const int sl = wave / WPS, part = wave % WPS;
            const int sidx = wg * BSM + sl;
            const bool act = sidx < nb, lead = part == 0;
            const uint32_t Xk = (uint32_t)__builtin_amdgcn_readlane((int)xv, act ? sidx : 0);
            uint32_t d2s = NONE, vs = NONE;
            double Vs = 0.0, cnear_s = f64_inf();  // (grid_nn: known to the leader after the stream, to the others from gctl)
            if (!grid_nn) {
                if (lane < NWAVE) {
                    const u32x2 v = ((RRT_LDS u32x2 *)nnx)[sl * NWAVE + lane];
                    d2s = v.x;
                    vs = v.y;
                }
                wave_min_key_idx(d2s, vs);
                Vs = act ? ld_f64<COH>(&at32(vcost, vs)) : 0.0;
                cnear_s = Vs + sqrt_u24(d2s);
            }
            double pc = f64_inf();
            uint32_t pi = NONE, nnear = 0, ntests = 0, tcells = 0;
            float amin = 0.0f;  // leader: see finish_parent
            bool free_s = false;
            int cells = 0;
            uint32_t bm_word = 0;
            u64 nnmask = 0, rmask = 0, dupmask = 0, pnn[NPMAX] = {}, pr[NPMAX] = {}, pdup[NPMAX] = {};
            const uint32_t cell = (uint32_t)ux(Xk) * (uint32_t)H + (uint32_t)uy(Xk);
            uint32_t own_nlist = 0;
            uint32_t vsxy = Xk;
            LosPending lp;
            lp.major = 0;
            lp.v = 0;
            // the sample's word of the `sampled` bitmap: an agent-scope load that is a memory round trip where the grid's cells are L2
            // hits -- asked for here, in front of the near-set stream, not between the stream and the lines of sight (loads return
            // in order: there the tests' answers waited for it)
            if (lead && act) bm_word = ld_u32<COH>(&at32(bitmap, cell >> 5));
            if (lead && act && !grid_nn) {  // started here, finished behind the near-set stream
                vsxy = node_xy(vs);
                lp = los_issue(og, H, vsxy, Xk, lane);
            }
            if (star || grid_nn) {
                Top2 tt;
                tt.init();
                uint32_t hp = 0, nd2 = NONE, nidx = NONE;
                if (act) stream_cells(Xk, j0, false, cnear_s, -1.0, 0u, part, WPS, tt, hp, own_nlist, nd2, nidx);  // (grid_nn: no bound yet)
#ifdef RRT_STAMPS
                gs0 = __builtin_amdgcn_s_memtime();
#endif
                if (lane == 0) {
                    GSlot sl_;
                    sl_.c1 = tt.c1;
                    sl_.c2 = tt.c2;
                    sl_.i1 = tt.i1;
                    sl_.i2 = tt.i2;
                    sl_.hits = hp;
                    sl_.nlist = own_nlist;
                    sl_.nn_d2 = nd2;
                    sl_.nn_idx = nidx;
                    sl_.pad[0] = sl_.pad[1] = 0;
                    sl_.x1 = sx.x1;
                    sl_.x2 = sx.x2;
                    sl_.nn_xy = sx.nn_xy;
                    sl_.nn_vlo = sx.nn_vlo;
                    sl_.nn_vhi = sx.nn_vhi;
                    gslot[wave] = sl_;
                }
                __syncthreads();
            }
            if constexpr (FASTL) {
                // While the group's leader combines the shares and tests the snapshot's candidates, the other waves test the lines of
                // sight from the samples IN FLIGHT within r_rewire (the two blocks ahead of the commit, the earlier samples of this
                // block) to this sample, four per wave and memory round trip: should one of them be inserted and turn out the cheaper
                // parent, the committer has the answer (rrt.py:519) in the record.  Entry e = the e-th such sample, oldest block first.
                if (!lead && act && star && rad < 64) {
                    const u64 below = (1ull << lane) - 1ull;
                    uint32_t xo_[NP + 1], base = 0;
                    int ent_[NP + 1];
#pragma unroll
                    for (int s_ = 0; s_ <= NP; ++s_) {  // set 0 = the oldest previous block ... NP = this block
                        uint32_t xo;
                        bool in;
                        if (s_ < NP) {
                            xo = xqp_lds[NP - 1 - s_][lane];
                            in = (NP - 1 - s_) < nprev && dist2(xo, Xk) < r2;
                        } else {
                            xo = (lane < sidx) ? xq_lds[lane] : Xk;
                            in = lane < sidx && dist2(xo, Xk) < r2;
                        }
                        const u64 m = __ballot(in);
                        ent_[s_] = in ? (int)(base + (uint32_t)__builtin_popcountll(m & below)) : -1;
                        xo_[s_] = xo;
                        base += (uint32_t)__builtin_popcountll(m);
                    }
                    const uint32_t total = base < (uint32_t)PL_MAX ? base : (uint32_t)PL_MAX;
                    for (uint32_t e0 = 4u * (uint32_t)(part - 1); e0 < total; e0 += 4u * (uint32_t)(WPS - 1)) {
                        uint32_t a4[4], id4[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            a4[c] = Xk;
                            id4[c] = 0;
#pragma unroll
                            for (int s_ = 0; s_ <= NP; ++s_) {
                                const u64 hm = __ballot(ent_[s_] == (int)(e0 + (uint32_t)c));
                                if (hm != 0) {
                                    const int hl = (int)__builtin_ctzll(hm);
                                    a4[c] = (uint32_t)__builtin_amdgcn_readlane((int)xo_[s_], hl);
                                    id4[c] = (uint32_t)(s_ * 64 + hl);
                                }
                            }
                        }
                        const int nc = (int)(total - e0 < 4u ? total - e0 : 4u);
                        bool ok4[4];
                        int cells4[4];
                        los_batch_n<4>(og, H, a4, nc, Xk, lane, ok4, cells4);
                        uint32_t ent = 0;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (lane == c) ent = id4[c] | ((uint32_t)cells4[c] << 8) | (ok4[c] ? 0x8000u : 0u);
                        if (lane < nc) L.plist[sl][e0 + (uint32_t)lane] = (uint16_t)ent;
                    }
                }
            }
#ifdef RRT_STAMPS
            gs1 = __builtin_amdgcn_s_memtime();
#endif
            bool consume = false;
            double lbc = -1.0;
            uint32_t lbi = 0;
            if (lead && act) {
                // (grid_nn) ONE memory round trip for the leader: the shares' slots hold the coordinates and the vcost of the nearest hit
                // and the coordinates of their two cheapest entries (all from the records the stream read anyway), so the line of sight
                // from the nearest vertex and those from the group's two cheapest candidates go out together.  (Round 3: the nearest's
                // cost and coordinates were fetched first, then its line of sight, then the candidates' coordinates, then theirs.)
                Top2 tt;
                tt.init();
                uint32_t tx1 = 0, tx2 = 0;
                bool overflow = false, tested3 = false, ok1 = false, ok2 = false;
                int cc1 = 0, cc2 = 0;
                if (grid_nn) {
                    uint32_t gd = NONE, gi = NONE, gh = 0, gxy = 0, gvlo = 0, gvhi = 0, fx1 = 0, fx2 = 0;
                    bool ovf_l = false;
                    if (lane < WPS) {
                        const GSlot o = gslot[sl * WPS + lane];
                        gd = o.nn_d2;
                        gi = o.nn_idx;
                        gh = o.hits;
                        gxy = o.nn_xy;
                        gvlo = o.nn_vlo;
                        gvhi = o.nn_vhi;
                        tt.c1 = o.c1;
                        tt.i1 = o.i1;
                        tt.c2 = o.c2;
                        tt.i2 = o.i2;
                        fx1 = o.x1;
                        fx2 = o.x2;
                        ovf_l = o.nlist > clist_cap;
                    }
                    const uint32_t gd_l = gd, gi_l = gi;
                    wave_min_key_idx(gd, gi);
                    const uint32_t hits_all = wave_sum_u32(gh);
                    overflow = __ballot(ovf_l) != 0;
                    if (star) {
                        top2_reduce_xy(tt, fx1, fx2, tx1, tx2);
                        nnear += hits_all;
                    } else {
                        tt.init();
                    }
                    if (hits_all != 0) {  // the nearest of the shares' hits is the nearest node; its record named its place and its cost
                        d2s = gd;
                        vs = gi;
                        const u64 mn = __ballot(gi_l != NONE && gd_l == gd && gi_l == gi);
                        const int ln_ = (int)__builtin_ctzll(mn);
                        vsxy = (uint32_t)__builtin_amdgcn_readlane((int)gxy, ln_);
                        Vs = __longlong_as_double((long long)(((u64)(uint32_t)__builtin_amdgcn_readlane((int)gvhi, ln_) << 32) |
                                                              (uint32_t)__builtin_amdgcn_readlane((int)gvlo, ln_)));
                    } else {  // an empty ball: this wave scans the nodes
                        wave_scan_nearest(Xk, j0, d2s, vs);
                        Vs = ld_f64<COH>(&at32(vcost, vs));
                        vsxy = node_xy(vs);
                    }
                    cnear_s = Vs + sqrt_u24(d2s);
                    if (star) {
                        // the cheapest entry at or above the bound, where the two cheapest of the ball tell (see finish_parent)
                        amin = (tt.i1 == NONE) ? FINF : !(tt.c1 < cnear_s) ? lower_f32(tt.c1) : (tt.i2 == NONE) ? FINF : !(tt.c2 < cnear_s) ? lower_f32(tt.c2) : 0.0f;
                        if (tt.i2 != NONE && !(tt.c2 < cnear_s)) {
                            tt.i2 = NONE;
                            tt.c2 = f64_inf();
                        }
                        if (tt.i1 != NONE && !(tt.c1 < cnear_s)) tt.init();
                    }
                    if (hits_all != 0 && rad < 64 && !overflow) {  // (every segment shorter than 64 cells: one cell per lane)
                        const uint32_t a3[3] = {vsxy, tt.i1 != NONE ? tx1 : Xk, tt.i2 != NONE ? tx2 : Xk};
                        bool ok3[3];
                        int cells3[3];
                        los_batch_n<3>(og, H, a3, tt.i2 != NONE ? 3 : (tt.i1 != NONE ? 2 : 1), Xk, lane, ok3, cells3);
                        free_s = ok3[0];
                        cells = cells3[0];
                        ok1 = ok3[1];
                        cc1 = cells3[1];
                        ok2 = ok3[2];
                        cc2 = cells3[2];
                        tested3 = true;
                    } else {
                        lp = los_issue(og, H, vsxy, Xk, lane);
                    }
                } else if (star) {  // (the nearest came from phase A: the shares streamed under the bound)
                    uint32_t hits_l = 0, fx1 = 0, fx2 = 0;
                    bool ovf_l = false;
                    if (lane < WPS) {
                        const GSlot o = gslot[sl * WPS + lane];
                        tt.c1 = o.c1;
                        tt.i1 = o.i1;
                        tt.c2 = o.c2;
                        tt.i2 = o.i2;
                        fx1 = o.x1;
                        fx2 = o.x2;
                        hits_l = o.hits;
                        ovf_l = o.nlist > clist_cap;
                    }
                    top2_reduce_xy(tt, fx1, fx2, tx1, tx2);
                    nnear += wave_sum_u32(hits_l);
                    overflow = __ballot(ovf_l) != 0;
                    if (tt.i2 != NONE && !(tt.c2 < cnear_s)) {
                        tt.i2 = NONE;
                        tt.c2 = f64_inf();
                    }
                    if (tt.i1 != NONE && !(tt.c1 < cnear_s)) tt.init();
                }
                if (!tested3) free_s = los_finish(lp, og, H, vsxy, Xk, lane, cells);
                // earlier samples of this block that could interact once inserted
                const uint32_t xo = (lane < sidx) ? xq_lds[lane] : Xk;
                const uint32_t dk = dist2(xo, Xk);
                nnmask = __ballot(lane < sidx && dk < d2s);
                rmask = __ballot(lane < sidx && star && dk < r2);
                dupmask = __ballot(lane < sidx && xo == Xk);
                if (PIPE) {  // ... and every sample of the blocks in flight, which are being committed meanwhile
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) {
                        if (p2 < nprev) {
                            const uint32_t xop = xqp_lds[p2][lane];
                            const uint32_t dp = dist2(xop, Xk);
                            pnn[p2] = __ballot(dp < d2s);
                            pr[p2] = __ballot(star && dp < r2);
                            pdup[p2] = __ballot(xop == Xk);
                        }
                    }
                }
#ifdef RRT_STAMPS
                gs2 = __builtin_amdgcn_s_memtime();
#endif
                if (star) {
                    if (overflow) {  // a radius far beyond the cell size: this wave resolves the sample on its own
                        nnear = 0;
                        amin = 0.0f;
                        snapshot_parent(Xk, j0, false, cnear_s, pc, pi, nnear, ntests, tcells);
                    } else if (tt.i1 != NONE) {
                        // the two cheapest, both lines of sight in flight together (rrt.py:519); the second counts only if needed
                        if (!tested3) los_wave2(og, H, tx1, tt.i2 != NONE ? tx2 : Xk, tt.i2 != NONE, Xk, lane, ok1, cc1, ok2, cc2);
                        ntests += 1;
                        tcells += (uint32_t)cc1;
                        if (ok1) {
                            pc = tt.c1;
                            pi = tt.i1;
                        } else if (tt.i2 != NONE) {
                            ntests += 1;
                            tcells += (uint32_t)cc2;
                            if (ok2) {
                                pc = tt.c2;
                                pi = tt.i2;
                            } else {
                                consume = true;
                                lbc = tt.c2;
                                lbi = tt.i2 + 1;
                            }
                        }
                    }
                }
            }
#ifdef RRT_STAMPS
            if (lead) gs3 = __builtin_amdgcn_s_memtime();
#endif
            if (star) {
                if (lead && lane == 0) {
                    GCtl c;
                    c.lbc = lbc;
                    c.lbi = lbi;
                    c.consume = (act && consume) ? 1u : 0u;
                    c.bound = cnear_s;
                    gctl[sl] = c;
                }
                __syncthreads();
                const GCtl c = gctl[sl];
                // which sixteenth of the cost range [lbc, bound) an open candidate's cost falls in (monotone in the cost)
                auto qbin = [](double cn, double lo, double hi) -> int {
                    const double w = hi - lo;
                    if (!(w > 0.0)) return 0;
                    const int b = (int)((cn - lo) * (16.0 / w));
                    return b < 0 ? 0 : (b > 15 ? 15 : b);
                };
                bool by_queue = false;
                if constexpr (GQ) {
                    // A sample whose two cheapest candidates are blocked (7 % of the samples: it sits behind a wall, dozens of cheaper
                    // vertices do not see it) was its block's straggler: the open candidates sit in the lists of the few waves whose
                    // share of the record stream held them, and each of those waves walked its list one segment per lane, up to four
                    // dependent passes of sixteen cell loads -- 20 k cycles, with 64 workers in nearly every block, and the committer's
                    // fetch of the next block waits for the last record (profiles/r04_experiments.md).  Now every wave prices its parked
                    // entries and puts the open ones into the group's queue; behind a barrier wave w takes entries 8 w .. 8 w + 7 and
                    // tests all eight in one memory round trip (64 lanes per segment: the radius is below 64 cells here); behind the
                    // next the leader picks the first passing entry in (cost, index) order and counts the tests the sequential walk
                    // makes (rrt.py:515-521).
                    if (rad < 64) {
                        RRT_LDS u32x4 *const oq = (RRT_LDS u32x4 *)&brec[1][0] + (size_t)sl * OQCAP;
#ifdef RRT_STAMPS
                        unsigned long long qt0 = __builtin_amdgcn_s_memtime(), qt1 = qt0, qt2 = qt0, qt3 = qt0, qt4 = qt0;
#endif
                        if (c.consume != 0u) {
                            const float boundf = screen_of(c.bound);
                            float am = FINF;
                            for (uint32_t p0 = 0; p0 < own_nlist; p0 += 64) {
                                const uint32_t p = p0 + (uint32_t)lane;
                                u32x4 e = {NONE, 0u, 0u, 0u};
                                if (p < own_nlist) e = lget(p);  // {index, xy, vcost}
                                const double V = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                                const uint32_t ed2 = dist2(e.y, Xk);
                                double cn = f64_inf();
                                if (p < own_nlist) {
                                    const float cf = (float)V + __builtin_amdgcn_sqrtf((float)ed2);
                                    bool below = false;
                                    if (cf < boundf) {
                                        const double cx = V + sqrt_u24(ed2);
                                        below = cx < c.bound;
                                        if (below && !key_lt(cx, e.x, c.lbc, c.lbi)) cn = cx;
                                    }
                                    if (!below) am = __builtin_fminf(am, __builtin_fmaxf(cf * (1.0f - 1.0e-6f) - 4.0e-3f, 0.0f));
                                }
                                const bool open = cn < c.bound;
                                const unsigned long long om = __ballot(open);
                                if (om != 0) {
                                    const uint32_t cnt = (uint32_t)__builtin_popcountll(om);
                                    uint32_t qb = 0;
                                    if (lane == 0) qb = __hip_atomic_fetch_add(&L.oq_cnt[sl], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    qb = uni32(qb);
                                    if (qb + cnt > OQCAP) {
                                        if (lane == 0) L.oq_ovf = 1u;  // (a radius with more open entries than the queue holds: the lists' own walk)
                                    } else if (open) {
                                        const unsigned long long cb = (unsigned long long)__double_as_longlong(cn);
                                        oq[qb + (uint32_t)__builtin_popcountll(om & ((1ull << lane) - 1ull))] = u32x4{e.x, e.y, (uint32_t)cb, (uint32_t)(cb >> 32)};
                                        if constexpr (BSM == 1) __hip_atomic_fetch_add(&L.qhist[qbin(cn, c.lbc, c.bound)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                }
                            }
                            am = wave_min_f32_nonneg(am);
                            if (lane == 0) gslot[wave].pad[0] = __float_as_uint(am);
                        }
#ifdef RRT_STAMPS
                        qt1 = __builtin_amdgcn_s_memtime();
#endif
                        __syncthreads();
#ifdef RRT_STAMPS
                        qt2 = __builtin_amdgcn_s_memtime();
#endif
                        if (uni32(L.oq_ovf) == 0u) {
                            by_queue = true;
                            const uint32_t nq = c.consume != 0u ? uni32(L.oq_cnt[sl]) : 0u;
                            // A BIG queue (hundreds of open candidates: a dense tree behind a wall) took three round trips of 128 tests, and
                            // the walk of rrt.py:515-521 needs a dozen of them: the candidates in (cost, index) order up to the first one
                            // that sees the sample.  So the cheap ones first: the histogram of the costs (filled while the queue was) names
                            // the sixteenth b* of the cost range below which ~100 candidates lie; stage 1 tests those (entries dealt to the
                            // waves one by one, so that every wave has a few); only if none of them passes the rest follows (stage 2).
                            constexpr uint32_t QBIG = 128;
                            const bool big = BSM == 1 && nq > QBIG;
                            int bstar = 15;
                            if (big) {
                                uint32_t hc = lane < 16 ? L.qhist[lane] : 0u, inc = hc;
                                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
                                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
                                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
                                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);
                                const u64 reach = __ballot(lane < 16 && inc >= 96u);
                                bstar = reach ? (int)__builtin_ctzll(reach) : 15;
                            }
                            // one stage of tests over the entries this wave is dealt (entry w + 16 l in lane l): the selected ones, eight per
                            // round trip
                            auto test_dealt = [&](int blo, int bhi) {
                                for (uint32_t q0 = 0; q0 < nq; q0 += 64u * (uint32_t)WPS) {
                                    const uint32_t qe = q0 + (uint32_t)part + (uint32_t)WPS * (uint32_t)lane;
                                    u32x4 e = {NONE, 0u, 0u, 0u};
                                    bool sel = false;
                                    if (qe < nq) {
                                        e = oq[qe];
                                        const int b = qbin(__longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z)), c.lbc, c.bound);
                                        sel = b >= blo && b <= bhi;
                                    }
                                    u64 sm = __ballot(sel);
                                    uint32_t res = 0;
                                    while (sm) {
                                        uint32_t a8[LOSB];
                                        int l8[LOSB], nc = 0;
#pragma unroll
                                        for (int q8 = 0; q8 < LOSB; ++q8) {
                                            a8[q8] = Xk;
                                            l8[q8] = 0;
                                            if (sm) {
                                                l8[q8] = (int)__builtin_ctzll(sm);
                                                sm &= sm - 1;
                                                a8[q8] = (uint32_t)__builtin_amdgcn_readlane((int)e.y, l8[q8]);
                                                nc = q8 + 1;
                                            }
                                        }
                                        bool ok8[LOSB];
                                        int cells8[LOSB];
                                        los_batch(og, H, a8, nc, Xk, lane, ok8, cells8);
#pragma unroll
                                        for (int q8 = 0; q8 < LOSB; ++q8)
                                            if (q8 < nc && lane == l8[q8]) res = (uint32_t)cells8[q8] | (ok8[q8] ? 0u : 0x80000000u);
                                    }
                                    if (sel) oq[qe].y = res;  // cells read; bit 31: blocked
                                }
                            };
                            if (big) {
                                test_dealt(0, bstar);
                                __syncthreads();
                                if (lead) {  // did one of the cheap candidates pass?  (stage 2 otherwise: everybody has to know)
                                    bool pass = false;
                                    for (uint32_t pq = (uint32_t)lane; pq < nq; pq += 64) {
                                        const u32x4 e = oq[pq];
                                        const int b = qbin(__longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z)), c.lbc, c.bound);
                                        pass = pass || (b <= bstar && (e.y >> 31) == 0u);
                                    }
                                    const bool anyp = __ballot(pass) != 0;
                                    if (lane == 0) L.qstage2 = anyp ? 0u : 1u;
                                }
                                __syncthreads();
                                if (uni32(L.qstage2) != 0u) {
                                    if (bstar < 15) test_dealt(bstar + 1, 15);
                                    bstar = 15;
                                }
                            }
                            for (uint32_t e0 = 8u * (uint32_t)part; !big && e0 < nq; e0 += 8u * (uint32_t)WPS) {
                                const int nc = (int)(nq - e0 < 8u ? nq - e0 : 8u);
                                uint32_t axy = Xk;
                                if (lane < nc) axy = oq[e0 + (uint32_t)lane].y;  // (the entry's coordinates; the result of its test goes there)
                                uint32_t a8[LOSB];
#pragma unroll
                                for (int q8 = 0; q8 < LOSB; ++q8) a8[q8] = (uint32_t)__builtin_amdgcn_readlane((int)axy, q8);
                                bool ok8[LOSB];
                                int cells8[LOSB];
                                los_batch(og, H, a8, nc, Xk, lane, ok8, cells8);
                                uint32_t res = 0;
#pragma unroll
                                for (int q8 = 0; q8 < LOSB; ++q8)
                                    if (lane == q8) res = (uint32_t)cells8[q8] | (ok8[q8] ? 0u : 0x80000000u);
                                if (lane < nc) oq[e0 + (uint32_t)lane].y = res;  // cells read; bit 31: blocked
                            }
#ifdef RRT_STAMPS
                            qt3 = __builtin_amdgcn_s_memtime();
#endif
                            __syncthreads();
#ifdef RRT_STAMPS
                            qt4 = __builtin_amdgcn_s_memtime();
                            if (lead && lane == 0 && c.consume != 0u && g >= 1) {
                                const int o = (qt4 - wres0 > 30000ull) ? 24 : 16;
                                atomicAdd(&D->dbg2[o + 0], 1ull);
                                atomicAdd(&D->dbg2[o + 1], qt0 - gs3);  // gctl + barrier
                                atomicAdd(&D->dbg2[o + 2], qt1 - qt0);  // pricing into the queue
                                atomicAdd(&D->dbg2[o + 3], qt2 - qt1);  // barrier
                                atomicAdd(&D->dbg2[o + 4], qt3 - qt2);  // tests
                                atomicAdd(&D->dbg2[o + 5], qt4 - qt3);  // barrier
                                atomicAdd(&D->dbg2[o + 6], (unsigned long long)nq);
                                atomicAdd(&D->dbg2[o + 7], gs3 - wres0);  // everything in front of the blocked-candidate path
                            }
#endif
                            if (lead && c.consume != 0u) {
                                double wc = f64_inf();
                                uint32_t wi = NONE;
                                for (uint32_t p0 = 0; p0 < nq; p0 += 64) {  // the first passing entry in (cost, index) order
                                    const uint32_t pq = p0 + (uint32_t)lane;
                                    double cn = f64_inf();
                                    uint32_t ci = NONE;
                                    if (pq < nq) {
                                        const u32x4 e = oq[pq];
                                        const double ce = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                                        // (a big queue: only the entries of the sixteenths up to b* were tested; the others still hold their
                                        //  coordinates where the result goes, and none of them can come before a tested entry that passed)
                                        if ((!big || qbin(ce, c.lbc, c.bound) <= bstar) && (e.y >> 31) == 0u) {
                                            cn = ce;
                                            ci = e.x;
                                        }
                                    }
                                    wave_min_f64_idx(cn, ci);
                                    if (ci != NONE && key_lt(cn, ci, wc, wi)) {
                                        wc = cn;
                                        wi = ci;
                                    }
                                }
                                uint32_t nt = 0, tcl = 0;  // the tests the sequential walk makes: up to and including that entry, or all
                                for (uint32_t pq = (uint32_t)lane; pq < nq; pq += 64) {
                                    const u32x4 e = oq[pq];
                                    const double cn = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                                    if ((wi == NONE || !key_lt(wc, wi, cn, e.x)) && (!big || qbin(cn, c.lbc, c.bound) <= bstar)) {
                                        nt += 1;
                                        tcl += e.y & 0x7fffffffu;
                                    }
                                }
                                ntests += wave_sum_u32(nt);
                                tcells += wave_sum_u32(tcl);
                                pc = wc;
                                pi = wi;
                                float am = FINF;
                                if (lane < WPS) am = __uint_as_float(gslot[sl * WPS + lane].pad[0]);
                                amin = grid_nn ? wave_min_f32_nonneg(am) : 0.0f;
                                if (lane == 0) L.oq_cnt[sl] = 0;  // (the next block's waves append behind its own barriers)
                                if (BSM == 1 && lane < 16) L.qhist[lane] = 0;
                                if (BSM == 1 && lane == 0) L.qstage2 = 0;
                            }
                        }
                    }
                }
                if (!by_queue) {
                if (c.consume != 0u) {  // every wave of the group: its own parked entries above the lower bound
                    double wc;
                    uint32_t wi, nval;
                    float wam;
                    consume_list(Xk, c.bound, c.lbc, c.lbi, own_nlist, wc, wi, nval, wam);
                    if (lane == 0) {
                        GSlot sl_;
                        sl_.c1 = wc;
                        sl_.c2 = f64_inf();
                        sl_.i1 = wi;
                        sl_.i2 = NONE;
                        sl_.hits = 0;
                        sl_.nlist = nval;
                        sl_.nn_d2 = sl_.nn_idx = NONE;
                        sl_.pad[0] = __float_as_uint(wam);
                        sl_.pad[1] = 0;
                        gslot[wave] = sl_;
                    }
                }
                __syncthreads();
                if (c.consume != 0u) {
                    // every wave: the first passing entry of the group (lane pp < WPS reads share pp's), then the tests the
                    // sequential loop makes over its own entries; the leader adds the counts up behind one more barrier
                    double gc = f64_inf();
                    uint32_t gi = NONE;
                    if (lane < WPS) {
                        gc = gslot[sl * WPS + lane].c1;
                        gi = gslot[sl * WPS + lane].i1;
                    }
                    wave_min_f64_idx(gc, gi);
                    uint32_t nt = 0, tcl = 0;
                    count_tests(gslot[wave].nlist, gc, gi, nt, tcl);
                    if (lane == 0) {
                        gslot[wave].hits = nt;
                        gslot[wave].i2 = tcl;
                    }
                    if (lead) {
                        pc = gc;
                        pi = gi;
                    }
                }
                __syncthreads();
                if (lead && c.consume != 0u) {
                    uint32_t nt = 0, tcl = 0;
                    float am = FINF;
                    if (lane < WPS) {
                        nt = gslot[sl * WPS + lane].hits;
                        tcl = gslot[sl * WPS + lane].i2;
                        am = __uint_as_float(gslot[sl * WPS + lane].pad[0]);
                    }
                    ntests += wave_sum_u32(nt);
                    tcells += wave_sum_u32(tcl);
                    amin = grid_nn ? wave_min_f32_nonneg(am) : 0.0f;  // (shares that streamed under the bound parked nothing above it)
                }
                if (GQ && lead && lane == 0) {  // (the queue path gave up: overflow)
                    L.oq_cnt[sl] = 0;
                    L.oq_ovf = 0;
                }
                if (GQ && BSM == 1 && lead && lane < 16) L.qhist[lane] = 0;
                }  // !by_queue
            }
#ifdef RRT_STAMPS
            if (lead && act && lane == 0 && PIPE && g >= 1) {  // where a group's time goes: all blocks [8..], blocks of more than 32 k cycles [0..]
                const unsigned long long ge = __builtin_amdgcn_s_memtime();
                const int o = (ge - wres0 > 32000ull) ? 0 : 8;
                atomicAdd(&D->dbg2[o + 0], 1ull);
                atomicAdd(&D->dbg2[o + 1], gs0 - wres0);  // block top .. own share of the stream done
                atomicAdd(&D->dbg2[o + 2], gs1 - gs0);    // .. first barrier passed (the slowest wave's share)
                atomicAdd(&D->dbg2[o + 3], gs2 - gs1);    // .. leader: nearest, its cost, line of sight, masks
                atomicAdd(&D->dbg2[o + 4], gs3 - gs2);    // .. leader: the two cheapest candidates' lines of sight
                atomicAdd(&D->dbg2[o + 5], ge - gs3);     // .. the blocked-candidate barriers and lists
                if (consume) atomicAdd(&D->dbg2[o + 6], 1ull);
                if (nnear == 0) atomicAdd(&D->dbg2[o + 7], 1ull);
            }
#endif
            if (lead && act && lane == 0) {
                BRec r;
                r.d2s = d2s;
                r.vs = vs;
                r.los_s = (free_s ? 0x80000000u : 0u) | (uint32_t)cells;
                r.flags = (bm_word >> (cell & 31)) & 1u;
                r.Vs = Vs;
                r.cbest = (pi != NONE) ? pc : cnear_s;
                r.vbest = (pi != NONE) ? pi : vs;
                r.pstat = (ntests << 20) | (tcells & 0xfffffu);
                r.nnmask = nnmask;
                r.rmask = rmask;
                r.dupmask = dupmask;
                r.nnear = nnear;
                r.amin = amin;
                r.pc = pc;
#pragma unroll
                for (int p2 = 0; p2 < NPMAX; ++p2) {
                    r.pnn[p2] = pnn[p2];
                    r.pr[p2] = pr[p2];
                    r.pdup[p2] = pdup[p2];
                }
#pragma unroll
                for (int w = 0; w < PL_WORDS; ++w) r.plist[w] = 0;
                if constexpr (FASTL) {  // the group's list (its waves filled it in front of the last barriers); a radius beyond the batched test: void
                    uint32_t nent = (uint32_t)__builtin_popcountll(rmask);
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) nent += (uint32_t)__builtin_popcountll(pr[p2]);
                    if (rad >= 64 && nent != 0) nent = 255u;
                    r.flags |= (nent < 255u ? nent : 255u) << 8;
                    const RRT_LDS u64 *pl = (const RRT_LDS u64 *)&L.plist[sl][0];
#pragma unroll
                    for (int w = 0; w < PL_WORDS; ++w) r.plist[w] = pl[w];
                }
                brec[0][sidx] = r;
            }
        }
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
        wcyc_acc += __builtin_amdgcn_s_memtime() - tb0;
#endif
        STAMP(2);
        WST0();
        __syncthreads();
        WST(8);
        if constexpr (WPS == 1) {
            if (uni32(L.help_any) != 0u) {
                for (int w = 0; w < NWAVE; ++w) {
                    const uint32_t n = uni32(L.help_n[w]);
                    if (n == 0) continue;
                    const uint32_t Xw = uni32(L.help_x[w]);
                    for (uint32_t c = (uint32_t)((wave - w) & (NWAVE - 1)); c * 4u < n; c += NWAVE) {  // this wave's groups of four candidates
                        const uint32_t pq = c * 4u + (uint32_t)(lane & 3);
                        uint32_t axy = Xw;
                        if (pq < n) axy = lget_w(w, pq).y;
                        uint32_t a4[4];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) a4[q4] = (uint32_t)__builtin_amdgcn_readlane((int)axy, q4);
                        const int nc = (int)(n - c * 4u < 4u ? n - c * 4u : 4u);
                        bool ok4[4];
                        int cells4[4];
                        los_batch_n<4>(og, H, a4, nc, Xw, lane, ok4, cells4);
                        uint32_t res = 0;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4)
                            if (lane == q4) res = (uint32_t)cells4[q4] | (ok4[q4] ? 0u : 0x80000000u);
                        if (lane < nc) lput_y_w(w, c * 4u + (uint32_t)lane, res);  // cells read; bit 31: blocked
                    }
                }
                __syncthreads();
                if (my_open != 0) {  // the owner: the cheapest passing candidate, the tests the sequential loop makes, the record
                    double wc = f64_inf();
                    uint32_t wi = NONE;
                    for (uint32_t p0 = 0; p0 < my_open; p0 += 64) {
                        const uint32_t pq = p0 + (uint32_t)lane;
                        double cn = f64_inf();
                        uint32_t ci = NONE;
                        if (pq < my_open) {
                            const u32x4 e = lget(pq);
                            lput_y(pq, e.y & 0x7fffffffu);
                            if ((e.y >> 31) == 0u) {
                                cn = __longlong_as_double((long long)(((unsigned long long)e.w << 32) | e.z));
                                ci = e.x;
                            }
                        }
                        wave_min_f64_idx(cn, ci);
                        if (ci != NONE && key_lt(cn, ci, wc, wi)) {
                            wc = cn;
                            wi = ci;
                        }
                    }
                    count_tests(my_open, wc, wi, my_ntests, my_tcells);
                    if (lane == 0) {
                        BRec &r = brec[0][my_sidx];
                        if (wi != NONE) {
                            r.cbest = wc;
                            r.vbest = wi;
                            r.pc = wc;
                        }
                        r.pstat = (my_ntests << 20) | (my_tcells & 0xfffffu);
                        L.help_n[wave] = 0;
                    }
                }
                if (t == 0) L.help_any = 0;
                __syncthreads();
            }
        }
        if (!WIDE) break;
        const int taken = (int)uni32(L.tick);  // (every wave reads it before any wave takes another ticket)
        __syncthreads();
        if (taken >= nb - wg * BSM) break;  // every sample of the share has been taken (its wave is past the barrier: resolved)
        }  // pass
        STAMP(3);
        }  // worker

        // ---------------- a pipelined team's workers: hand the records of block s over and, instead of waiting for its commit,
        //                  take the nodes of the commit of block s - 1 and go on with block s + 1 ----------------
        if (PIPE && g > 0) {
#ifdef RRT_STAMPS
            if (t == 0 && g <= 64) {
                const unsigned long long dres = __builtin_amdgcn_s_memtime() - wres0;
                if (dres > 26000ull) D->dbg2[192 + g - 1] += 1;
                if (dres > 32000ull) D->dbg2[256 + g - 1] += 1;
                if (dres > 40000ull) D->dbg2[320 + g - 1] += 1;
                if (dres > D->dbg2[384 + g - 1]) D->dbg2[384 + g - 1] = dres;
            }
#endif
            DBGT(0);
            const bool more = i0 + nb < n;
            const bool take = (int)epoch > LAG && (more || pipe_inf);  // there is a commit to take (an Informed worker always looks)
            if (wave == 0) {
                if (!void_blk) {
                    const RRT_LDS u64 *src = (const RRT_LDS u64 *)&brec[0][wg * BSM];
                    gu64 *dst = t_rec + (size_t)(epoch % NSLOT) * 64 * BREC_WORDS + (size_t)wg * BSM * BREC_WORDS;
                    for (int w = lane; w < BSM * BREC_WORDS; w += 64) __hip_atomic_store(dst + w, src[w], RRT_RLX_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing wave drains, then signals
                }
                if (lane == 0) __hip_atomic_store(t_arrive + 32 * g, epoch, RRT_RLX_AGENT);
                if (g == 1) TSMARK(epoch, 5);
                if (g == 8) TSMARK(epoch, 8);
            }
            // The commit this worker has to take next is some way off: meanwhile it scans the nodes it already has for the NEXT
            // block's samples (known: only an Informed block can be cut short); after the take only the steps that hold new
            // nodes are left.
            pre_i = -1;
            if (!informed && more && !grid_nn) {
                pre_i = i0 + nb;
                pre_j = j0;
                const int nbn = (n - pre_i) < SB ? (n - pre_i) : SB;
                pre_xv = lane < nbn ? at32(samples, (uint32_t)(pre_i + lane)) : 0u;
                uint32_t xsn[BSA];
#pragma unroll
                for (int k = 0; k < BSA; ++k) {
                    const int sk = wg * BSM + k;
                    uint32_t X = (uint32_t)__builtin_amdgcn_readlane((int)pre_xv, sk);
                    if (sk >= nbn) X = (uint32_t)__builtin_amdgcn_readlane((int)pre_xv, 0);
                    xsn[k] = X << 4;
                    pre_best[k] = NONE;
                }
                scan_steps(0, pre_j / CHUNK, pre_j, xsn, pre_best);
            }
            if (wave == 0) {
                bool ok = true;
                DBGT(1);
                if (take) {
                    ok = team_wait(t_go, epoch - LAG, t_fail);
                    if (g == 1) TSMARK(epoch, 6);
                    if (g == 8) TSMARK(epoch, 9);
                    DBGT(2);
                    if (!grid_nn) {  // (the scan of the node array reads it with plain loads)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ONE acquire per workgroup: drops this CU's L1
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ... and holds the barrier until it has completed
                    }
                }
                if (lane == 0) blk.pad0 = ok ? 0 : 1;
#pragma unroll
                for (int p2 = NP - 1; p2 > 0; --p2) xqp_lds[p2][lane] = xqp_lds[p2 - 1][lane];
                xqp_lds[0][lane] = xv;  // the next block's "previous" samples (a block that has a successor is full)
            }
            __syncthreads();
            if (blk.pad0 != 0) {
                team_failed = true;
                break;
            }
            i = i0 + nb;
            nprev = void_blk ? 0 : (nprev < NP ? nprev + 1 : NP);
            if (take) {
                BlkWords u;
#pragma unroll
                for (int w = 0; w < 5; ++w) u.w[w] = __hip_atomic_load(t_state + (size_t)((epoch - LAG) % NSLOT) * 8 + w, RRT_RLX_AGENT);
                if (pipe_inf && ((u.b.pad1 & ST_FLAG_STOP) != 0 || u.b.i >= n)) break;  // the run is over (or waits for the host)
                const int jn = unis32(u.b.j);
                if (t < jn - j0) {  // at most SB new nodes: append them to this CU's node cache and cell fill counts
                    const uint32_t Xn = ld_u32<COH>(&at32(nodes_g, (uint32_t)(j0 + t)));
                    if (j0 + t < lds_nodes) nodes_lds[j0 + t] = Xn;
                    if (cells_on) __hip_atomic_fetch_add(&cellcnt[cell_of(Xn)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                j = jn;
                if (pipe_inf) {
                    nsoln = unis32(u.b.nsoln);
                    vbest_soln = unis32(u.b.vbest_soln);
                    cmin_soln = unif64(u.b.cmin_soln);
                    c_ell = unif64(u.b.c_ell);
                    if ((u.b.pad1 & ST_FLAG_RESTART) != 0) {
                        // that commit ended early or moved the ellipse: the block just handed over is void; wait for the
                        // committer's empty turn, then start over from the true state without a previous block
                        __syncthreads();  // (every wave has read blk.pad0)
                        if (wave == 0) {
                            const bool ok2 = team_wait(t_go, epoch, t_fail);
                            if (lane == 0) blk.pad0 = ok2 ? 0 : 1;
                        }
                        __syncthreads();
                        if (blk.pad0 != 0) {
                            team_failed = true;
                            break;
                        }
                        BlkWords u2;
#pragma unroll
                        for (int w = 0; w < 5; ++w) u2.w[w] = __hip_atomic_load(t_state + (size_t)(epoch % NSLOT) * 8 + w, RRT_RLX_AGENT);
                        if ((u2.b.pad1 & ST_FLAG_STOP) != 0) break;
                        i = unis32(u.b.i);
                        nprev = 0;
                    }
                }
            }
            __syncthreads();
            if (wave == 0 && g == 1) TSMARK(epoch, 7);
            if (wave == 0 && g == 8) TSMARK(epoch, 10);
            DBGT(3);
            continue;
        }

        // ---------------- team members g > 0: hand the records over, wait for the commit, take the new nodes ----------------
        if (G > 1 && !PIPE && g > 0) {
            if (wave == 0) {
                // this member's 16 records, LDS -> HBM: whole 128-byte lines per wave instruction, write-through (8-byte
                // stores of single lanes are partial-line fabric writes and delay everything queued behind them)
                const RRT_LDS u64 *src = (const RRT_LDS u64 *)&brec[0][g * BSM];
                for (int w = lane; w < BSM * BREC_WORDS; w += 64) __hip_atomic_store(t_rec + (size_t)g * BSM * BREC_WORDS + w, src[w], RRT_RLX_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing wave drains, then signals
                if (lane == 0) __hip_atomic_store(t_arrive + 32 * g, epoch, RRT_RLX_AGENT);
            }
            pre_i = -1;
            if (!informed && i0 + nb < n && !grid_nn) {  // the next block's samples are known (only an Informed block can be cut short)
                pre_i = i0 + nb;
                pre_j = j0;
                const int nbn = (n - pre_i) < SB ? (n - pre_i) : SB;
                pre_xv = lane < nbn ? samples[pre_i + lane] : 0u;
                uint32_t xsn[BSA];
#pragma unroll
                for (int k = 0; k < BSA; ++k) {
                    const int sk = g * BSM + k;
                    uint32_t X = (uint32_t)__builtin_amdgcn_readlane((int)pre_xv, sk);
                    if (sk >= nbn) X = (uint32_t)__builtin_amdgcn_readlane((int)pre_xv, 0);
                    xsn[k] = X << 4;
                    pre_best[k] = NONE;
                }
                scan_steps(0, pre_j / CHUNK, pre_j, xsn, pre_best);  // whole steps below the first one the commit can touch
            }
            if (wave == 0) {
                const bool ok = team_wait(t_go, epoch, t_fail);
                if (!grid_nn) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ONE acquire per workgroup: drops this CU's L1
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ... and holds the barrier until it has completed
                }
                if (lane == 0) blk.pad0 = ok ? 0 : 1;
            }
            __syncthreads();
            if (blk.pad0 != 0) {
                team_failed = true;
                break;
            }
            BlkWords u;
#pragma unroll
            for (int w = 0; w < 5; ++w) u.w[w] = __hip_atomic_load(t_state + w, RRT_RLX_AGENT);  // vector loads past the L1
            i = unis32(u.b.i);
            nsoln = unis32(u.b.nsoln);
            vbest_soln = unis32(u.b.vbest_soln);
            cmin_soln = unif64(u.b.cmin_soln);
            c_ell = unif64(u.b.c_ell);
            const int jn = unis32(u.b.j);
            if (t < jn - j0) {  // at most SB new nodes: append them to this CU's node cache and cell fill counts
                const uint32_t Xn = ld_u32<COH>(&at32(nodes_g, (uint32_t)(j0 + t)));
                if (j0 + t < lds_nodes) nodes_lds[j0 + t] = Xn;
                if (cells_on) __hip_atomic_fetch_add(&cellcnt[cell_of(Xn)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            j = jn;
            __syncthreads();
            continue;
        }

        // ---------------- C: commit (member 0; wave 0, one lane per sample) ----------------
        // References to nodes the commit itself inserts are kept as sample references until the store pass knows every node
        // index: 0x80000000 + 64 set + kk = sample kk of previous block `set` (pipelined teams: oldest first), set = NP: of this block.
        // They compare like the node indices they stand for (above every snapshot index, previous block first, sample order).
        BRec r;
        r.d2s = r.vs = r.los_s = r.flags = r.vbest = r.pstat = r.nnear = 0;
        r.amin = 0.0f;
        r.nnmask = r.rmask = r.dupmask = 0;
#pragma unroll
        for (int p2 = 0; p2 < NPMAX; ++p2) r.pnn[p2] = r.pr[p2] = r.pdup[p2] = 0;
        r.Vs = r.cbest = r.pc = 0.0;
        const u64 lbit = 1ull << lane;
        const u64 ltmask = lbit - 1ull;  // lanes below
        bool acc0 = false, goalhit = false, remote_ok = true;
        u64 harm = 0, acc_exact = 0, fin = 0, fin_acc = 0;  // fin: samples already re-resolved (by the parallel rounds)
        u64 popt = 0, inter = 0, known = 0, aknown = 0;  // parallel rounds: samples whose final result is known, the accepted ones
        bool rounds_on = false;
        bool pbad = false;  // an inserted sample of a previous block affects this sample
        // FASTC (a committer whose records carry the in-flight lines of sight, PL_MAX): psurv = the list entries of PREVIOUS blocks
        // that were inserted and would be tried as this sample's parent before the snapshot's choice (their costs are exact);
        // lists: this sample's list is usable (not void); such a sample is settled lane-parallel, without a round (fast_settle)
        constexpr bool FASTC = PIPE && BSM < 16;
        uint32_t psurv = 0, nent = 0;
        bool lists = false;
        // harm: the earlier samples within r_rewire that, once inserted at their cost, would be tried as this sample's parent
        // before the snapshot's choice: cost-through-it < cost through the snapshot parent (ties go to the lower index = the
        // snapshot, rrt.py:518-521).  A single-precision bound settles almost every pair; bit k is re-evaluated when sample
        // k's cost becomes exact.
        // nc + sqrt(d2) < cb, decided in single precision on either side whenever the margin allows (|error| of the f32 sum
        // < 1e-3 below 2^12 + relative 2e-7): the f64 square root only runs for close calls.
        auto cheaper_through = [&](double nc, uint32_t d2, double cb) -> bool {
            const float f = (float)nc + __builtin_amdgcn_sqrtf((float)d2), cf = (float)cb;
            if (f * (1.0f - 1.0e-6f) - 4.0e-3f >= cf * (1.0f + 1.0e-6f) + 4.0e-3f) return false;
            if (f * (1.0f + 1.0e-6f) + 4.0e-3f < cf * (1.0f - 1.0e-6f) - 4.0e-3f) return true;
            return nc + sqrt_u24(d2) < cb;
        };
        auto harmful = [&](int kk) -> bool { return cheaper_through(newcost[kk], dist2(xq_lds[kk], xv), r.cbest); };
        // Sample k on its own, re-resolved against snapshot + inserted nodes of this block (accepted: acc_k) and of the previous
        // ones (ap[], their count bases jp[]): any wave.  Its record in LDS is replaced by the final one; returns acceptance and cost.
        // Block references: 0x80000000 + 64 * set + kk with set 0 = the oldest previous block ... NP = this block.
        auto resolve_sample = [&](int k, u64 acc_k, const u64 (&ap)[NP], const int (&jp)[NP], bool check_full, bool &acc, double &cbest) {
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
            const unsigned long long rs0 = __builtin_amdgcn_s_memtime();
            bool rs_redo = false;
#endif
            BRecWords rku;  // the same record in every lane: kept in scalar registers
            rku.r = brec[bsel][k];
#pragma unroll
            for (int w = 0; w < BREC_WORDS; ++w) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rku.w[w]);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rku.w[w] >> 32));
                rku.w[w] = ((u64)hi << 32) | lo;
            }
            const BRec &rk = rku.r;
            const uint32_t Xk = (uint32_t)__builtin_amdgcn_readlane((int)xv, k);
            uint32_t vn = rk.vs, d2n = rk.d2s;
            double Vn = rk.Vs;
            bool nocoll = (rk.los_s >> 31) != 0;
            uint32_t cells = rk.los_s & 0x7fffffffu;
            double pc = rk.pc;
            uint32_t pi = (rk.pc < f64_inf()) ? rk.vbest : NONE, nnear = rk.nnear;
            uint32_t ntests = rk.pstat >> 20, tcells = rk.pstat & 0xfffffu;
            const uint32_t xo = (lane < SB) ? xq_lds[lane] : Xk;  // lane kk: sample kk
            const uint32_t dk = dist2(xo, Xk);
            // pipelined: lane kk also stands for sample kk of each previous block (inserted ones: ap[p], exact costs)
            uint32_t xop[NP], dkp[NP];
            u64 pnm[NP], pany = 0;
            bool pdup_hit = false;
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) {
                xop[p2] = PIPE ? xqp_lds[p2][lane] : Xk;
                dkp[p2] = dist2(xop[p2], Xk);
                pnm[p2] = PIPE ? (rk.pnn[p2] & ap[p2]) : 0ull;
                pany |= pnm[p2];
                pdup_hit = pdup_hit || (PIPE && (rk.pdup[p2] & ap[p2]) != 0);
            }
            const int snapj = PIPE ? jp[NP - 1] : j0;  // the node count the sample's owner resolved it against
            const bool dup = (rk.flags & 1u) != 0 || (rk.dupmask & acc_k) != 0 || pdup_hit;
            const u64 nm = rk.nnmask & acc_k;
            bool nn_inblock = false;
            if (nm | pany) {  // nearest is an inserted block node: smallest distance, lowest node index on ties (oldest block first)
                nn_inblock = true;
                uint32_t bestd = NONE, axy = Xk;
#pragma unroll
                for (int p2 = NP - 1; p2 >= 0; --p2) {
                    if (pnm[p2] != 0) {
                        uint32_t kdp = (pnm[p2] & lbit) ? dkp[p2] : NONE, kkp = (uint32_t)lane;
                        wave_min_key_idx(kdp, kkp);
                        if (kdp < bestd) {
                            bestd = kdp;
                            vn = 0x80000000u + (uint32_t)(NP - 1 - p2) * 64u + kkp;
                            Vn = prevcost[p2][kkp];
                            axy = (uint32_t)__builtin_amdgcn_readlane((int)xop[p2], (int)kkp);
                        }
                    }
                }
                if (nm != 0) {
                    uint32_t kd = (nm & lbit) ? dk : NONE, kk = (uint32_t)lane;
                    wave_min_key_idx(kd, kk);
                    if (kd < bestd) {
                        bestd = kd;
                        vn = 0x80000000u + (uint32_t)NP * 64u + kk;
                        Vn = newcost[kk];
                        axy = (uint32_t)__builtin_amdgcn_readlane((int)xo, (int)kk);
                    }
                }
                d2n = bestd;
                int cc = 0;
                nocoll = los_wave(og, H, axy, Xk, lane, cc);  // rrt.py:424
                cells = (uint32_t)cc;
            }
            acc = nocoll && !dup && !(check_full && j0 + __builtin_popcountll(acc_k) == n);  // rrt.py:425
            uint32_t vbest = vn;
            cbest = Vn + sqrt_u24(d2n);
            if (acc && star) {
                const double cnear = cbest;
                if (nn_inblock) {
                    // (only when the ball can hold an entry between the two bounds: the owner's lower bound of the cheapest entry at
                    //  or above its bound says so -- nearly never, and the search of the ball is what makes a round slow)
                    if (pi == NONE && (double)rk.amin < cnear && cnear > rk.Vs + sqrt_u24(rk.d2s)) {  // entries between the two bounds were never priced: redo the snapshot
                                                          // part (a parent found below the old bound stays the cheapest: same tests)
                        ntests = 0;
                        tcells = 0;
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
                        rs_redo = true;
#endif
                        snapshot_parent(Xk, snapj, true, cnear, pc, pi, nnear, ntests, tcells);
                    } else if (pi != NONE && !(pc < cnear)) {
                        pc = f64_inf();
                        pi = NONE;
                    }
                }
                // inserted block nodes within r_rewire, in (cost, index) order, while they beat the snapshot's best
                u64 rm = rk.rmask & acc_k;
                u64 rmp[NP], rmany = rm;
                nnear += (uint32_t)__builtin_popcountll(rm);
#pragma unroll
                for (int p2 = 0; p2 < NP; ++p2) {
                    rmp[p2] = PIPE ? (rk.pr[p2] & ap[p2]) : 0ull;
                    rmany |= rmp[p2];
                    nnear += (uint32_t)__builtin_popcountll(rmp[p2]);
                }
                while (rmany) {
                    // every lane's cheapest candidate among "its" sample of this block and of the previous ones, then ONE
                    // (cost, reference) minimum over the wave
                    double cn = f64_inf();
                    uint32_t ci = NONE;
                    if (rm & lbit) {
                        const double c = newcost[lane] + sqrt_u24(dk);
                        if (c < cnear) {
                            cn = c;
                            ci = 0x80000000u + (uint32_t)NP * 64u + (uint32_t)lane;
                        }
                    }
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) {
                        if (PIPE && (rmp[p2] & lbit)) {
                            const double c = prevcost[p2][lane] + sqrt_u24(dkp[p2]);
                            const uint32_t ref = 0x80000000u + (uint32_t)(NP - 1 - p2) * 64u + (uint32_t)lane;
                            if (c < cnear && key_lt(c, ref, cn, ci)) {
                                cn = c;
                                ci = ref;
                            }
                        }
                    }
                    wave_min_f64_idx(cn, ci);
                    if (ci == NONE || !key_lt(cn, ci, pc, pi)) break;
                    const uint32_t kk = ci & 63u;
                    const int set = (int)((ci - 0x80000000u) >> 6);  // 0 .. NP-1: previous blocks, oldest first; NP: this block
                    uint32_t axy = (uint32_t)__builtin_amdgcn_readlane((int)xo, (int)kk);
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2)
                        if (set == NP - 1 - p2) axy = (uint32_t)__builtin_amdgcn_readlane((int)xop[p2], (int)kk);
                    int cc = 0;
                    const bool ok = los_wave(og, H, axy, Xk, lane, cc);  // rrt.py:519
                    ntests += 1;
                    tcells += (uint32_t)cc;
                    if (ok) {
                        pc = cn;
                        pi = ci;
                        break;
                    }
                    if (set == NP) rm &= ~(1ull << kk);
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2)
                        if (set == NP - 1 - p2) rmp[p2] &= ~(1ull << kk);
                    rmany = rm;
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) rmany |= rmp[p2];
                }
                if (pi != NONE) {
                    vbest = pi;
                    cbest = pc;
                }
            }
            if (lane == 0) {  // the final record of this sample (pad = 1: nnear already counts the block's nodes)
                BRec &f = brec[bsel][k];
                f.vs = vn;
                f.los_s = cells;
                f.cbest = cbest;
                f.vbest = vbest;
                f.pstat = (ntests << 20) | (tcells & 0xfffffu);
                f.nnear = nnear;
                f.flags = rk.flags | 2u;
                newcost[k] = cbest;
            }
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
            if (t == 0) {
                dbg[rs_redo ? 10 : 12] += __builtin_amdgcn_s_memtime() - rs0;
                dbg[rs_redo ? 11 : 5] += 1;
            }
#endif
        };

        // FASTC: this lane's sample keeps its nearest vertex and its acceptance; inserted samples in flight (hm: of this block, by sample;
        // ps: of the previous blocks, by list position) are cheaper than its snapshot parent.  The walk of rrt.py:515-521 tries them in
        // (cost, index) order until one has a free line of sight -- the record's list holds every such test's answer, so the lane
        // settles it alone: the first free one becomes the parent; the blocked ones in front of it (or all of them, if none is free) are
        // the tests the walk made.  The sample's record in LDS is replaced by the final one, as a round would.
        auto fast_settle = [&](u64 hm, uint32_t ps, uint32_t ne, u64 rmask_own) {
            if constexpr (FASTC) {
                BRec &f = brec[bsel][lane];
                // candidates by list position: the previous blocks' (ps), and this block's (sample kk sits behind all previous-block
                // entries, at its rank among the earlier samples of this block within r_rewire)
                const uint32_t own0 = ne - (uint32_t)__builtin_popcountll(rmask_own);
                uint32_t cm = ps;
                while (hm) {
                    const int kk = __builtin_ctzll(hm);
                    hm &= hm - 1;
                    cm |= 1u << (own0 + (uint32_t)__builtin_popcountll(rmask_own & lowmask64(kk)));
                }
                auto entry_of = [&](uint32_t pe) -> uint32_t { return (uint32_t)(f.plist[pe >> 2] >> (16u * (pe & 3u))) & 0xffffu; };
                auto entry_cost = [&](uint32_t e16) -> double {
                    const int kk = (int)(e16 & 63u), set = (int)((e16 >> 6) & 3u);
                    if (set == NP) return newcost[kk] + sqrt_u24(dist2(xq_lds[kk], xv));
                    return prevcost[NP - 1 - set][kk] + sqrt_u24(dist2(xqp_lds[NP - 1 - set][kk], xv));
                };
                double bc = f64_inf();
                uint32_t bref = NONE, bcells = 0, nblocked = 0;
                for (uint32_t m = cm; m != 0; m &= m - 1) {
                    const uint32_t e16 = entry_of((uint32_t)__builtin_ctz(m));
                    if ((e16 >> 15) == 0u) {  // blocked
                        nblocked += 1;
                        continue;
                    }
                    const double c = entry_cost(e16);
                    const uint32_t ref = 0x80000000u + (e16 & 0xffu);  // (64 * set + sample: compares like the node index it stands for)
                    if (key_lt(c, ref, bc, bref)) {
                        bc = c;
                        bref = ref;
                        bcells = (e16 >> 8) & 127u;
                    }
                }
                uint32_t nt = bref != NONE ? 1u : 0u, tc = bcells;
                if (nblocked != 0) {  // blocked candidates: the ones in front of the winner (all, without one) were tested
                    for (uint32_t m = cm; m != 0; m &= m - 1) {
                        const uint32_t e16 = entry_of((uint32_t)__builtin_ctz(m));
                        if ((e16 >> 15) != 0u) continue;
                        if (bref == NONE || key_lt(entry_cost(e16), 0x80000000u + (e16 & 0xffu), bc, bref)) {
                            nt += 1;
                            tc += (e16 >> 8) & 127u;
                        }
                    }
                }
                const uint32_t ps0 = f.pstat;
                f.pstat = (((ps0 >> 20) + nt) << 20) | (((ps0 & 0xfffffu) + tc) & 0xfffffu);
                if (bref != NONE) {
                    f.cbest = bc;
                    f.vbest = bref;
                    newcost[lane] = bc;
                }
            }
        };

        // One wave: the SB records of block `ep` from the hand-off area into brec[half].  The block's records are one contiguous
        // piece (64 x BREC_WORDS x 8 bytes), so the wave takes it as whole 16-byte chunks, lane by lane: 80 cache lines instead of
        // 64 x BREC_WORDS separate 8-byte requests to lines another XCD wrote (that fetch was 8 k cycles, profiles/r04_experiments.md).
        auto fetch_records = [&](uint32_t ep, int half) {
            constexpr int NCH = SB * BREC_WORDS * 8 / 16, NIT = (NCH + 63) / 64;  // (the SB records of a block are the front of its 64-record slot)
            static_assert(NCH * 16 == SB * BREC_WORDS * 8, "whole chunks");
            const unsigned char *src = tb + TEAM_OFF_REC + (size_t)(ep % NSLOT) * 64 * BREC_WORDS * 8;
            u32x4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane;
                v[it] = ld_b128_agent(src + 16 * (c < NCH ? c : NCH - 1));
            }
            fence_b128s(v);
            RRT_LDS u32x4 *dst = (RRT_LDS u32x4 *)&brec[half][0];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = it * 64 + lane;
                if (c < NCH) dst[c] = v[it];
            }
        };

        // ---- part A (wave 0): the records, the optimistic picture, and for a pipelined committer the samples that can be
        //      re-resolved side by side ----
        if (wave == 0) {
            if (G > 1) {  // the other members' records: poll the arrival flags, then loads that bypass the L1
                DBGT(7);
                if (PIPE && prefetched) {  // wave 1 fetched them during the last commit
                    remote_ok = uni32(pre_state[bsel]) == 1u;
                } else {
                    remote_ok = team_wait_all(t_arrive, 1, PIPE ? G : G - 1, epoch, t_fail, lane);
                    if (PIPE) {
                        if (remote_ok) fetch_records(epoch, bsel);
                    } else if (remote_ok && lane >= BSM && lane < nb) {
                        const gu64 *src = t_rec + (size_t)(epoch % NSLOT) * 64 * BREC_WORDS + (size_t)lane * BREC_WORDS;
                        BRecWords u;
#pragma unroll
                        for (int w = 0; w < BREC_WORDS; ++w) u.w[w] = __hip_atomic_load(src + w, RRT_RLX_AGENT);
                        brec[bsel][lane] = u.r;
                    }
                }
            }
            DBGT(7);
            if (ROLE == ROLE_COMMIT) TSMARK(epoch, 11);
            if (!remote_ok) {
                if (lane == 0) blk.pad0 = 1;
            } else {
                if (lane < nb) r = brec[bsel][lane];  // lane s: sample s
                acc0 = lane < nb && (r.los_s >> 31) != 0 && (r.flags & 1u) == 0;  // accepted if nothing in the block interferes
                goalhit = informed && lane < nb && dist2(xv, xg) < goal_d2;
                if (lane < nb) newcost[lane] = r.cbest;  // optimistic (snapshot-resolved) cost; exact once the sample has committed
                if (PIPE && epoch == 1) {
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) jp0[p2] = j0;
                }
                if constexpr (FASTC) {
                    nent = (r.flags >> 8) & 0xffu;
                    lists = lane < nb && nent <= (uint32_t)PL_MAX;
                }
                if (lane < nb && acc0) {
                    u64 rm = r.rmask;
                    while (rm) {
                        const int kk = __builtin_ctzll(rm);
                        rm &= rm - 1;
                        if (harmful(kk)) harm |= 1ull << kk;
                    }
                }
                // pipelined: the nodes the previous block inserted are exact; the samples of this block were resolved without them
                if (PIPE && lane < nb) {
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) {
                        // inserted samples of a previous block that are nearer than this sample's nearest or sit on its cell
                        pbad = pbad || ((r.pnn[p2] | r.pdup[p2]) & A_prev[p2]) != 0;
                        if (acc0) {
                            u64 rm = r.pr[p2] & A_prev[p2];
                            // (FASTC: the list holds the samples within r_rewire oldest block first, sample order within a block: the entry of
                            //  sample kk of previous block p2 sits behind the entries of the older blocks, at its rank among the block's)
                            uint32_t pbase = 0;
#pragma unroll
                            for (int p3 = NP - 1; p3 > p2; --p3) pbase += (uint32_t)__builtin_popcountll(r.pr[p3]);
                            while (rm) {
                                const int kk = __builtin_ctzll(rm);
                                rm &= rm - 1;
                                if (cheaper_through(prevcost[p2][kk], dist2(xqp_lds[p2][kk], xv), r.cbest)) {
                                    if (FASTC && lists) psurv |= 1u << (pbase + (uint32_t)__builtin_popcountll(r.pr[p2] & lowmask64(kk)));
                                    else pbad = true;
                                }
                            }
                        }
                    }
                }
                if (PIPE) {
                    // Which affected samples can be re-resolved at once, each by a wave of its own?  Those whose earlier interacting
                    // samples are all known: at first the ones that keep their snapshot result whatever happens -- not affected
                    // themselves and (transitively) not interacting with an affected one ("tainted"); after a round also the
                    // samples it settled and the ones that turn out to keep their snapshot result given those.
                    popt = __ballot(acc0);
                    inter = (r.nnmask | r.dupmask | r.rmask) & ltmask;
                    const bool slow0 = lane < nb && (((r.nnmask | r.dupmask | harm) & popt & ltmask) != 0 || pbad || psurv != 0);
                    u64 taint = __ballot(slow0);
                    for (int it = 0; it < 6; ++it) {
                        const u64 t2 = taint | __ballot(lane < nb && (inter & taint) != 0);
                        if (t2 == taint) break;
                        taint = t2;
                        if (it == 5) taint = ~0ull;  // no fixed point yet: nothing is known beforehand
                    }
                    known = ~taint;
                    aknown = popt & ~taint;
                    rounds_on = j0 + __builtin_popcountll(popt) < n;  // (not in the run's last block: `j != n` needs exact counts)
                }
            }
        }
        DBGT(1);
        if (PIPE) {  // parallel rounds: wave w re-resolves sample par.list[w]; a round's results make further samples known
            u64 lastmask = 0;
            for (;;) {
                DBGT(2);
                if (wave == 0) {
                    uint32_t cnt = 0;
                    u64 newacc = 0;
                    if (lastmask != 0) {  // the samples the last round settled: final, whatever the ordered loop finds
                        newacc = __ballot(((lastmask >> lane) & 1ull) != 0 && par.accs[lane] != 0u);
                        fin |= lastmask;
                        fin_acc |= newacc;
                        known |= lastmask;
                        aknown |= newacc;
                    }
                    if (remote_ok && rounds_on && (~known & lowmask64(nb)) != 0) {  // (some sample's result is still open)
                        if (newacc != 0) {
                            // their costs are exact now: the later samples they are a candidate parent of look again
                            u64 rm = r.rmask & newacc;
                            if (lane < nb && acc0) {
                                while (rm) {
                                    const int kk = __builtin_ctzll(rm);
                                    rm &= rm - 1;
                                    harm = (harm & ~(1ull << kk)) | (harmful(kk) ? (1ull << kk) : 0ull);
                                }
                            }
                        }
                        u64 L0 = 0;
                        for (;;) {  // samples whose earlier interacting samples are all known: unaffected -> known; affected -> this round
                            const bool ready = lane < nb && (known & lbit) == 0 && (inter & ~known) == 0;
                            const bool hd = (harm & aknown & ltmask) != 0;  // an inserted sample of this block would be tried before the snapshot's choice
                            // sd: the sample has to be resolved again (a round); fd: only its parent is in question and the record holds
                            // every line of sight that question needs -- settled here, by its lane
                            const bool sd = ((r.nnmask | r.dupmask) & aknown & ltmask) != 0 || pbad || (goalhit && acc0) || (hd && !(FASTC && lists));
                            const bool fd = FASTC && !sd && (hd || psurv != 0);
                            const u64 cb = __ballot(ready && !sd && !fd);
                            const u64 fb = FASTC ? __ballot(ready && fd) : 0ull;
                            if (cb == 0 && fb == 0) {
                                L0 = __ballot(ready && sd && !goalhit);  // (a goal hit ends or cuts the block: ordered loop)
                                break;
                            }
                            known |= cb;
                            aknown |= cb & popt;
                            if constexpr (FASTC) {
                                if (fb != 0) {
                                    if (ready && fd) fast_settle(harm & aknown & ltmask, psurv, nent, r.rmask);
                                    fin |= fb;  // final, like the samples a round settled (all of them were accepted at the snapshot and stay so)
                                    fin_acc |= fb;
                                    known |= fb;
                                    aknown |= fb;
#ifdef RRT_STAMPS
                                    if (t == 0) dbg[13] += (unsigned long long)__builtin_popcountll(fb) << 32;  // (upper half: samples settled by their lanes)
#endif
                                    // their costs are exact now: the later samples they are a candidate parent of look again
                                    u64 rm = r.rmask & fb;
                                    if (lane < nb && acc0 && (known & lbit) == 0) {
                                        while (rm) {
                                            const int kk = __builtin_ctzll(rm);
                                            rm &= rm - 1;
                                            harm = (harm & ~(1ull << kk)) | (harmful(kk) ? (1ull << kk) : 0ull);
                                        }
                                    }
                                }
                            }
                        }
                        // the first CW of them (one per wave of this workgroup), in sample order: lane k knows its place in the list
                        const uint32_t place = (uint32_t)__builtin_popcountll(L0 & ltmask);
                        const bool listed = ((L0 >> lane) & 1ull) != 0 && place < (uint32_t)CW;
                        if (listed) par.list[place] = (uint32_t)lane;
                        lastmask = __ballot(listed);
                        cnt = (uint32_t)__builtin_popcountll(lastmask);
#ifdef RRT_STAMPS
                        if (t == 0) dbg[6] += cnt;
                        if (t == 0 && cnt != 0) dbg[13] += 1;
#endif
                        if (lane == 0 && cnt != 0) {
                            par.acc_opt = aknown;
#pragma unroll
                            for (int p2 = 0; p2 < NP; ++p2) {
                                par.aprev[p2] = A_prev[p2];
                                par.jp0[p2] = jp0[p2];
                            }
                        }
                    }
                    if (lane == 0) par.count = cnt;
                }
                DBGT(8);
                __syncthreads();
                const uint32_t cnt = uni32(par.count);
                if (cnt == 0) break;
                if ((uint32_t)wave < cnt) {
                    const int k = (int)uni32(par.list[wave]);
                    bool acc;
                    double cb;
                    u64 ap[NP];
                    int jp[NP];
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) {
                        ap[p2] = uni64(par.aprev[p2]);
                        jp[p2] = unis32(par.jp0[p2]);
                    }
                    resolve_sample(k, uni64(par.acc_opt) & lowmask64(k), ap, jp, false, acc, cb);
                    if (lane == 0) par.accs[k] = acc ? 1u : 0u;
                }
                __syncthreads();
            }
        }
        DBGT(2);
        // A pipelined committer's last two waves meanwhile fetch the next block: its records (workers that run ahead have handed them
        // over already) and its samples.  Only an Informed block can end early, so the next block is known.  (Behind the rounds, not
        // at the top of the block: there the last records are still ~9 k cycles away and the rounds' first barrier waits for this
        // wave -- measured, profiles/r04_experiments.md.)
        const bool pre_next = PIPE && (!informed || LAG >= 2) && i0 + nb < n;
        const bool pre_smp = pre_next && !informed;
        if (PIPE && wave == CW - 1 && pre_next) {
#ifdef RRT_STAMPS
            const unsigned long long pf0 = __builtin_amdgcn_s_memtime();
#endif
            TSMARK(epoch, 2);
#if defined(RRT_STAMPS) && !defined(RRT_STAMPS_LIGHT)
            bool ok = true;  // team_wait_all, counting who is late
            {
                const u64 t0w = wall_clock64();
                u64 lastmiss = 0;
                for (;;) {
                    const bool mine = lane < G ? __hip_atomic_load(t_arrive + 32 * (1 + lane), RRT_RLX_AGENT) >= epoch + 1 : true;
                    const u64 miss = __ballot(!mine);
                    if (miss == 0) break;
                    if (!mine) D->dbg2[lane] += 1;
                    lastmiss = miss;
                    if (__hip_atomic_load(t_fail, RRT_RLX_AGENT) != 0u || wall_clock64() - t0w > TEAM_TIMEOUT_TICKS) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (lastmiss != 0 && (lastmiss & (lastmiss - 1)) == 0 && ((lastmiss >> lane) & 1ull)) D->dbg2[64 + lane] += 1;
            }
#else
            const bool ok = team_wait_all(t_arrive, 1, G, epoch + 1, t_fail, lane);
#endif
#ifdef RRT_STAMPS
            const unsigned long long pf1 = __builtin_amdgcn_s_memtime();
#endif
            TSMARK(epoch, 3);
            if (ok) fetch_records(epoch + 1, bsel ^ 1);
            TSMARK(epoch, 4);
            if (lane == 0) pre_state[bsel ^ 1] = ok ? 1u : 2u;
#ifdef RRT_STAMPS
            if (lane == 0) dbg[14] += ((__builtin_amdgcn_s_memtime() - pf0) & 0xffffffffull) | ((pf1 - pf0) << 32);  // (upper half: the wait for the flags)
#endif
        }
        if (PIPE && wave == CW - 2 && pre_smp) {  // (the samples: a wave of their own, one memory round trip less in a row)
            const int in = i0 + nb;
            const int nbn = (n - in) < SB ? (n - in) : SB;
            if (lane < nbn) xq_next[bsel ^ 1][lane] = at32(samples, (uint32_t)(in + lane));
        }
        // ---- part B (wave 0): decide in order, store, publish ----
        if (wave == 0 && remote_ok) {
            int cur = 0;
            bool cut = false;
            // ---- C1: decide.  Runs of samples that keep their snapshot result are only marked; a sample that can be affected is
            //      re-resolved on its own and its record in LDS replaced by the final one.  Nothing is stored to HBM yet: the
            //      decisions only read the snapshot and the block's samples. ----
            const double c_ell0 = c_ell;
            const bool myacc0 = (fin & lbit) ? (fin_acc & lbit) != 0 : acc0;  // a settled sample's acceptance is known
            while (cur < nb && !cut) {
                const u64 pend = __ballot(myacc0 && lane >= cur);
                const u64 Aopt = acc_exact | (pend & ltmask);  // exact below cur, optimistic in [cur, lane)
                // A sample keeps its snapshot result unless an earlier inserted sample of the block is nearer than its nearest,
                // sits on its cell, or (accepted samples only) is a harmful candidate parent.
                const bool slow = lane >= cur && lane < nb && (fin & lbit) == 0 &&
                                  (((r.nnmask | r.dupmask | harm) & Aopt) != 0 || pbad || psurv != 0 || (goalhit && acc0));
                const unsigned long long bad = __ballot(slow);
                const int k0 = bad ? (int)__builtin_ctzll(bad) : nb;
                acc_exact |= pend & lowmask64(k0) & ~lowmask64(cur);
                cur = k0;
                if (cur >= nb) break;
                // ---- sample `cur` on its own ----
                {
                    const int k = cur;
                    const int jk = j0 + __builtin_popcountll(acc_exact);  // nodes when this sample is tried
                    const uint32_t Xk = (uint32_t)__builtin_amdgcn_readlane((int)xv, k);
                    bool acc;
                    double cbest;
                    resolve_sample(k, acc_exact, A_prev, jp0, true, acc, cbest);

                    if (acc) {
                        if (informed && dist2(Xk, xg) < goal_d2) {  // rrt.py:744-745
                            const bool first = nsoln == 0;
                            nsoln++;
                            if (cbest < cmin_soln) {  // np.argmin keeps the first minimum (rrt.py:632)
                                cmin_soln = cbest;
                                vbest_soln = jk;
                                c_ell = cmin_soln + sqrt_u24(dist2(xg, Xk));
                                cut = true;  // the ellipse changed: later samples of this block are stale (rrt.py:698-700)
                            }
                            if (first) cut = true;  // sampling switches from free space to the ellipse (rrt.py:695)
                        }
                        acc_exact |= 1ull << k;
                        const bool redo = lane > k && lane < nb && acc0 && ((r.rmask >> k) & 1ull) != 0;  // its cost is exact now
                        if (__ballot(redo) != 0 && redo) harm = (harm & ~(1ull << k)) | (harmful(k) ? (1ull << k) : 0ull);
                    }
                    cur = k + 1;
                }
            }
            DBGT(3);
            // ---- C2: commit samples [0, cur) in one lane-parallel pass ----
            {
                if (j0 + __builtin_popcountll(acc_exact) > n) acc_exact &= ~(1ull << (63 - __builtin_clzll(acc_exact)));  // rrt.py:425 `j != n`: only the run's last sample
                const bool inr = lane < cur;
                const BRec f0 = brec[bsel][lane];  // (lanes at and above `cur`: read, never used)
                BRec f = f0;
                // sample references -> node indices, now that every acceptance is known
                auto node_of = [&](uint32_t v) -> uint32_t {
                    if (v == NONE || (v & 0x80000000u) == 0) return v;
                    const int kk = (int)(v & 63u);
                    const int set = (int)((v - 0x80000000u) >> 6);  // 0 .. NP-1: previous blocks, oldest first; NP: this block
                    uint32_t node = (uint32_t)j0 + (uint32_t)__builtin_popcountll(acc_exact & lowmask64(kk));
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2)
                        if (set == NP - 1 - p2) node = (uint32_t)jp0[p2] + (uint32_t)__builtin_popcountll(A_prev[p2] & lowmask64(kk));
                    return node;
                };
                f.vs = node_of(f.vs);
                f.vbest = node_of(f.vbest);
                const bool myacc = inr && ((acc_exact >> lane) & 1ull) != 0;
                const int jmine = j0 + __builtin_popcountll(acc_exact & ltmask);  // j as this sample sees it
                if (inr) {
                    statred[lane * 5 + 0] += (unsigned long long)jmine;
                    statred[lane * 5 + 1] += (unsigned long long)(f.los_s & 0x7fffffffu);
                    if (logs) {
                        const size_t o = (size_t)q * bv.n_cap + i0 + lane;
                        bv.nearest_log[o] = (int32_t)f.vs;
                        bv.accept_log[o] = (uint8_t)myacc;
                        bv.cbest_log[o] = ell ? c_ell0 : __longlong_as_double(0x7ff8000000000000ll);
                        bv.j_log[o] = jmine;
                    }
                }
                if (myacc) {
                    if (star) {
                        uint32_t nprevnear = 0;
#pragma unroll
                        for (int p2 = 0; p2 < NP; ++p2) nprevnear += PIPE ? (uint32_t)__builtin_popcountll(f.pr[p2] & A_prev[p2]) : 0u;
                        statred[lane * 5 + 2] += (f.flags & 2u) ? f.nnear
                                                       : f.nnear + (uint32_t)__builtin_popcountll(f.rmask & acc_exact & ltmask) + nprevnear;
                        statred[lane * 5 + 4] += f.pstat >> 20;
                        statred[lane * 5 + 3] += f.pstat & 0xfffffu;
                    }
                    const uint32_t cellbit = (uint32_t)ux(xv) * (uint32_t)H + (uint32_t)uy(xv);
                    if (jmine < lds_nodes) nodes_lds[jmine] = xv;
                    const unsigned long long cb = (unsigned long long)__double_as_longlong(f.cbest);
                    if (G > 1) {
                        // a team: exactly the bytes the other members will read go out write-through (agent-scope stores), so
                        // that publishing the block needs no L2 write-back, only the wait for these stores
                        __hip_atomic_store((gu32 *)&at32(nodes_g, (uint32_t)jmine), xv, RRT_RLX_AGENT);
                        __hip_atomic_store((gu64 *)&at32(vcost, (uint32_t)jmine), (u64)cb, RRT_RLX_AGENT);
                        __hip_atomic_store((gu32 *)&at32(parent, (uint32_t)jmine), f.vbest, RRT_RLX_AGENT);
                    } else {
                        at32(nodes_g, (uint32_t)jmine) = xv;
                        at32(vcost, (uint32_t)jmine) = f.cbest;
                        at32(parent, (uint32_t)jmine) = (int32_t)f.vbest;
                    }
                    atomicOr(&at32(bitmap, cellbit >> 5), 1u << (cellbit & 31));  // rrt.py:426
                    if (cells_on) {  // the node's cell record: near set of RRT*, nearest-neighbour search
                        const int c = cell_of(xv);
                        const uint32_t slot = __hip_atomic_fetch_add(&cellcnt[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        u32x4 rc = {xv, (uint32_t)jmine, (uint32_t)cb, (uint32_t)(cb >> 32)};
                        if (G > 1) {  // (a reader takes a record only after the go flag: two 8-byte halves cannot be seen torn)
                            gu64 *dst = (gu64 *)&at32(cellrec, (uint32_t)c * (uint32_t)ccap + slot);
                            __hip_atomic_store(dst, ((u64)jmine << 32) | xv, RRT_RLX_AGENT);
                            __hip_atomic_store(dst + 1, (u64)cb, RRT_RLX_AGENT);
                        } else {
                            at32(cellrec, (uint32_t)c * (uint32_t)ccap + slot) = rc;
                        }
                    }
                }
                j = j0 + __builtin_popcountll(acc_exact);
                if (PIPE) {  // this block becomes the (newest) previous one
#pragma unroll
                    for (int p2 = NP - 1; p2 > 0; --p2) {
                        xqp_lds[p2][lane] = xqp_lds[p2 - 1][lane];
                        prevcost[p2][lane] = prevcost[p2 - 1][lane];
                        A_prev[p2] = A_prev[p2 - 1];
                        jp0[p2] = jp0[p2 - 1];
                    }
                    xqp_lds[0][lane] = xv;
                    prevcost[0][lane] = f.cbest;
                    A_prev[0] = acc_exact;
                    jp0[0] = j0;
                }
            }
            i = i0 + cur;
            if (lane == 0) {
                BlkState b;
                b.i = i;
                b.j = j;
                b.nsoln = nsoln;
                b.vbest_soln = vbest_soln;
                b.pad0 = 0;
                b.pad1 = (PIPE && cut) ? ST_FLAG_RESTART : 0;
                b.cmin_soln = cmin_soln;
                b.c_ell = c_ell;
                blk = b;
            }
            // state first (write-through), then everything the commit stored, then the flag
            DBGT(9);
            if (G > 1) publish_state(epoch, (PIPE && cut) ? ST_FLAG_RESTART : 0);
            if (ROLE == ROLE_COMMIT) TSMARK(epoch, 1);
            DBGT(4);
        }
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
