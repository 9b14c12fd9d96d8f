/*
 * rrt_hip.h -- C ABI of the MI355X (gfx950) RRT / RRT* tree-expansion engine.
 *
 * The reference (rland93/rrtplanner) has no FFI: its boundary is the body of
 * RRTStandard.plan / RRTStar.plan / RRTStarInformed.plan (rrtplanner/rrt.py:386-447,
 * :466-556, :653-758).  This header is the boundary a maintainer binds instead of those
 * loops (INTEGRATION.md shows the ctypes stub).  Plain C: pointers and sizes only, no
 * torch / numpy types.  All functions return 0 (RRT_OK) or a positive "host action
 * needed" status or a negative error; no exception crosses the boundary.
 * rrt_last_error_string() describes the last failure.
 *
 * Threading: a ctx (and the batches made from it) is used by one host thread at a
 * time; distinct ctxs may be used by distinct host threads at the same time (each owns
 * a HIP stream).  They share the device: a launch sizes its teams of compute units by
 * what the launches in flight of the same process have left free (rrt_batch_team_info),
 * so concurrent batches slow each other down but never wait for CUs they cannot get.
 * Kernels of OTHER processes on the same GPU are not seen by that registry; there the
 * bounded hand-off wait (0.5 s, then one CU per query) remains the safety net.
 */
#ifndef RRT_HIP_H
#define RRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_OK 0
#define RRT_NEED_UNITBALL 1        /* Informed: query reached ellipse sampling (rrt.py:697) without unit-ball data */
#define RRT_E_ARG (-1)
#define RRT_E_GOAL_UNREACHABLE (-2) /* rrt.py:317-318 would index og[INT64_MIN,..]: no line of sight, j < n */
#define RRT_E_HIP (-3)
#define RRT_E_NOGRID (-4)
#define RRT_E_UNSUPPORTED (-5)     /* rrt_batch_create / rrt_plan: grid larger than 2048 x 2048 or n > 262143 (the expansion kernels' packed keys; such a
                                      planner runs host-driven over rrt_tree_query instead, slower, same results); rrt_set_grid: more than 32767 cells per axis */
#define RRT_E_COMM (-6)            /* multi-GPU gather: librccl missing, no communicator, RCCL error, unequal slabs */

#define RRT_ALG_STANDARD 0 /* RRTStandard.plan     rrt.py:386 */
#define RRT_ALG_STAR 1     /* RRTStar.plan         rrt.py:466 */
#define RRT_ALG_INFORMED 2 /* RRTStarInformed.plan rrt.py:653 */
/* Dubins-vehicle planners (BASELINE.json configs[4]).  The reference only advertises them (README.md:12,18-19) and ships no
 * module: NO REFERENCE PARITY, semantics defined in include/rrt_dubins.h (state = cell + discrete heading, edge = shortest
 * Dubins word of turning radius rho, cost = its arc length, collision = its sampled sweep; nearest / within / accept /
 * choose-parent exactly as rrt.py:418-437 / :498-548).  Need a batch created with RRT_FLAG_DUBINS. */
#define RRT_ALG_DUBINS 3      /* parent = nearest */
#define RRT_ALG_DUBINS_STAR 4 /* choose parent within r_rewire */

#define RRT_FLAG_LOGS 1u   /* keep per-iteration logs (nearest, accept, ellipse cost, j) on the device */
#define RRT_FLAG_SERIAL 2u /* use the one-sample-per-iteration kernel instead of the 16-sample block kernel */
#define RRT_FLAG_NOTEAM 4u /* block kernel on ONE workgroup (CU) per query.  Default: a team of up to 64 CUs per query, the
                              largest for which all teams of the batch are resident on the device together */
#define RRT_FLAG_TEAM_FAULT 8u /* testing: one member of every team leaves at once; the hand-offs of the others time out
                                  and the batch must finish with one CU per query */
#define RRT_FLAG_NOPIPE 16u /* teams of 8 and more CUs: do not pipeline super-blocks (the commit of block s under the resolution
                              of block s + 1; RRTStandard / RRTStar only) */
#define RRT_FLAG_REWIRE 32u /* opt-in TRUE RRT* rewire with cost propagation -- NOT the reference's behaviour: its
                              rewire step never fires (rrt.py:532-536 prices the rewire with vcosts[vn] + d, never below vcosts[vn]).
                              Semantics: oracle/rrt_oracle.c; runs on the one-sample-per-iteration kernel.  Default off. */
#define RRT_FLAG_DUBINS 64u /* the batch runs Dubins queries (RRT_ALG_DUBINS / RRT_ALG_DUBINS_STAR) only, one CU per query: 16 samples per
                              round on per-node headings and cell records (rrt_dubins_block.h); with RRT_FLAG_SERIAL the
                              one-sample-per-iteration kernel, kept as a cross-check.  rrt_plan sets it by itself for such a query. */
#define RRT_FLAG_NOPIPE1 32768u /* one CU per query, RRTStandard / RRTStar: the 16-samples-per-pass block kernel instead of the barrier-free
                                  pipeline (rrt_pipe.h); kept as a cross-check, the results are the same */
#define RRT_FLAG_TEAM_MAX(g) ((uint32_t)(g) << 8) /* cap the team size at g CUs per query (g = 2, 4, ... 64; 0 = no cap) */

typedef struct rrt_ctx rrt_ctx;
typedef struct rrt_batch rrt_batch;

/* One planning query == the arguments of plan() plus the planner's state that the loop
 * reads (rrt.py:51-85, :454-464, :563-577). */
typedef struct rrt_query {
    int32_t alg;            /* RRT_ALG_* */
    int32_t n;              /* self.n: attempted samples and node capacity (rrt.py:62) */
    int32_t xs[2];          /* xstart */
    int32_t xg[2];          /* xgoal */
    int64_t r2_rewire;      /* smallest integer R with (d2 < r_rewire*r_rewire) <=> (d2 < R)   (rrt.py:180) */
    int64_t goal_d2;        /* smallest integer G with (r2norm(d) < r_goal) <=> (d2 < G)      (rrt.py:744) */
    const int32_t *samples; /* host (n,2): free[rand_gen.choice(F)] for iteration i (rrt.py:240) */
    double C[4];            /* Informed: rotation_to_world_frame(xstart,xgoal), row-major (rrt.py:601-613) */
    /* Dubins planners only (ignored otherwise) */
    const int32_t *headings; /* host (n): heading index of sample i, 0 <= h < nh */
    double rho;              /* turning radius in cells, > 0 */
    int32_t nh;              /* number of discrete headings, 1 .. 256 */
    int32_t hs, hg;          /* heading index of the start / goal pose */
    int32_t pad_;
    /* optional: the same samples already packed, x | y << 16 per sample (n words).  When non-NULL it is used instead of
     * `samples` (which may then be NULL): a host that keeps its free cells packed saves the (n,2) int64 gather of rrt.py:240 */
    const uint32_t *samples_packed;
} rrt_query;

/* Result of one query: the arrays plan() hands to build_graph (rrt.py:334-369).
 * pts/vcost/parent are caller-allocated with n+1 rows; rows [0, rows) are written.
 * Optional logs (NULL to skip, need RRT_FLAG_LOGS) have n rows. */
typedef struct rrt_result {
    int32_t *pts;         /* (n+1,2) */
    double *vcost;        /* (n+1)   */
    int32_t *parent;      /* (n+1)   -1 = root / none */
    int32_t *nearest_log; /* (n) vnearest of iteration i */
    uint8_t *accept_log;  /* (n) 1 if iteration i inserted a node */
    double *cbest_log;    /* (n) ellipse cost c of iteration i (rrt.py:698-699), NaN when free-sampled */
    int32_t *j_log;       /* (n) j at the top of iteration i (key of self.ellipses, rrt.py:701) */
    int32_t status;       /* RRT_OK / RRT_NEED_UNITBALL / RRT_E_GOAL_UNREACHABLE */
    int32_t j;            /* tree nodes before go2goal */
    int32_t vgoal;        /* rrt.py:319 / :331 */
    int32_t found;        /* go2goal connected the goal */
    int32_t i_switch;     /* first iteration sampled from the ellipse (n if none) */
    int32_t rows;         /* len(points) after go2goal: j+1 live rows are valid; n+1 if found else n in the reference */
    int64_t sum_j;        /* statistics for the algorithmic-byte model */
    int64_t sum_cells_nn;
    int64_t sum_near;
    int64_t sum_cells_cand;
    int64_t n_los_cand;
    int64_t n_rewired;    /* RRT_FLAG_REWIRE: nodes re-parented / descendant costs recomputed (0 otherwise: rrt.py:536 is never true) */
    int64_t n_propagated;
    int32_t *head;        /* Dubins planners: (n+1) heading index of every node; NULL to skip */
    int64_t n_words;      /* Dubins planners: shortest-word evaluations the device made (0 otherwise) */
} rrt_result;

/* ---- context / grid -------------------------------------------------------------- */
int rrt_ctx_create(int32_t device_id, rrt_ctx **out);
int rrt_ctx_destroy(rrt_ctx *ctx); /* destroy the batches created on a context before the context itself */
const char *rrt_last_error_string(rrt_ctx *ctx); /* ctx may be NULL */
/* RRT.__init__ / set_og (rrt.py:64-65, :261-272): og_nonzero is (W,H) C-order, 1 = obstacle.  Up to 32767 cells per axis; the
 * expansion kernels (rrt_batch_*, rrt_plan*) take grids up to 2048 x 2048, larger ones serve rrt_tree_query and
 * rrt_prim_collisionfree only. */
int rrt_set_grid(rrt_ctx *ctx, const uint8_t *og_nonzero, int32_t W, int32_t H);

/* Device-resident noise grids (counterpart of oggen.perlin_occupancygrid, oggen.py:7-45: fractal gradient noise, min-max
 * normalised over all frames, obstacle where value < thresh).  `octaves` lattices of unit gradients (host-drawn, seeded):
 * octave o has dims[3*o..3*o+2] = (nz, nx, ny) lattice points, cell edge cells[o] pixels and amplitude amps[o]; grads holds
 * the lattices back to back, 3 doubles per point.  All `frames` grids stay in HBM; frame 0 becomes the active grid.
 * og_out (optional, host, frames*W*H bytes) receives a copy (the host needs it for free = argwhere(og == 0)). */
int rrt_noise_grids(rrt_ctx *ctx, int32_t W, int32_t H, int32_t frames, float thresh, int32_t octaves, const int32_t *dims,
                    const double *cells, const double *amps, const double *grads, uint8_t *og_out);
/* make frame k of the resident noise grids the active grid (no upload; anim.py:92-93 style replanning) */
int rrt_select_frame(rrt_ctx *ctx, int32_t frame);
/* counts the calls that rewrote the context's grid storage (rrt_set_grid, rrt_noise_grids).  A caller that keeps frames
 * resident records it after rrt_noise_grids and compares before rrt_select_frame: a different value means the frames are gone. */
int rrt_grid_generation(rrt_ctx *ctx, uint64_t *generation);
/* wait for everything queued on the context's stream and on its device */
int rrt_ctx_sync(rrt_ctx *ctx);

/* ---- resident batches: Q independent queries on the ctx's grid ------------------------- */
int rrt_batch_create(rrt_ctx *ctx, int32_t Q, int32_t n_cap, uint32_t flags, rrt_batch **out);
int rrt_batch_destroy(rrt_batch *b);
int rrt_batch_set_query(rrt_batch *b, int32_t q, const rrt_query *query); /* uploads samples, arms query q */
/* Informed: unit-ball points (rrt.py:582-586) for iterations ub_offset .. ub_offset+count-1 */
int rrt_batch_set_unitball(rrt_batch *b, int32_t q, const double *unitball, int32_t count, int32_t ub_offset);
int rrt_batch_rearm(rrt_batch *b);  /* reset every query's tree, keep the uploaded inputs */
int rrt_batch_launch(rrt_batch *b); /* asynchronous on the ctx stream; runs / resumes every unfinished query */
int rrt_batch_sync(rrt_batch *b);
/* CUs working on each query (the team size chosen at rrt_batch_create) and how often a team hand-off timed out: such a launch
 * is continued once with one CU per query, the next launch uses the team again */
int rrt_batch_team(rrt_batch *b, int32_t *cus_per_query, int32_t *fallbacks);
/* out = {workers per query the batch was created with (the shape on an otherwise idle device), workers per query of the last
 * launch, launches continued with one CU per query after a hand-off timed out, launches that ran a SMALLER team because other
 * launches of this process held compute units of the device}.  The members of a team wait for each other, so all of them must be
 * resident at once: every launch claims its compute units in a per-device registry of the library and, when they are not all
 * free, takes the largest team that fits next to the launches in flight (rrt_batch_sync returns the claim). */
int rrt_batch_team_info(rrt_batch *b, int32_t out[4]);
/* 1 if the last launch ran the pipelined team kernel (one more CU per query, which only commits) */
int rrt_batch_pipelined(rrt_batch *b, int32_t *pipelined);
/* name of the expansion kernel the last launch ran (the one before it, for a batch not launched yet: the one it would run),
 * as rocprofv3 prints it: "rrt_expand_block_kernel<64, 1, true, false>"; buf receives at most len - 1 characters */
int rrt_batch_kernel_name(rrt_batch *b, char *buf, int32_t len);
int rrt_batch_elapsed_ms(rrt_batch *b, float *ms); /* HIP events around the last launch's kernels (incl. a launch that timed out) */
int rrt_batch_get_result(rrt_batch *b, int32_t q, rrt_result *out);
/* diagnostic builds (-DRRT_STAMPS): shader cycles wave 0 of query q spent in scan / barrier / nearest+line of sight /
 * choose parent / insert / go2goal; zeros in the product build */
int rrt_batch_debug_cycles(rrt_batch *b, int32_t q, uint64_t out[38]); /* [0..5] phases, [6..37] per-wave owner-phase cycles */
/* device-resident result slab of the batch: [vcost f64 | nodes u32 (x | y<<16) | parent i32], each [Q][stride], then
 * {status, j, vgoal, found} i32 per query (filled by rrt_gather); stride = (bytes - 16 Q) / (16 Q) */
int rrt_batch_result_block(rrt_batch *b, void **dev_ptr, int64_t *bytes);

/* ---- multi-GPU: one process per GPU, query q on rank q mod world (no data-path collective), one all-gather of the result
 * slabs over RCCL / xGMI at the end of a batch.  The reference has no counterpart (single process).
 * librccl is opened on the first of these calls; the single-GPU path never loads it. ---- */
#define RRT_COMM_ID_BYTES 128
/* Optional, before the first rrt_comm_* call of the process: the collective library to open instead of librccl.so.1 -- any library
 * that exports ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclAllReduce / ncclGetErrorString with RCCL's
 * signatures.  tests/fake_rccl is one for ranks that share ONE GPU (RCCL itself refuses two ranks on a device), so that the
 * world > 1 path can run on a one-GPU box.  RRT_E_COMM once a library has been opened; path NULL = back to the default. */
int rrt_comm_use_library(const char *path);
/* rank 0: a fresh communicator id (ncclGetUniqueId); the host ships the bytes to the other ranks over any side channel */
int rrt_comm_unique_id(uint8_t id[RRT_COMM_ID_BYTES]);
/* collective over all `world` ranks (ncclCommInitRank on the context's device) */
int rrt_comm_init(rrt_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[RRT_COMM_ID_BYTES]);
int rrt_comm_destroy(rrt_ctx *ctx);
/* vals[0..count) := reduction over all ranks (op 0 sum, 1 max, 2 min), count <= 64; synchronous: also the barrier */
int rrt_comm_allreduce_f64(rrt_ctx *ctx, double *vals, int32_t count, int32_t op);
/* ncclAllGather of the batch's result slab on the context's stream: rank r's slab lands at gathered_dev + r * bytes_per_rank on
 * every rank.  Every rank must bring a batch of the same Q and capacity: checked by a 16-byte all-reduce in front of EVERY gather
 * (RRT_E_COMM on all ranks when the sizes differ).  That check is host-synchronous (it is rrt_comm_allreduce_f64: also a barrier
 * of the ranks), so the call returns after all ranks have entered it; only the all-gather itself is left asynchronous on the
 * stream (rrt_ctx_sync / the next synchronous call waits for it). */
int rrt_gather(rrt_batch *b, void **gathered_dev, int64_t *bytes_per_rank);
/* after rrt_gather ON THIS BATCH (the slabs of another batch, even one of equal size, are refused): query q of rank `rank` from
 * the gathered slabs into caller-allocated host arrays.  On entry out->rows is the CAPACITY of pts / vcost / parent in rows (the
 * remote query's row count is only known from its slab: RRT_E_ARG when it exceeds the capacity, nothing is written then); on
 * return rows = rows written; status, j, vgoal, found are filled; logs and statistics are not gathered */
int rrt_gather_fetch(rrt_batch *b, int32_t rank, int32_t q, rrt_result *out);

/* ---- one-shot wrappers (what plan() binds) --------------------------------------------- */
int rrt_plan(rrt_ctx *ctx, const rrt_query *query, uint32_t flags, rrt_result *out);
int rrt_plan_resume(rrt_ctx *ctx, const double *unitball, int32_t count, rrt_result *out);
int rrt_plan_batch(rrt_ctx *ctx, int32_t Q, const rrt_query *queries, rrt_result *out);

/* ---- host-driven planners: a caller-supplied cost function (rrt.py:55, :70-80 accepts any Python callable) cannot run on the
 * device, so for such a planner the loop of rrt.py:498-548 / :690-748 stays on the host and asks the device, once per
 * iteration, for what it needs of the tree and the grid: near()[0], within() and the lines of sight of rrt.py:506, :519, :537.
 * The vertex list lives on the device (one 4-byte append per accepted sample). -------------------------------------------- */
typedef struct rrt_tree rrt_tree;
int rrt_tree_create(rrt_ctx *ctx, int32_t capacity, rrt_tree **out);
int rrt_tree_destroy(rrt_tree *t);
int rrt_tree_reset(rrt_tree *t); /* no vertices */
/* points[j] = (x, y) for the next free row j (rrt.py:411, :525); *index receives j.  Asynchronous. */
int rrt_tree_append(rrt_tree *t, int32_t x, int32_t y, int32_t *index);
/* One iteration's questions about sample (x, y), against the context's current grid:
 *   *nearest       near(points, x)[0] over the live rows, lowest index among equal distance          (rrt.py:503)
 *   *within_count  number of live rows with d2 < r2; within_idx[0 .. min(count, cap)) their indices, ascending (rrt.py:513)
 *   los_free[0]    collisionfree(og, points[nearest], x);  los_free[1 + k] = collisionfree(og, points[within_idx[k]], x)
 * los_free holds cap + 1 bytes.  RRT_E_ARG when the tree is empty or (x, y) lies outside the grid. */
int rrt_tree_query(rrt_tree *t, int32_t x, int32_t y, int64_t r2, int32_t *nearest, int32_t *within_count, int32_t *within_idx, uint8_t *los_free,
                   int32_t cap);

/* ---- primitives of the path (parity tests, micro-benchmarks) --------------------------- */
/* RRT.collisionfree (rrt.py:183-229) for m segments ab[k] = {ax,ay,bx,by}; cells = grid cells the
 * reference's walk reads before it returns. */
int rrt_prim_collisionfree(rrt_ctx *ctx, const int32_t *ab, int32_t m, uint8_t *out_free, int32_t *out_cells);
/* near()[0] (rrt.py:150-155,:422) and |within()| (rrt.py:176-181) of m query points against j nodes */
int rrt_prim_nearest_within(rrt_ctx *ctx, const int32_t *pts, int32_t j, const int32_t *xq, int32_t m,
                            int64_t r2, int32_t *out_nearest, int32_t *out_within_count,
                            int64_t *out_within_idxsum);
/* sqrt of the integers lo .. lo+count-1 as the kernels compute it (r2norm, rrt.py:24) */
int rrt_prim_sqrt_u32(rrt_ctx *ctx, uint32_t lo, uint32_t count, double *out);
/* the block kernel's short sqrt, valid for radicands below 2^24 */
int rrt_prim_sqrt_u24(rrt_ctx *ctx, uint32_t lo, uint32_t count, double *out);
/* sqrt of arbitrary doubles as the kernels compute it (ellipse minor axis, rrt.py:622) */
int rrt_prim_sqrt_f64(rrt_ctx *ctx, const double *in, uint32_t count, double *out);

#ifdef __cplusplus
}
#endif
#endif /* RRT_HIP_H */
