// rrt_engine.hip -- C ABI (include/rrt_hip.h) over the gfx950 kernels of rrt_kernels.h.
// Host side: HIP memory, one stream per context, events for kernel timing.  No torch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is opened on demand (rrt_comm_init), the single-GPU path never loads it
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

// The expansion kernels are only LAUNCHED from this translation unit; kernels_tu.hip holds their definitions, dealt to several
// translation units that compile side by side (the single unit of round 3 took two and a half minutes).
#define RRT_BLOCK_DECL_ONLY
#define RRT_SERIAL_DECL_ONLY
#include "rrt_hip.h"
#include "rrt_kernels.h"
#include "rrt_block.h"
#include "rrt_kernel_decls.h"
#include "rrt_prims.h"

using namespace rrtdev;

static thread_local std::string g_last_error;

constexpr int RRT_GRID_FAST = 2048;  // the expansion kernels: coordinates below 2^11, squared distances below 2^24 (packed scan keys, sqrt_u24)
constexpr int RRT_GRID_MAX = 32767;  // the host-driven path (rrt_tree_query, rrt_prim_collisionfree): 16-bit packed coordinates whose differences fit int16

// Host copy of a batch's query descriptors in page-locked memory: the per-step copies to and from the device (rrt_batch_rearm,
// rrt_batch_sync) are then plain DMA transfers in stream order, with no staging copy and no hidden synchronisation.
struct PinnedDescs {
    QDesc *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), count * sizeof(QDesc), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        n = count;
        for (size_t k = 0; k < n; ++k) p[k] = QDesc{};
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    QDesc *data() { return p; }
    QDesc *begin() { return p; }
    QDesc *end() { return p + n; }
    QDesc &operator[](size_t k) { return p[k]; }
};

struct rrt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t *og = nullptr;      // active grid, device (W,H): og_buf + frame * W * H
    uint8_t *og_buf = nullptr;  // allocation holding 1 uploaded grid or `nframes` generated grids
    size_t og_buf_bytes = 0;
    int32_t nframes = 1;
    int32_t W = 0, H = 0;
    std::string err;
    rrt_batch *single = nullptr;  // batch behind rrt_plan / rrt_plan_resume
    uint32_t single_flags = 0;
    int max_lds = 0;
    int num_cu = 0;
    uint64_t grid_gen = 0;  // bumped by every call that rewrites or reallocates og_buf (rrt_set_grid, rrt_noise_grids)
    // multi-GPU result gather (RCCL over xGMI); all null / 1 until rrt_comm_init
    ncclComm_t comm = nullptr;
    int32_t comm_rank = 0, comm_world = 1;
    unsigned char *gather_buf = nullptr;  // [world][slab bytes of the batch gathered last]
    size_t gather_bytes = 0;
    const rrt_batch *gather_owner = nullptr;  // the batch whose slabs gather_buf holds (rrt_gather_fetch serves no other)
    double *d_red = nullptr;  // small device scratch of rrt_comm_allreduce_f64
};

struct rrt_batch {
    rrt_ctx *ctx = nullptr;
    int32_t Q = 0, n_cap = 0, node_stride = 0, bitmap_words = 0, lds_chunks = 1, spill_stride = 0;
    int32_t gridW = 0, gridH = 0;
    uint32_t flags = 0;
    bool use_block = false;     // block-parallel kernel (rrt_block.h) instead of the one-sample-per-iteration kernel
    bool dub_block = false;     // Dubins planners on the 16-samples-per-round kernel (rrt_dubins_block.h); RRT_FLAG_SERIAL keeps the one-sample kernel
    int32_t blk_lds_chunks = 1; // node chunks cached in LDS by the block kernel: teams of 8 and more workers ...
    int32_t blk_lds_chunks16 = 1; // ... and the kernels that also keep their parked-entry lists there
    int32_t team = 1;           // workgroups (CUs) per query of the block kernel that scan and resolve (rrt_block.h, teams)
    bool pipe_team = false;     // the team is pipelined: one more workgroup per query, which only commits
    bool pipe = false;          // the last launch ran the pipelined team kernel
    int32_t last_team = 0;      // workers per query of the last launch (1 after a hand-off timed out)
    bool last_inf = false;      // the last launch ran the Informed instantiation
    bool last_pipe1 = false;    // the last launch ran the barrier-free one-CU kernel (rrt_pipe.h)
    bool last_wide = false;     // the last launch ran a team variant with more than 16 samples per member
    int32_t team_fallbacks = 0; // launches repeated with one CU per query after a team hand-off timed out
    int32_t team_qpad = 0;      // Q rounded up to a multiple of 8: block = member * team_qpad + query
    int32_t team_want = TEAM_MAX;  // the caller's cap on the team size
    int32_t claimed_cus = 0;    // compute units this batch's launch in flight holds in the device's registry (0: nothing in flight)
    int32_t shrunk = 0;         // launches that ran a smaller team than the batch was created with because other launches held CUs
    unsigned char *d_team = nullptr;  // [Q][TEAM_BYTES] sync words, state, exchanged records; zeroed before every launch
    QDesc *d_desc = nullptr;
    PinnedDescs h_desc;  // page-locked
    size_t serial_lds_static = 0;  // static LDS of the one-sample-per-iteration kernel + 1 (0 = not asked yet)
    uint32_t *d_samples = nullptr, *d_nodes = nullptr, *d_bitmap = nullptr;
    double *d_vcost = nullptr, *d_unitball = nullptr, *d_cbest_log = nullptr;
    int32_t *d_parent = nullptr, *d_nearest_log = nullptr, *d_j_log = nullptr;
    uint8_t *d_accept_log = nullptr;
    uint2 *d_spill = nullptr;
    int32_t *d_kids = nullptr;      // RRT_FLAG_REWIRE: [3][Q][node_stride] first child / next sibling / previous sibling
    uint32_t *d_frontier = nullptr; //                  [Q][2 * node_stride]
    int32_t *d_vsoln = nullptr;     //                  [Q][node_stride]
    uint8_t *d_heading = nullptr;   // RRT_FLAG_DUBINS: [Q][node_stride] node headings
    uint8_t *d_shead = nullptr;     //                  [Q][n_cap] sample headings
    double *d_dubpath = nullptr;    //                  [Q][NWAVE * WCAP][5] the words of the current iteration's near-set entries
    std::vector<uint8_t> stage8;
    uint4 *d_cellrec = nullptr;    // block kernel: near-set records, [Q][rec_stride]
    uint32_t *d_cellcnt = nullptr; // [Q][MAX_CELLS]
    int64_t rec_stride = 0;
    unsigned char *d_slab = nullptr;  // result slab: [vcost f64 | nodes u32 | parent i32], each [Q][node_stride]
    size_t slab_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    float ms_before = 0.f;      // kernel time of the launch a fallback relaunch replaced (rrt_batch_elapsed_ms adds it)
    bool one_cu_once = false;   // the next launch runs one CU per query whatever b->team says (continuation after a timeout)
    std::vector<uint32_t> stage;  // host staging for packed samples
};

static const void *block_kernel_of(int team, bool pipe, bool inf);
static size_t block_kernel_static_lds(int team);

// The limit is a property of the kernel on a device, shared by every batch that launches it: it is only ever raised, to the
// largest request seen, and hipFuncSetAttribute is called when a launch needs more than the kernel already has -- once per
// kernel and size in practice, instead of once per launch.
static hipError_t raise_dynamic_lds(int device, const void *kern, int bytes) {
    static std::mutex mu;
    static std::vector<std::tuple<int, const void *, int>> have;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &t : have)
        if (std::get<0>(t) == device && std::get<1>(t) == kern) {
            if (std::get<2>(t) >= bytes) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess) std::get<2>(t) = bytes;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have.emplace_back(device, kern, bytes);
    return e;
}

static int fail(rrt_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, RRT_E_HIP, "%s: %s", #call, hipGetErrorString(e_));     \
    } while (0)

// Wait for the context's stream by polling (no interrupt wake-up of a sleeping host thread: on a host that parks the waiting
// thread the default wait costs up to a millisecond per step, against a 9 ms launch).  A wait that lasts longer than
// `spin_ms` falls through to the blocking wait, where the wake-up no longer matters and a spinning core would.
static hipError_t wait_stream_spin(hipStream_t stream, double spin_ms = 100.0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0;; ++it) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e;
        if ((it & 1023u) == 1023u && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > spin_ms)
            return hipStreamSynchronize(stream);
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
}

// device temporaries of one call: freed on every return path
struct DevTmp {
    std::vector<void *> ptrs;
    ~DevTmp() {
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
    }
    template <typename T>
    hipError_t alloc(T **out, size_t bytes) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T *>(p);
        return e;
    }
};

extern "C" const char *rrt_last_error_string(rrt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int rrt_ctx_create(int32_t device_id, rrt_ctx **out) {
    if (!out) return fail(nullptr, RRT_E_ARG, "rrt_ctx_create: out is NULL");
    int ndev = 0;
    HIPCHK(nullptr, hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev)
        return fail(nullptr, RRT_E_ARG, "rrt_ctx_create: device %d of %d", device_id, ndev);
    HIPCHK(nullptr, hipSetDevice(device_id));
    rrt_ctx *c = new rrt_ctx();
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&c->max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&c->num_cu, hipDeviceAttributeMultiprocessorCount, device_id);
    if (e != hipSuccess) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, RRT_E_HIP, "rrt_ctx_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return RRT_OK;
}

extern "C" int rrt_ctx_destroy(rrt_ctx *ctx) {
    if (!ctx) return RRT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->single) rrt_batch_destroy(ctx->single);
    (void)rrt_comm_destroy(ctx);
    if (ctx->og_buf) (void)hipFree(ctx->og_buf);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RRT_OK;
}

extern "C" int rrt_set_grid(rrt_ctx *ctx, const uint8_t *og_nonzero, int32_t W, int32_t H) {
    if (!ctx || !og_nonzero || W < 1 || H < 1) return fail(ctx, RRT_E_ARG, "rrt_set_grid: bad argument");
    if (W > RRT_GRID_MAX || H > RRT_GRID_MAX)
        return fail(ctx, RRT_E_UNSUPPORTED, "rrt_set_grid: %dx%d exceeds %d cells per axis (16-bit packed coordinates)", W, H, RRT_GRID_MAX);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->og_buf && ctx->og_buf_bytes != (size_t)W * H) {
        ctx->grid_gen += 1;
        ctx->og = nullptr;
        HIPCHK(ctx, hipFree(ctx->og_buf));
        ctx->og_buf = nullptr;
    }
    if (!ctx->og_buf) {
        HIPCHK(ctx, hipMalloc(&ctx->og_buf, (size_t)W * H));
        ctx->og_buf_bytes = (size_t)W * H;
    }
    ctx->og = ctx->og_buf;
    ctx->nframes = 1;
    ctx->grid_gen += 1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->og, og_nonzero, (size_t)W * H, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->W = W;
    ctx->H = H;
    return RRT_OK;
}

// ---- device-resident noise grids (counterpart of oggen.perlin_occupancygrid, oggen.py:7-45) ----
namespace {
__device__ __forceinline__ double fade5(double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); }
__device__ __forceinline__ uint32_t f32_order(float v) {  // monotone map float -> uint
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unorder(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// value[f][x][y] = float32(sum over octaves amp * gradient_noise(f/cell, x/cell, y/cell)), evaluated in f64 in exactly the
// order of rrtplanner_amd/oggen.py (_gradient_noise3 / noise_field); running min / max of the float32 values.
__global__ void noise_eval_kernel(int W, int H, int F, int octaves, const int32_t *dims, const double *cells, const double *amps,
                                  const double *grads, float *val, uint32_t *minmax) {
    const size_t total = (size_t)F * W * H;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(p % H), x = (int)((p / H) % W), f = (int)(p / ((size_t)H * W));
        double acc = 0.0;
        size_t goff = 0;
        for (int o = 0; o < octaves; ++o) {
            const int nz = dims[3 * o], nx = dims[3 * o + 1], ny = dims[3 * o + 2];
            const double cell = cells[o];
            const double z = (double)f / cell, xx = (double)x / cell, yy = (double)y / cell;
            const double z0 = floor(z), x0 = floor(xx), y0 = floor(yy);
            const double fz = z - z0, fx = xx - x0, fy = yy - y0;
            const double uz = fade5(fz), ux = fade5(fx), uy = fade5(fy);
            const double *g = grads + goff;
            double out = 0.0;
            for (int dz = 0; dz < 2; ++dz) {
                const double wz = dz ? uz : 1.0 - uz;
                for (int dx = 0; dx < 2; ++dx) {
                    const double wx = dx ? ux : 1.0 - ux;
                    for (int dy = 0; dy < 2; ++dy) {
                        const double wy = dy ? uy : 1.0 - uy;
                        const double *gg = g + 3 * ((((size_t)((int)z0 + dz)) * nx + ((int)x0 + dx)) * ny + ((int)y0 + dy));
                        const double dot = gg[0] * (fz - (double)dz) + gg[1] * (fx - (double)dx) + gg[2] * (fy - (double)dy);
                        out = out + wz * wx * wy * dot;
                    }
                }
            }
            acc = acc + amps[o] * out;
            goff += (size_t)nz * nx * ny * 3;
        }
        const float v = (float)acc;
        val[p] = v;
        const uint32_t k = f32_order(v);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    atomicMin(&minmax[0], lo);
    atomicMax(&minmax[1], hi);
}

// xynoise -= min ; xynoise /= (max - min) ; og = where(xynoise >= thresh, 0, 1)   (oggen.py:40-44, float32 arithmetic)
__global__ void noise_thresh_kernel(size_t total, const float *val, const uint32_t *minmax, float thresh, uint8_t *og) {
    const float mn = f32_unorder(minmax[0]), mx = f32_unorder(minmax[1]);
    const float den = mx - mn;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const float n = (val[p] - mn) / den;
        og[p] = (n >= thresh) ? (uint8_t)0 : (uint8_t)1;
    }
}
}  // namespace

extern "C" int rrt_noise_grids(rrt_ctx *ctx, int32_t W, int32_t H, int32_t frames, float thresh, int32_t octaves, const int32_t *dims,
                               const double *cells, const double *amps, const double *grads, uint8_t *og_out) {
    if (!ctx || W < 1 || H < 1 || frames < 1 || octaves < 1 || !dims || !cells || !amps || !grads)
        return fail(ctx, RRT_E_ARG, "rrt_noise_grids: bad argument");
    if (W > 2048 || H > 2048) return fail(ctx, RRT_E_UNSUPPORTED, "rrt_noise_grids: %dx%d exceeds 2048x2048", W, H);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t total = (size_t)frames * W * H;
    size_t ngrad = 0;
    for (int o = 0; o < octaves; ++o) {
        const int nz = dims[3 * o], nx = dims[3 * o + 1], ny = dims[3 * o + 2];
        if (nz < 2 || nx < 2 || ny < 2 || !(cells[o] > 0.0) || std::floor((frames - 1) / cells[o]) + 2 > nz ||
            std::floor((W - 1) / cells[o]) + 2 > nx || std::floor((H - 1) / cells[o]) + 2 > ny)
            return fail(ctx, RRT_E_ARG, "rrt_noise_grids: lattice %d does not cover the field", o);
        ngrad += (size_t)nz * nx * ny * 3;
    }
    if (ctx->og_buf && ctx->og_buf_bytes != total) {
        ctx->grid_gen += 1;
        ctx->og = nullptr;
        HIPCHK(ctx, hipFree(ctx->og_buf));
        ctx->og_buf = nullptr;
    }
    if (!ctx->og_buf) {
        HIPCHK(ctx, hipMalloc(&ctx->og_buf, total));
        ctx->og_buf_bytes = total;
    }
    ctx->grid_gen += 1;  // og_buf is rewritten (and possibly reallocated) from here on
    DevTmp tmp;
    int32_t *d_dims = nullptr;
    double *d_cells = nullptr, *d_amps = nullptr, *d_grads = nullptr;
    float *d_val = nullptr;
    uint32_t *d_mm = nullptr;
    HIPCHK(ctx, tmp.alloc(&d_dims, (size_t)octaves * 3 * sizeof(int32_t)));
    HIPCHK(ctx, tmp.alloc(&d_cells, (size_t)octaves * sizeof(double)));
    HIPCHK(ctx, tmp.alloc(&d_amps, (size_t)octaves * sizeof(double)));
    HIPCHK(ctx, tmp.alloc(&d_grads, ngrad * sizeof(double)));
    HIPCHK(ctx, tmp.alloc(&d_val, total * sizeof(float)));
    HIPCHK(ctx, tmp.alloc(&d_mm, 2 * sizeof(uint32_t)));
    const uint32_t mm0[2] = {0xffffffffu, 0u};
    HIPCHK(ctx, hipMemcpyAsync(d_dims, dims, (size_t)octaves * 3 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_cells, cells, (size_t)octaves * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_amps, amps, (size_t)octaves * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_grads, grads, ngrad * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_mm, mm0, sizeof mm0, hipMemcpyHostToDevice, ctx->stream));
    const unsigned blocks = (unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(noise_eval_kernel, dim3(blocks), dim3(256), 0, ctx->stream, W, H, frames, octaves, d_dims, d_cells, d_amps, d_grads,
                       d_val, d_mm);
    hipLaunchKernelGGL(noise_thresh_kernel, dim3(blocks), dim3(256), 0, ctx->stream, total, d_val, d_mm, thresh, ctx->og_buf);
    HIPCHK(ctx, hipGetLastError());
    if (og_out) HIPCHK(ctx, hipMemcpyAsync(og_out, ctx->og_buf, total, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->W = W;
    ctx->H = H;
    ctx->nframes = frames;
    ctx->og = ctx->og_buf;
    return RRT_OK;
}

extern "C" int rrt_grid_generation(rrt_ctx *ctx, uint64_t *generation) {
    if (!ctx || !generation) return fail(ctx, RRT_E_ARG, "rrt_grid_generation: NULL");
    *generation = ctx->grid_gen;
    return RRT_OK;
}

extern "C" int rrt_select_frame(rrt_ctx *ctx, int32_t frame) {
    if (!ctx || !ctx->og_buf) return fail(ctx, RRT_E_NOGRID, "rrt_select_frame: no grid");
    if (frame < 0 || frame >= ctx->nframes) return fail(ctx, RRT_E_ARG, "rrt_select_frame: frame %d of %d", frame, ctx->nframes);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->og = ctx->og_buf + (size_t)frame * ctx->W * ctx->H;
    return RRT_OK;
}

// Near-set record grid of one query: cell edge 2^shift pixels, about half the rewire radius, at most MAX_CELLS
// cells.  A cell holds at most 4^shift + 1 nodes (one per pixel, plus xstart once more: rrt.py:425) and at most n + 1.
#ifndef RRT_CELL_DIV
#define RRT_CELL_DIV 2.0  // a cell is at most r_rewire / RRT_CELL_DIV wide (and at least 16) ...
#endif
#ifndef RRT_CELL_DIV_PIPE
#define RRT_CELL_DIV_PIPE 4.0  // ... for a query that the one-CU pipeline runs (rrt_pipe.h: its streams leave out the cells beyond the radius,
                               // and a step costs them the same however many cells begin in it).  Measured: profiles/r03_experiments.md
                               // The divisor is fixed when the query is SET (rrt_batch_set_query: team == 1 and not Informed); the kernel is
                               // picked at LAUNCH.  A team batch that the CU registry shrank to one CU per query therefore runs the pipeline
                               // on the coarser cells, and a continuation (block kernel) may run on the finer ones.  Both kernels derive
                               // the cells they stream from the radius and cell_shift, so results are the same; only the tuning differs,
                               // and timings of such launches are not comparable with the tuned shape (rrt_batch_team_info says which ran).
#endif
static void cell_geometry(int W, int H, int64_t r2, int n, double div, int &shift, int &ncx, int &ncy, int &cap) {
    double r = std::sqrt((double)(r2 < 1 ? 1 : r2));
    shift = 4;
    while ((1 << (shift + 1)) <= r / div && shift < 11) ++shift;
    for (;; ++shift) {
        ncx = (W + (1 << shift) - 1) >> shift;
        ncy = (H + (1 << shift) - 1) >> shift;
        if ((long long)ncx * ncy <= MAX_CELLS) break;
    }
    long long c = (1LL << (2 * shift)) + 1;
    cap = (int)(c < (long long)n + 1 ? c : (long long)n + 1);
}

static int64_t cell_records_needed(int W, int H, int n) {
    int64_t best = 0;
    for (int s = 4; s <= 11; ++s) {
        const int64_t ncx = (W + (1 << s) - 1) >> s, ncy = (H + (1 << s) - 1) >> s;
        if (ncx * ncy > MAX_CELLS) continue;
        int64_t c = (1LL << (2 * s)) + 1;
        if (c > (int64_t)n + 1) c = (int64_t)n + 1;
        if (ncx * ncy * c > best) best = ncx * ncy * c;
    }
    return best;
}

// Compute units that launches in flight have claimed, per device (every batch of this process: all contexts, all host threads).
// A team kernel's members wait for each other, so they must all be resident at once: a launch looks here and takes the largest
// team whose workgroups fit NEXT TO what is in flight (one CU per query if nothing else does); rrt_batch_sync gives the claim back.
namespace {
std::mutex g_cu_mutex;
std::vector<int> g_cu_claimed;  // [device]
int cu_claim(int device, int num_cu, int want_cus, int min_cus) {
    // claims `want_cus` if that many are free, else nothing (returns the free count, negative-free as 0, through *the caller's retry*)
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    if ((int)g_cu_claimed.size() <= device) g_cu_claimed.resize((size_t)device + 1, 0);
    const int free_cus = num_cu - g_cu_claimed[(size_t)device];
    if (want_cus <= free_cus || want_cus <= min_cus) {
        g_cu_claimed[(size_t)device] += want_cus;
        return -1;  // granted
    }
    return free_cus < 0 ? 0 : free_cus;
}
void cu_release(int device, int cus) {
    if (cus <= 0) return;
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    if ((int)g_cu_claimed.size() > device) g_cu_claimed[(size_t)device] -= cus;
}
}  // namespace

extern "C" int rrt_batch_destroy(rrt_batch *b) {
    if (!b) return RRT_OK;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    void *ptrs[] = {b->d_desc,  b->d_samples,   b->d_slab,        b->d_bitmap, b->d_unitball,  b->d_cellrec,
                    b->d_spill, b->d_cbest_log, b->d_nearest_log, b->d_j_log,  b->d_accept_log, b->d_cellcnt,
                    b->d_team,  b->d_kids,      b->d_frontier,    b->d_vsoln,  b->d_heading,   b->d_shead,
                    b->d_dubpath};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    cu_release(b->ctx->device, b->claimed_cus);
    b->claimed_cus = 0;
    if (b->ctx->single == b) b->ctx->single = nullptr;
    if (b->ctx->gather_owner == b) b->ctx->gather_owner = nullptr;
    b->h_desc.release();
    delete b;
    return RRT_OK;
}

// The team shape for Q queries on `cus` compute units: the largest team (CUs per query) with every member of every team resident
// at once.  Blocks are dealt round-robin to the 8 XCDs, so block = member * stride + query with stride = 0 (mod 8) keeps a team of
// up to 16 on one XCD (one L2); larger teams use stride = 4 / 2 (mod 8): two / four XCDs with 16 members each (speed only).
// A pipelined team (g workers + a committer) beats an unpipelined one of twice its size (config 2: 8+1 CUs 24.7 ms vs 16 CUs
// 27.3 ms; config 4's share: 3+1 CUs vs 4 CUs, profiles/r02_experiments.md).
struct TeamShape {
    int team = 1, qpad = 0;
    bool pipe = false;
    int cus() const { return team > 1 ? qpad * (team + (pipe ? 1 : 0)) : 0; }  // workgroups that must be resident together
};
static TeamShape pick_team(int Q, int want, bool allow_pipe, int cus);
// The batch got two CUs per query without a committer, by itself (no cap from the caller, pipelines allowed): see rrt_batch_launch.
static bool two_cus_run_the_pipeline(const rrt_batch *b);
static TeamShape pick_team(int Q, int want, bool allow_pipe, int cus) {
    TeamShape t;
    t.qpad = (Q + 7) & ~7;
    auto stride_of = [&](int g) {
        const int step = g <= 16 ? 8 : (g == 32 ? 4 : 2);
        int stride = ((Q + step - 1) / step) * step;
        if (g == 32 && stride % 8 == 0) stride += 4;
        if (g == 64 && stride % 4 == 0) stride += 2;
        return stride;
    };
    for (int g : {2, 4, 8, 16, 32, 64})
        if (g <= want && stride_of(g) * g <= cus) {
            t.team = g;
            t.qpad = stride_of(g);
        }
    int pipe_g = 0;
    if (allow_pipe)
        for (int g : {2, 3, 4, 8, 16, 32, 64})
            if (g <= want && stride_of(g) * (g + 1) <= cus) pipe_g = g;
    if (pipe_g != 0 && t.team <= 2 * pipe_g) {
        t.team = pipe_g;
        t.qpad = stride_of(pipe_g);
        t.pipe = true;
    }
    return t;
}

extern "C" int rrt_batch_create(rrt_ctx *ctx, int32_t Q, int32_t n_cap, uint32_t flags, rrt_batch **out) {
    if (!ctx || !out || Q < 1 || n_cap < 1) return fail(ctx, RRT_E_ARG, "rrt_batch_create: bad argument");
    if (!ctx->og) return fail(ctx, RRT_E_NOGRID, "rrt_batch_create: call rrt_set_grid first");
    if (ctx->W > RRT_GRID_FAST || ctx->H > RRT_GRID_FAST)
        return fail(ctx, RRT_E_UNSUPPORTED, "rrt_batch_create: the expansion kernels take grids up to %dx%d (24-bit squared distances); a %dx%d grid runs "
                    "through the host-driven path (rrt_tree_query)", RRT_GRID_FAST, RRT_GRID_FAST, ctx->W, ctx->H);
    if ((long long)n_cap + 1 > 64LL * CHUNK)
        return fail(ctx, RRT_E_UNSUPPORTED, "rrt_batch_create: n=%d exceeds %d nodes", n_cap, 64 * CHUNK - 1);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rrt_batch *b = new rrt_batch();
    b->ctx = ctx;
    b->Q = Q;
    b->n_cap = n_cap;
    b->flags = flags;
    b->gridW = ctx->W;
    b->gridH = ctx->H;
    // whole 4096-node steps: the block kernel scans complete steps, unfilled slots hold a copy of node 0
    b->node_stride = ((n_cap + 1 + CHUNK - 1) / CHUNK) * CHUNK;
    b->bitmap_words = (int32_t)(((size_t)ctx->W * ctx->H + 31) / 32);
    int chunks = (n_cap + 1 + CHUNK - 1) / CHUNK;
    if ((flags & RRT_FLAG_DUBINS) && (flags & RRT_FLAG_REWIRE)) {
        delete b;
        return fail(ctx, RRT_E_UNSUPPORTED, "rrt_batch_create: the opt-in rewire is not available for the Dubins planners");
    }
    // the opt-in rewire and the Dubins planners run on the one-sample-per-iteration kernel
    b->use_block = !(flags & (RRT_FLAG_SERIAL | RRT_FLAG_REWIRE | RRT_FLAG_DUBINS));
    b->dub_block = (flags & RRT_FLAG_DUBINS) && !(flags & RRT_FLAG_SERIAL);
    if (b->use_block && !(flags & RRT_FLAG_NOTEAM)) {
        int want = (int)((flags >> 8) & 0x7fu);
        if (want == 0) want = TEAM_MAX;
#ifdef RRT_STAMPS
        if (const char *e = getenv("RRT_TEAM")) want = atoi(e);  // diagnostic build only: cap the team size
#endif
        b->team_want = want;
        const TeamShape ts = pick_team(Q, want, !(flags & RRT_FLAG_NOPIPE), ctx->num_cu);  // the shape on an otherwise idle device
        b->team = ts.team;
        b->team_qpad = ts.qpad;
        b->pipe_team = ts.pipe;
    }
    b->spill_stride = chunks * CHUNK * (b->team + 1);  // per member (and a pipelined team's committer): 256 parked entries per wave and node chunk; also go2goal's cost array
    {   // block kernel LDS: [node cache | cell fill counts 16 KiB | the waves' parked-entry lists 64 KiB (teams of up to 4 workers and
        // single CUs, where one wave resolves a sample: rrt_block.h LDSLIST)]
        const size_t budget = (size_t)ctx->max_lds - block_kernel_static_lds(b->team);
        const size_t fixed = (size_t)MAX_CELLS * sizeof(uint32_t);
        if (b->dub_block) b->rec_stride = cell_records_needed(ctx->W, ctx->H, n_cap);
        if (b->use_block) {
            int nc = (int)((budget - fixed) / ((size_t)CHUNK * sizeof(uint32_t)));
            b->blk_lds_chunks = nc > chunks ? chunks : nc;
            int nc16 = (int)((budget - fixed - BLOCK_LIST_LDS_BYTES) / ((size_t)CHUNK * sizeof(uint32_t)));
            b->blk_lds_chunks16 = nc16 > chunks ? chunks : nc16;
            b->rec_stride = cell_records_needed(ctx->W, ctx->H, n_cap);
        }
    }
    b->lds_chunks = chunks < 1 ? 1 : (chunks > MAX_LDS_CHUNKS ? MAX_LDS_CHUNKS : chunks);
    if ((flags & RRT_FLAG_DUBINS) && b->lds_chunks > 5) b->lds_chunks = 5;  // the Dubins kernel keeps the packed near set (36 KiB) in LDS too
    if (hipError_t e_ = b->h_desc.alloc((size_t)Q); e_ != hipSuccess) {
        rrt_batch_destroy(b);
        return fail(ctx, RRT_E_HIP, "hipHostMalloc(%zu): %s", (size_t)Q * sizeof(QDesc), hipGetErrorString(e_));
    }
    for (auto &d : b->h_desc) d.status = ST_IDLE;
    const size_t q = (size_t)Q;
#define ALLOC(ptr, bytes)                                   \
    do {                                                    \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes)); \
        if (e_ != hipSuccess) {                             \
            rrt_batch_destroy(b);                           \
            return fail(ctx, RRT_E_HIP, "hipMalloc(%zu): %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                   \
    } while (0)
    ALLOC(b->d_desc, q * sizeof(QDesc));
    ALLOC(b->d_samples, q * n_cap * sizeof(uint32_t));
    // [vcost f64 | nodes u32 | parent i32][Q][node_stride], then {status, j, vgoal, found} i32 per query (written by rrt_gather)
    b->slab_bytes = q * b->node_stride * (sizeof(double) + sizeof(uint32_t) + sizeof(int32_t)) + q * 4 * sizeof(int32_t);
    ALLOC(b->d_slab, b->slab_bytes);
    b->d_vcost = reinterpret_cast<double *>(b->d_slab);
    b->d_nodes = reinterpret_cast<uint32_t *>(b->d_slab + q * b->node_stride * sizeof(double));
    b->d_parent = reinterpret_cast<int32_t *>(b->d_slab + q * b->node_stride * (sizeof(double) + sizeof(uint32_t)));
    ALLOC(b->d_bitmap, q * b->bitmap_words * sizeof(uint32_t));
    ALLOC(b->d_spill, q * b->spill_stride * sizeof(uint2));
    if (b->use_block || b->dub_block) {
        ALLOC(b->d_cellrec, q * (size_t)b->rec_stride * sizeof(uint4));
        ALLOC(b->d_cellcnt, q * (size_t)MAX_CELLS * sizeof(uint32_t));
        if (b->use_block && b->team > 1) ALLOC(b->d_team, q * (size_t)TEAM_BYTES);
    }
    if (flags & RRT_FLAG_REWIRE) {
        ALLOC(b->d_kids, 3 * q * b->node_stride * sizeof(int32_t));
        ALLOC(b->d_frontier, 2 * q * b->node_stride * sizeof(uint32_t));
        ALLOC(b->d_vsoln, q * b->node_stride * sizeof(int32_t));
    }
    if (flags & RRT_FLAG_DUBINS) {
        ALLOC(b->d_heading, q * b->node_stride);
        ALLOC(b->d_shead, q * n_cap);
        if (!b->dub_block) ALLOC(b->d_dubpath, q * (size_t)(NWAVE * WCAP) * 5 * sizeof(double));
    }
    if (flags & RRT_FLAG_LOGS) {
        ALLOC(b->d_nearest_log, q * n_cap * sizeof(int32_t));
        ALLOC(b->d_accept_log, q * n_cap * sizeof(uint8_t));
        ALLOC(b->d_cbest_log, q * n_cap * sizeof(double));
        ALLOC(b->d_j_log, q * n_cap * sizeof(int32_t));
    }
#undef ALLOC
    hipError_t e = hipMemsetAsync(b->d_slab, 0, b->slab_bytes, ctx->stream);
    if (e == hipSuccess) e = hipEventCreate(&b->ev0);
    if (e == hipSuccess) e = hipEventCreate(&b->ev1);
    if (e == hipSuccess) e = hipMemcpyAsync(b->d_desc, b->h_desc.data(), q * sizeof(QDesc), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        rrt_batch_destroy(b);
        return fail(ctx, RRT_E_HIP, "rrt_batch_create: %s", hipGetErrorString(e));
    }
    *out = b;
    return RRT_OK;
}

static void arm_desc(QDesc &d) {
    d.status = ST_RUNNING;
    d.i = 0;
    d.j = 1;
    d.nsoln = 0;
    d.vbest_soln = -1;
    d.cmin_soln = INFINITY;
    d.vgoal = 0;
    d.found = 0;
    d.i_switch = d.n;
    d.ub_offset = 0;
    d.ub_count = 0;
    d.sum_j = d.sum_cells_nn = d.sum_near = d.sum_cells_cand = d.n_los_cand = 0;
    d.n_rewired = d.n_propagated = 0;
    d.n_words = 0;
    for (auto &c : d.cyc) c = 0;
    for (auto &c : d.wcyc) c = 0;
#ifdef RRT_STAMPS
    for (auto &c : d.dbg2) c = 0;
    for (auto &c : d.ts) c = 0;
#endif
}

extern "C" int rrt_batch_set_query(rrt_batch *b, int32_t q, const rrt_query *qu) {
    if (!b || !qu) return fail(nullptr, RRT_E_ARG, "rrt_batch_set_query: NULL");
    rrt_ctx *ctx = b->ctx;
    if (q < 0 || q >= b->Q) return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: q=%d of %d", q, b->Q);
    if (qu->alg < 0 || qu->alg > RRT_ALG_DUBINS_STAR) return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: alg=%d", qu->alg);
    const bool dub = qu->alg >= RRT_ALG_DUBINS;
    if (dub != ((b->flags & RRT_FLAG_DUBINS) != 0))
        return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: alg=%d on a batch created %s RRT_FLAG_DUBINS", qu->alg, dub ? "without" : "with");
    if (qu->n < 1 || qu->n > b->n_cap) return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: n=%d, capacity %d", qu->n, b->n_cap);
    if (dub) {
        if (!qu->headings || !(qu->rho > 0.0) || !std::isfinite(qu->rho) || qu->nh < 1 || qu->nh > 256 || qu->hs < 0 || qu->hs >= qu->nh ||
            qu->hg < 0 || qu->hg >= qu->nh)
            return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: Dubins query needs headings, rho > 0, 1 <= nh <= 256 and start / goal headings below nh");
        for (int k = 0; k < qu->n; ++k)
            if (qu->headings[k] < 0 || qu->headings[k] >= qu->nh) return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: heading of sample %d outside [0, %d)", k, qu->nh);
    }
    if (b->gridW != ctx->W || b->gridH != ctx->H)
        return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: grid changed shape since rrt_batch_create");
    const int W = ctx->W, H = ctx->H;
    if (qu->xs[0] < 0 || qu->xs[0] >= W || qu->xs[1] < 0 || qu->xs[1] >= H || qu->xg[0] < 0 || qu->xg[0] >= W ||
        qu->xg[1] < 0 || qu->xg[1] >= H)
        return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: start/goal outside the %dx%d grid", W, H);
    if (!qu->samples && !qu->samples_packed) return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: samples is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    b->stage.resize((size_t)qu->n);
    for (int k = 0; k < qu->n; ++k) {
        int x, y;
        if (qu->samples_packed) {
            x = (int)(qu->samples_packed[k] & 0xffffu);
            y = (int)(qu->samples_packed[k] >> 16);
        } else {
            x = qu->samples[2 * k];
            y = qu->samples[2 * k + 1];
        }
        if (x < 0 || x >= W || y < 0 || y >= H) return fail(ctx, RRT_E_ARG, "rrt_batch_set_query: sample %d outside the grid", k);
        b->stage[(size_t)k] = ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16);
    }
    QDesc &d = b->h_desc[(size_t)q];
    d = QDesc{};
    d.alg = qu->alg;
    d.n = qu->n;
    d.xs[0] = qu->xs[0];
    d.xs[1] = qu->xs[1];
    d.xg[0] = qu->xg[0];
    d.xg[1] = qu->xg[1];
    const int64_t cap = 1 << 24;  // any d2 on a 2048x2048 grid is below this
    d.r2_rewire = (uint32_t)(qu->r2_rewire < 0 ? 0 : (qu->r2_rewire > cap ? cap : qu->r2_rewire));
    d.goal_d2 = (uint32_t)(qu->goal_d2 < 0 ? 0 : (qu->goal_d2 > cap ? cap : qu->goal_d2));
    for (int k = 0; k < 4; ++k) d.C[k] = qu->C[k];
    if (dub) {
        d.rho = qu->rho;
        d.nh = qu->nh;
        d.hs = qu->hs;
        d.hg = qu->hg;
        b->stage8.resize((size_t)qu->n);
        for (int k = 0; k < qu->n; ++k) b->stage8[(size_t)k] = (uint8_t)qu->headings[k];
        HIPCHK(ctx, hipMemcpyAsync(b->d_shead + (size_t)q * b->n_cap, b->stage8.data(), (size_t)qu->n, hipMemcpyHostToDevice, ctx->stream));
    }
    const bool pipe1 = b->use_block && (b->team == 1 || two_cus_run_the_pipeline(b)) && !(b->flags & RRT_FLAG_NOPIPE1) && qu->alg != RRT_ALG_INFORMED;
    cell_geometry(W, H, (int64_t)d.r2_rewire, b->n_cap, pipe1 ? RRT_CELL_DIV_PIPE : RRT_CELL_DIV, d.cell_shift, d.ncx, d.ncy, d.cell_cap);
    arm_desc(d);
    HIPCHK(ctx, hipMemcpyAsync(b->d_samples + (size_t)q * b->n_cap, b->stage.data(), (size_t)qu->n * sizeof(uint32_t),
                               hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->d_desc + q, &d, sizeof(QDesc), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // staging buffer is reused
    return RRT_OK;
}

extern "C" int rrt_batch_set_unitball(rrt_batch *b, int32_t q, const double *unitball, int32_t count, int32_t ub_offset) {
    if (!b || !unitball) return fail(nullptr, RRT_E_ARG, "rrt_batch_set_unitball: NULL");
    rrt_ctx *ctx = b->ctx;
    if (q < 0 || q >= b->Q || count < 0 || ub_offset < 0 || (long long)ub_offset + count > b->n_cap)
        return fail(ctx, RRT_E_ARG, "rrt_batch_set_unitball: bad range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!b->d_unitball) HIPCHK(ctx, hipMalloc((void **)&b->d_unitball, (size_t)b->Q * 2 * b->n_cap * sizeof(double)));
    QDesc &d = b->h_desc[(size_t)q];
    if (d.status != ST_NEED_UB) return fail(ctx, RRT_E_ARG, "rrt_batch_set_unitball: query %d is not waiting for unit-ball data", q);
    // entries are indexed by iteration - ub_offset on the device
    HIPCHK(ctx, hipMemcpyAsync(b->d_unitball + (size_t)q * 2 * b->n_cap, unitball, (size_t)count * 2 * sizeof(double),
                               hipMemcpyHostToDevice, ctx->stream));
    d.ub_offset = ub_offset;
    d.ub_count = count;
    d.status = ST_RUNNING;
    HIPCHK(ctx, hipMemcpyAsync(b->d_desc + q, &d, sizeof(QDesc), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}

extern "C" int rrt_batch_rearm(rrt_batch *b) {
    if (!b) return fail(nullptr, RRT_E_ARG, "rrt_batch_rearm: NULL");
    rrt_ctx *ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (auto &d : b->h_desc)
        if (d.status != ST_IDLE) arm_desc(d);
    HIPCHK(ctx, hipMemcpyAsync(b->d_desc, b->h_desc.data(), (size_t)b->Q * sizeof(QDesc), hipMemcpyHostToDevice, ctx->stream));
    return RRT_OK;
}

static BatchView make_view(rrt_batch *b) {
    BatchView v{};
    v.desc = b->d_desc;
    v.samples = b->d_samples;
    v.nodes = b->d_nodes;
    v.vcost = b->d_vcost;
    v.parent = b->d_parent;
    v.bitmap = b->d_bitmap;
    v.spill = b->d_spill;
    v.unitball = b->d_unitball;
    v.nearest_log = b->d_nearest_log;
    v.accept_log = b->d_accept_log;
    v.cbest_log = b->d_cbest_log;
    v.j_log = b->d_j_log;
    v.og = b->ctx->og;
    v.W = b->ctx->W;
    v.H = b->ctx->H;
    v.n_cap = b->n_cap;
    v.node_stride = b->node_stride;
    v.bitmap_words = b->bitmap_words;
    v.lds_chunks = b->lds_chunks;
    v.spill_stride = b->spill_stride;
    v.cellrec = b->d_cellrec;
    v.cellcnt = b->d_cellcnt;
    v.rec_stride = b->rec_stride;
    v.team = b->d_team;
    v.Q = b->Q;
    v.team_qpad = b->team_qpad;
    v.team_fault = (b->flags & RRT_FLAG_TEAM_FAULT) ? 1 : 0;
    v.member0 = 0;
    if (b->d_kids) {
        const size_t qs = (size_t)b->Q * b->node_stride;
        v.kid_first = b->d_kids;
        v.kid_next = b->d_kids + qs;
        v.kid_prev = b->d_kids + 2 * qs;
        v.frontier = b->d_frontier;
        v.vsoln = b->d_vsoln;
    }
    v.heading = b->d_heading;
    v.sample_heading = b->d_shead;
    v.dub_path = b->d_dubpath;
    return v;
}

typedef void (*block_kernel_fn)(BatchView);

// wide: a pipelined team of 2 workers with 32 samples per member (rrt_block.h, BSM > 16).  Measured (profiles/r03_experiments.md):
// config 4's query on 2 + 1 CUs 8.92 -> 7.87 ms; three workers with 21 samples each gained nothing (6.32 vs 6.25 ms) and are not built.
template <bool INF>
static block_kernel_fn block_kernel_fn_inf(int team, bool pipe, bool wide) {
    if constexpr (!INF) {  // (the launch never takes the wide team for a batch with Informed queries)
        if (pipe && wide && team == 2) return rrt_expand_block_kernel<2, 32, true, false>;
    }
    if (pipe) {
        switch (team) {
            case 64: return rrt_expand_block_kernel<64, 1, true, INF>;
            case 32: return rrt_expand_block_kernel<32, 2, true, INF>;
            case 16: return rrt_expand_block_kernel<16, 4, true, INF>;
            case 8: return rrt_expand_block_kernel<8, 8, true, INF>;
            case 4: return rrt_expand_block_kernel<4, 16, true, INF>;
            case 3: return rrt_expand_block_kernel<3, 16, true, INF>;
            default: return rrt_expand_block_kernel<2, 16, true, INF>;
        }
    }
    switch (team) {
        case 64: return rrt_expand_block_kernel<64, 1, false, INF>;
        case 32: return rrt_expand_block_kernel<32, 2, false, INF>;
        case 16: return rrt_expand_block_kernel<16, 4, false, INF>;
        case 8: return rrt_expand_block_kernel<8, 8, false, INF>;
        case 4: return rrt_expand_block_kernel<4, 16, false, INF>;
        case 2: return rrt_expand_block_kernel<2, 16, false, INF>;
        default: return rrt_expand_block_kernel<1, 16, false, INF>;
    }
}

static block_kernel_fn block_kernel_fn_of(int team, bool pipe, bool inf, bool wide = false) {
    return inf ? block_kernel_fn_inf<true>(team, pipe, wide) : block_kernel_fn_inf<false>(team, pipe, wide);
}

static const void *block_kernel_of(int team, bool pipe, bool inf, bool wide = false) {
    return reinterpret_cast<const void *>(block_kernel_fn_of(team, pipe, inf, wide));
}

// static LDS of the kernels a batch of this team size may launch: its own variants and the one-CU kernel that continues
// a batch after a hand-off timed out (rrt_batch_sync)
static size_t block_kernel_static_lds(int team) {
    hipFuncAttributes a{};
    size_t worst = 0;
    for (int g : {1, 2, 3, 4, 8, 16, 32, 64})  // (a launch may run any smaller team when other launches hold compute units)
        for (bool pipe : {false, true})
            for (bool inf : {false, true}) {
                if (g > team || (pipe && g < 2)) continue;
                if (!pipe && g == 3) continue;  // (three workers exist only as a pipelined team)
                for (bool wide : {false, true}) {
                    if (wide && !(pipe && g == 2)) continue;
                    if (hipFuncGetAttributes(&a, block_kernel_of(g, pipe, inf, wide)) != hipSuccess) return 16384;
                    worst = a.sharedSizeBytes > worst ? a.sharedSizeBytes : worst;
                }
            }
    return (worst + 255) & ~(size_t)255;
}

static size_t expand_lds_bytes(int lds_chunks) {
    return (size_t)lds_chunks * CHUNK * sizeof(uint32_t);  // dynamic part: the node cache (lists and slots are static LDS)
}

static bool two_cus_run_the_pipeline(const rrt_batch *b) {
    return b->use_block && b->team == 2 && !b->pipe_team && b->team_want >= TEAM_MAX && !(b->flags & (RRT_FLAG_NOPIPE | RRT_FLAG_NOPIPE1));
}

extern "C" int rrt_batch_launch(rrt_batch *b) {
    if (!b) return fail(nullptr, RRT_E_ARG, "rrt_batch_launch: NULL");
    rrt_ctx *ctx = b->ctx;
    if (!ctx->og) return fail(ctx, RRT_E_NOGRID, "rrt_batch_launch: no grid");
    if (b->gridW != ctx->W || b->gridH != ctx->H) return fail(ctx, RRT_E_ARG, "rrt_batch_launch: grid changed shape");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!b->one_cu_once) b->ms_before = 0.f;
    BatchView v = make_view(b);
    dim3 ig((unsigned)((b->bitmap_words + 255) / 256 > 64 ? 64 : (b->bitmap_words + 255) / 256), (unsigned)b->Q);
    if (b->use_block) {
        // after a hand-off timed out this one launch continues the batch with one CU per query; the team size the batch was
        // created with stays and the next launch uses it again
        int team = b->one_cu_once ? 1 : b->team;
        bool pipe_shape = b->pipe_team;
        const bool continuation = b->one_cu_once;  // (of a launch that stopped at a block boundary: the block kernel takes it from there)
        if (two_cus_run_the_pipeline(b) && !continuation) {
            // 86 - 128 queries: two CUs per query fit, a third (the committer of a pipelined team) does not.  The unpipelined team of
            // two is slower than the barrier-free pipeline on ONE of them (config 4's query x 128: 12.7 ms against 11.0; config 2's:
            // 45.6 against 37.4, profiles/r04_experiments.md §11) -- unless the batch holds Informed queries, which that kernel does not run.
            bool any_inf = false;
            for (const QDesc &d : b->h_desc)
                if (d.status == ST_RUNNING && d.alg == 2) any_inf = true;
            if (!any_inf) team = 1;
        }
        b->one_cu_once = false;
        cu_release(ctx->device, b->claimed_cus);  // (a launch that was never synchronised)
        b->claimed_cus = 0;
        if (team > 1) {
            // every member of every team must be resident at once: claim the CUs, or take the largest team that fits next to the
            // launches in flight on this device (other batches, other contexts, other host threads of this process)
            TeamShape ts;
            ts.team = team;
            ts.qpad = b->team_qpad;
            ts.pipe = pipe_shape;
            for (;;) {
                const int free_cus = cu_claim(ctx->device, ctx->num_cu, ts.cus(), 0);
                if (free_cus < 0) break;  // granted
                TeamShape smaller = pick_team(b->Q, b->team_want < ts.team ? b->team_want : ts.team, pipe_shape, free_cus);
                if (smaller.team >= ts.team && smaller.pipe == ts.pipe) smaller.team = 1;  // (the registry changed in between: do not loop)
                ts = smaller;
                if (ts.team <= 1) break;
            }
            if (ts.team != team) b->shrunk += 1;
            team = ts.team;
            pipe_shape = ts.pipe;
            v.team_qpad = ts.qpad;
            b->claimed_cus = team > 1 ? ts.cus() : 0;
        }
        if (team <= 1) {  // one CU per query needs no co-residency, but its workgroups occupy CUs all the same
            team = 1;
            b->claimed_cus = b->Q < ctx->num_cu ? b->Q : ctx->num_cu;
            (void)cu_claim(ctx->device, ctx->num_cu, b->claimed_cus, b->claimed_cus);
        }
        const bool lists = team <= 4;  // one wave per sample: its parked entries stay in LDS
        v.lds_chunks = lists ? b->blk_lds_chunks16 : b->blk_lds_chunks;
        const size_t blk_lds_bytes = (size_t)MAX_CELLS * sizeof(uint32_t) + (lists ? BLOCK_LIST_LDS_BYTES : 0) + (size_t)v.lds_chunks * CHUNK * sizeof(uint32_t);
        // a two-deep pipeline of super-blocks (one more workgroup per team, which only commits) for teams of 8 and more
        bool pipe = team > 1 && pipe_shape;
#ifdef RRT_STAMPS
        if (const char *e = getenv("RRT_PIPE")) pipe = pipe && atoi(e) != 0;  // diagnostic build only
#endif
        b->pipe = pipe;
        bool inf = false;  // any Informed query in this launch?
        for (const QDesc &d : b->h_desc)
            if (d.status == ST_RUNNING && d.alg == 2) inf = true;
#ifndef RRT_NO_WIDE
        // a pipelined team of 2 workers: 32 samples per member instead of 16 (the waves that are through take the extra ones; the
        // hand-overs of a block are shared by 64 samples instead of 32) -- unless a query's near-set radius is below a cell (that
        // variant has no brute-force scan)
        bool wide = pipe && !inf && team == 2;
        for (const QDesc &d : b->h_desc)
            if (d.status == ST_RUNNING && d.alg != RRT_ALG_STANDARD && d.r2_rewire < 257u) wide = false;
#else
        const bool wide = false;
#endif
        b->last_wide = wide;
        b->last_team = team;
        b->last_inf = inf;
        b->last_pipe1 = false;
        if (team == 1 && !inf && !continuation && !(b->flags & RRT_FLAG_NOPIPE1)) {
            // one CU per query, RRTStandard / RRTStar: the barrier-free pipeline (rrt_pipe.h; static LDS only)
            b->last_pipe1 = true;
            hipLaunchKernelGGL(rrt_init_kernel<0>, ig, dim3(256), 0, ctx->stream, v);
            HIPCHK(ctx, hipEventRecord(b->ev0, ctx->stream));
            hipLaunchKernelGGL(rrt_pipe_kernel, dim3((unsigned)b->Q), dim3(TPB), 0, ctx->stream, v);
            HIPCHK(ctx, hipEventRecord(b->ev1, ctx->stream));
            HIPCHK(ctx, hipGetLastError());
            b->timed = true;
            return RRT_OK;
        }
        HIPCHK(ctx, raise_dynamic_lds(ctx->device, block_kernel_of(team, pipe, inf, wide), (int)blk_lds_bytes));
        hipLaunchKernelGGL(rrt_init_kernel<0>, ig, dim3(256), 0, ctx->stream, v);
        if (team > 1) HIPCHK(ctx, hipMemsetAsync(b->d_team, 0, (size_t)b->Q * TEAM_BYTES, ctx->stream));  // every polled word, every launch
        HIPCHK(ctx, hipEventRecord(b->ev0, ctx->stream));
        const dim3 tg(team > 1 ? (unsigned)(v.team_qpad * (team + (pipe ? 1 : 0))) : (unsigned)b->Q);
        hipLaunchKernelGGL(block_kernel_fn_of(team, pipe, inf, wide), tg, dim3(TPB), blk_lds_bytes, ctx->stream, v);
        HIPCHK(ctx, hipEventRecord(b->ev1, ctx->stream));
        HIPCHK(ctx, hipGetLastError());
        b->timed = true;
        return RRT_OK;
    }
    cu_release(ctx->device, b->claimed_cus);
    b->claimed_cus = b->Q < ctx->num_cu ? b->Q : ctx->num_cu;  // one workgroup per query: no co-residency needed, the CUs are busy all the same
    (void)cu_claim(ctx->device, ctx->num_cu, b->claimed_cus, b->claimed_cus);
    if (b->dub_block) {
        hipLaunchKernelGGL(rrt_init_kernel<0>, ig, dim3(256), 0, ctx->stream, v);
        HIPCHK(ctx, hipEventRecord(b->ev0, ctx->stream));
        hipLaunchKernelGGL(rrt_dubins_block_kernel, dim3((unsigned)b->Q), dim3(TPB), 0, ctx->stream, v);
        HIPCHK(ctx, hipEventRecord(b->ev1, ctx->stream));
        HIPCHK(ctx, hipGetLastError());
        b->timed = true;
        return RRT_OK;
    }
    const size_t lds = expand_lds_bytes(b->lds_chunks);
    typedef void (*serial_kernel_fn)(BatchView);
    const serial_kernel_fn kern = (b->flags & RRT_FLAG_DUBINS)  ? static_cast<serial_kernel_fn>(rrt_expand_kernel<false, true>)
                                  : (b->flags & RRT_FLAG_REWIRE) ? static_cast<serial_kernel_fn>(rrt_expand_kernel<true, false>)
                                                                 : static_cast<serial_kernel_fn>(rrt_expand_kernel<false, false>);
    if (b->serial_lds_static == 0) {
        hipFuncAttributes fa{};
        HIPCHK(ctx, hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
        b->serial_lds_static = fa.sharedSizeBytes + 1;
    }
    const size_t lds_static = b->serial_lds_static - 1;
    if ((int)(lds + lds_static) > ctx->max_lds) return fail(ctx, RRT_E_HIP, "LDS request %zu exceeds %d", lds + lds_static, ctx->max_lds);
    HIPCHK(ctx, raise_dynamic_lds(ctx->device, reinterpret_cast<const void *>(kern), (int)lds));
    hipLaunchKernelGGL(rrt_init_kernel<0>, ig, dim3(256), 0, ctx->stream, v);
    HIPCHK(ctx, hipEventRecord(b->ev0, ctx->stream));
    hipLaunchKernelGGL(kern, dim3((unsigned)b->Q), dim3(TPB), lds, ctx->stream, v);
    HIPCHK(ctx, hipEventRecord(b->ev1, ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    b->timed = true;
    return RRT_OK;
}

extern "C" int rrt_batch_sync(rrt_batch *b) {
    if (!b) return fail(nullptr, RRT_E_ARG, "rrt_batch_sync: NULL");
    rrt_ctx *ctx = b->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(b->h_desc.data(), b->d_desc, (size_t)b->Q * sizeof(QDesc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, wait_stream_spin(ctx->stream));
    cu_release(ctx->device, b->claimed_cus);  // the launch is over: its compute units are free for the launches of other batches
    b->claimed_cus = 0;
#ifdef RRT_STAMPS
    if (getenv("RRT_STAMPS_DUMP")) {  // diagnostic build only: who the committer's record fetch waits for (QDesc::dbg2)
        const QDesc &d0 = b->h_desc[0];
        fprintf(stderr, "dbg2 groups of slow blocks (> 32 k) [count, own stream, first barrier, nearest + LoS, top-2 LoS, tail, consume, empty ball]:");
        for (int m = 0; m < 8; ++m) fprintf(stderr, " %llu", d0.dbg2[m]);
        fprintf(stderr, "\ndbg2 groups of the other blocks:");
        for (int m = 8; m < 16; ++m) fprintf(stderr, " %llu", d0.dbg2[m]);
        fprintf(stderr, "\ndbg2 groups queue, fast [count, gctl barrier, pricing, barrier, tests, barrier, queue entries, time before]:");
        for (int m = 16; m < 24; ++m) fprintf(stderr, " %llu", d0.dbg2[m]);
        fprintf(stderr, "\ndbg2 groups queue, slow (> 30 k at the last barrier):");
        for (int m = 24; m < 32; ++m) fprintf(stderr, " %llu", d0.dbg2[m]);
        fprintf(stderr, "\ndbg2 missing-polls:");
        for (int m = 0; m < 64; ++m) fprintf(stderr, " %llu", d0.dbg2[m]);
        fprintf(stderr, "\ndbg2 last-to-arrive:");
        for (int m = 0; m < 64; ++m) fprintf(stderr, " %llu", d0.dbg2[64 + m]);
        fprintf(stderr, "\ndbg2 resolve-cycles:");
        for (int m = 0; m < 64; ++m) fprintf(stderr, " %llu", d0.dbg2[128 + m]);
        fprintf(stderr, "\n");
        for (int a = 0; a < 4; ++a) {
            fprintf(stderr, "dbg2 %s:", a == 0 ? "resolve>26k" : a == 1 ? "resolve>32k" : a == 2 ? "resolve>40k" : "resolve-max");
            for (int m = 0; m < 64; ++m) fprintf(stderr, " %llu", d0.dbg2[192 + 64 * a + m]);
            fprintf(stderr, "\n");
        }
        for (int k = 0; k < 32; ++k) {
            fprintf(stderr, "ts block %d:", RRT_TS_BASE + k);
            for (int e = 0; e < 16; ++e) fprintf(stderr, " %lld", d0.ts[k * 16 + e] ? (long long)(d0.ts[k * 16 + e] - d0.ts[0]) : -1ll);
            fprintf(stderr, "\n");
        }
    }
#endif
    // A team whose members were not resident together stops at a block boundary with a consistent tree (ST_TEAM_FAIL, a
    // bounded wait expired).  Teams are only an optimisation: the batch continues from there with one CU per query.
    // (Only such launches are continued: the Dubins pipeline's stall exit also says ST_TEAM_FAIL, and there the status stays so that
    // rrt_batch_get_result reports RRT_E_HIP instead of "has not run".)
    bool team_fail = false;
    const bool can_continue = b->use_block && (b->team > 1 || b->last_pipe1);
    for (auto &d : b->h_desc)
        if (d.status == ST_TEAM_FAIL && can_continue) {
            d.status = ST_RUNNING;
            team_fail = true;
        }
    if (team_fail) {
        float ms0 = 0.f;
        (void)hipEventElapsedTime(&ms0, b->ev0, b->ev1);  // the launch that timed out counts in rrt_batch_elapsed_ms
        b->team_fallbacks += 1;
        b->one_cu_once = true;
        b->ms_before += ms0;
        HIPCHK(ctx, hipMemcpyAsync(b->d_desc, b->h_desc.data(), (size_t)b->Q * sizeof(QDesc), hipMemcpyHostToDevice, ctx->stream));
        int rc = rrt_batch_launch(b);
        if (rc != RRT_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(b->h_desc.data(), b->d_desc, (size_t)b->Q * sizeof(QDesc), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, wait_stream_spin(ctx->stream));
        cu_release(ctx->device, b->claimed_cus);
        b->claimed_cus = 0;
    }
    return RRT_OK;
}

extern "C" int rrt_batch_team(rrt_batch *b, int32_t *cus_per_query, int32_t *fallbacks) {
    if (!b || !cus_per_query) return fail(nullptr, RRT_E_ARG, "rrt_batch_team: NULL");
    *cus_per_query = b->use_block ? b->team : 1;
    if (fallbacks) *fallbacks = b->team_fallbacks;
    return RRT_OK;
}

extern "C" int rrt_batch_team_info(rrt_batch *b, int32_t out[4]) {
    if (!b || !out) return fail(nullptr, RRT_E_ARG, "rrt_batch_team_info: NULL");
    out[0] = b->use_block ? b->team : 1;
    out[1] = b->use_block ? (b->last_team > 0 ? b->last_team : b->team) : 1;
    out[2] = b->team_fallbacks;
    out[3] = b->shrunk;
    return RRT_OK;
}

extern "C" int rrt_batch_pipelined(rrt_batch *b, int32_t *pipelined) {
    if (!b || !pipelined) return fail(nullptr, RRT_E_ARG, "rrt_batch_pipelined: NULL");
    *pipelined = (b->use_block && b->team > 1 && b->pipe) ? 1 : 0;
    return RRT_OK;
}

extern "C" int rrt_batch_kernel_name(rrt_batch *b, char *buf, int32_t len) {
    if (!b || !buf || len < 1) return fail(nullptr, RRT_E_ARG, "rrt_batch_kernel_name: bad argument");
    char tmp[160];
    if (b->use_block) {
        const int team = b->last_team > 0 ? b->last_team : b->team;
        const int bsm = b->last_wide ? 32 : (team <= 4 ? 16 : 64 / team);
        if (b->last_pipe1) snprintf(tmp, sizeof tmp, "rrt_pipe_kernel");
        else snprintf(tmp, sizeof tmp, "rrt_expand_block_kernel<%d, %d, %s, %s>", team, bsm, (team > 1 && b->pipe) ? "true" : "false", b->last_inf ? "true" : "false");
    } else if (b->dub_block) {
        snprintf(tmp, sizeof tmp, "rrt_dubins_block_kernel");
    } else {
        snprintf(tmp, sizeof tmp, "rrt_expand_kernel<%s, %s>", (b->flags & RRT_FLAG_REWIRE) ? "true" : "false", (b->flags & RRT_FLAG_DUBINS) ? "true" : "false");
    }
    snprintf(buf, (size_t)len, "%s", tmp);
    return RRT_OK;
}

extern "C" int rrt_batch_elapsed_ms(rrt_batch *b, float *ms) {
    if (!b || !ms) return fail(nullptr, RRT_E_ARG, "rrt_batch_elapsed_ms: NULL");
    if (!b->timed) return fail(b->ctx, RRT_E_ARG, "rrt_batch_elapsed_ms: nothing launched");
    HIPCHK(b->ctx, hipEventElapsedTime(ms, b->ev0, b->ev1));
    *ms += b->ms_before;
    return RRT_OK;
}

extern "C" int rrt_batch_get_result(rrt_batch *b, int32_t q, rrt_result *out) {
    if (!b || !out) return fail(nullptr, RRT_E_ARG, "rrt_batch_get_result: NULL");
    rrt_ctx *ctx = b->ctx;
    if (q < 0 || q >= b->Q) return fail(ctx, RRT_E_ARG, "rrt_batch_get_result: q=%d of %d", q, b->Q);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const QDesc &d = b->h_desc[(size_t)q];
    if (d.status == ST_RUNNING || d.status == ST_IDLE)
        return fail(ctx, RRT_E_ARG, "rrt_batch_get_result: query %d has not run (launch + sync first)", q);
    out->status = d.status;
    out->j = d.j;
    out->vgoal = d.vgoal;
    out->found = d.found;
    out->i_switch = d.i_switch;
    out->rows = d.found ? d.n + 1 : d.n;
    out->sum_j = (int64_t)d.sum_j;
    out->sum_cells_nn = (int64_t)d.sum_cells_nn;
    out->sum_near = (int64_t)d.sum_near;
    out->sum_cells_cand = (int64_t)d.sum_cells_cand;
    out->n_los_cand = (int64_t)d.n_los_cand;
    out->n_rewired = (int64_t)d.n_rewired;
    out->n_propagated = (int64_t)d.n_propagated;
    out->n_words = (int64_t)d.n_words;
    const int live = d.j + (d.found ? 1 : 0);
    if (out->pts) {
        std::vector<uint32_t> tmp((size_t)live);
        HIPCHK(ctx, hipMemcpyAsync(tmp.data(), b->d_nodes + (size_t)q * b->node_stride, (size_t)live * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < live; ++k) {
            out->pts[2 * k] = (int32_t)(tmp[(size_t)k] & 0xffffu);
            out->pts[2 * k + 1] = (int32_t)(tmp[(size_t)k] >> 16);
        }
    }
    if (out->vcost)
        HIPCHK(ctx, hipMemcpyAsync(out->vcost, b->d_vcost + (size_t)q * b->node_stride, (size_t)live * sizeof(double),
                                   hipMemcpyDeviceToHost, ctx->stream));
    if (out->parent)
        HIPCHK(ctx, hipMemcpyAsync(out->parent, b->d_parent + (size_t)q * b->node_stride, (size_t)live * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, ctx->stream));
    if (out->head && b->d_heading) {
        std::vector<uint8_t> tmp((size_t)live);
        HIPCHK(ctx, hipMemcpyAsync(tmp.data(), b->d_heading + (size_t)q * b->node_stride, (size_t)live, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < live; ++k) out->head[k] = (int32_t)tmp[(size_t)k];
    }
    const int ni = d.i;  // iterations executed so far
    if (b->flags & RRT_FLAG_LOGS) {
        const size_t o = (size_t)q * b->n_cap;
        if (out->nearest_log)
            HIPCHK(ctx, hipMemcpyAsync(out->nearest_log, b->d_nearest_log + o, (size_t)ni * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (out->accept_log)
            HIPCHK(ctx, hipMemcpyAsync(out->accept_log, b->d_accept_log + o, (size_t)ni * sizeof(uint8_t), hipMemcpyDeviceToHost, ctx->stream));
        if (out->cbest_log)
            HIPCHK(ctx, hipMemcpyAsync(out->cbest_log, b->d_cbest_log + o, (size_t)ni * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (out->j_log)
            HIPCHK(ctx, hipMemcpyAsync(out->j_log, b->d_j_log + o, (size_t)ni * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (d.status == ST_TEAM_FAIL)
        return fail(ctx, RRT_E_HIP, b->dub_block ? "query %d: the Dubins pipeline stalled (a bounded wait inside the kernel expired at sample %d); the tree up to there is consistent"
                                                 : "query %d: the workgroups of its team (%d) were not resident together (a hand-off timed out); "
                                                   "use RRT_FLAG_NOTEAM when other kernels share the device", q, b->dub_block ? d.i : b->team);
    return d.status < 0 ? d.status : RRT_OK;
}

extern "C" int rrt_batch_debug_cycles(rrt_batch *b, int32_t q, uint64_t out[38]) {
    if (!b || !out || q < 0 || q >= b->Q) return fail(nullptr, RRT_E_ARG, "rrt_batch_debug_cycles: bad argument");
    for (int k = 0; k < 6; ++k) out[k] = b->h_desc[(size_t)q].cyc[k];
    for (int k = 0; k < 32; ++k) out[6 + k] = b->h_desc[(size_t)q].wcyc[k];
    return RRT_OK;
}

extern "C" int rrt_batch_result_block(rrt_batch *b, void **dev_ptr, int64_t *bytes) {
    if (!b || !dev_ptr || !bytes) return fail(nullptr, RRT_E_ARG, "rrt_batch_result_block: NULL");
    *dev_ptr = b->d_slab;
    *bytes = (int64_t)b->slab_bytes;
    return RRT_OK;
}


// ---- multi-GPU: one process per GPU, query q on rank q mod world, results all-gathered over RCCL ----
// librccl is opened on first use, so a single-GPU process never loads it.
namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::string g_rccl_path;  // rrt_comm_use_library: the library to open instead of librccl.so.1 (empty: the default)

std::mutex g_rccl_mutex;  // contexts of different host threads may ask for the library at the same time

int rccl_load(rrt_ctx *ctx) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return RRT_OK;
    void *h = nullptr;
    if (!g_rccl_path.empty()) {
        h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(ctx, RRT_E_COMM, "cannot open the collective library %s: %s", g_rccl_path.c_str(), dlerror());
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(ctx, RRT_E_COMM, "cannot open librccl.so.1: %s", dlerror());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.AllReduce || !a.GetErrorString) {
        dlclose(h);
        return fail(ctx, RRT_E_COMM, "librccl.so.1 lacks an expected symbol");
    }
    g_rccl = a;
    return RRT_OK;
}

#define RCCLCHK(ctx, call)                                                                                   \
    do {                                                                                                     \
        ncclResult_t r_ = (call);                                                                            \
        if (r_ != ncclSuccess) return fail(ctx, RRT_E_COMM, "%s: %s", #call, g_rccl.GetErrorString(r_));     \
    } while (0)

// {status, j, vgoal, found} of every query into the tail of the slab, so that a gathered slab is self-describing
__global__ void slab_meta_kernel(const QDesc *desc, int Q, int32_t *meta) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < Q) {
        meta[4 * q + 0] = desc[q].status;
        meta[4 * q + 1] = desc[q].j;
        meta[4 * q + 2] = desc[q].vgoal;
        meta[4 * q + 3] = desc[q].found;
    }
}
}  // namespace

extern "C" int rrt_comm_use_library(const char *path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return fail(nullptr, RRT_E_COMM, "rrt_comm_use_library: a collective library is already open in this process");
    g_rccl_path = path ? path : "";
    return RRT_OK;
}

extern "C" int rrt_comm_unique_id(uint8_t id[RRT_COMM_ID_BYTES]) {
    if (!id) return fail(nullptr, RRT_E_ARG, "rrt_comm_unique_id: NULL");
    static_assert(RRT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    int rc = rccl_load(nullptr);
    if (rc != RRT_OK) return rc;
    ncclUniqueId u;
    RCCLCHK(nullptr, g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return RRT_OK;
}

extern "C" int rrt_comm_init(rrt_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[RRT_COMM_ID_BYTES]) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail(ctx, RRT_E_ARG, "rrt_comm_init: bad argument");
    if (ctx->comm) return fail(ctx, RRT_E_ARG, "rrt_comm_init: the context already has a communicator");
    int rc = rccl_load(ctx);
    if (rc != RRT_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    RCCLCHK(ctx, g_rccl.CommInitRank(&ctx->comm, world, u, rank));
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return RRT_OK;
}

extern "C" int rrt_comm_destroy(rrt_ctx *ctx) {
    if (!ctx) return RRT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    if (ctx->gather_buf) (void)hipFree(ctx->gather_buf);
    if (ctx->d_red) (void)hipFree(ctx->d_red);
    ctx->gather_buf = nullptr;
    ctx->d_red = nullptr;
    ctx->gather_bytes = 0;
    ctx->gather_owner = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    return RRT_OK;
}

extern "C" int rrt_comm_allreduce_f64(rrt_ctx *ctx, double *vals, int32_t count, int32_t op) {
    if (!ctx || !vals || count < 1 || count > 64 || op < 0 || op > 2) return fail(ctx, RRT_E_ARG, "rrt_comm_allreduce_f64: bad argument");
    if (!ctx->comm) return fail(ctx, RRT_E_COMM, "rrt_comm_allreduce_f64: call rrt_comm_init first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_red) HIPCHK(ctx, hipMalloc((void **)&ctx->d_red, 64 * sizeof(double)));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_red, vals, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    RCCLCHK(ctx, g_rccl.AllReduce(ctx->d_red, ctx->d_red, (size_t)count, ncclDouble, ops[op], ctx->comm, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(vals, ctx->d_red, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}

extern "C" int rrt_ctx_sync(rrt_ctx *ctx) {
    if (!ctx) return fail(nullptr, RRT_E_ARG, "rrt_ctx_sync: NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipDeviceSynchronize());
    return RRT_OK;
}

extern "C" int rrt_gather(rrt_batch *b, void **gathered_dev, int64_t *bytes_per_rank) {
    if (!b) return fail(nullptr, RRT_E_ARG, "rrt_gather: NULL");
    rrt_ctx *ctx = b->ctx;
    if (!ctx->comm) return fail(ctx, RRT_E_COMM, "rrt_gather: call rrt_comm_init first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t need = b->slab_bytes * (size_t)ctx->comm_world;
    ctx->gather_owner = nullptr;
    {
        // Every rank must bring a slab of the same size (same Q and capacity).  The check is itself a collective, so it runs in
        // EVERY rrt_gather on every rank -- never conditionally on this rank's own cache state, which would let one rank enter
        // the all-reduce while another enters the all-gather.  It costs one 16-byte all-reduce per gather.
        double mm[2] = {(double)b->slab_bytes, -(double)b->slab_bytes};
        int rc = rrt_comm_allreduce_f64(ctx, mm, 2, 1);
        if (rc != RRT_OK) return rc;
        if (mm[0] != -mm[1]) return fail(ctx, RRT_E_COMM, "rrt_gather: ranks hold result slabs of different sizes (%.0f .. %.0f bytes)", -mm[1], mm[0]);
    }
    if (ctx->gather_bytes != need) {
        if (ctx->gather_buf) HIPCHK(ctx, hipFree(ctx->gather_buf));
        ctx->gather_buf = nullptr;
        ctx->gather_bytes = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->gather_buf, need));
        ctx->gather_bytes = need;
    }
    int32_t *meta = reinterpret_cast<int32_t *>(b->d_slab + b->slab_bytes - (size_t)b->Q * 4 * sizeof(int32_t));
    hipLaunchKernelGGL(slab_meta_kernel, dim3((unsigned)((b->Q + 63) / 64)), dim3(64), 0, ctx->stream, b->d_desc, b->Q, meta);
    HIPCHK(ctx, hipGetLastError());
    RCCLCHK(ctx, g_rccl.AllGather(b->d_slab, ctx->gather_buf, b->slab_bytes, ncclUint8, ctx->comm, ctx->stream));
    ctx->gather_owner = b;
    if (gathered_dev) *gathered_dev = ctx->gather_buf;
    if (bytes_per_rank) *bytes_per_rank = (int64_t)b->slab_bytes;
    return RRT_OK;
}

extern "C" int rrt_gather_fetch(rrt_batch *b, int32_t rank, int32_t q, rrt_result *out) {
    if (!b || !out) return fail(nullptr, RRT_E_ARG, "rrt_gather_fetch: NULL");
    rrt_ctx *ctx = b->ctx;
    if (!ctx->gather_buf || ctx->gather_owner != b || ctx->gather_bytes != b->slab_bytes * (size_t)ctx->comm_world)
        return fail(ctx, RRT_E_COMM, "rrt_gather_fetch: the gathered slabs are not this batch's (call rrt_gather on this batch first)");
    if (rank < 0 || rank >= ctx->comm_world || q < 0 || q >= b->Q) return fail(ctx, RRT_E_ARG, "rrt_gather_fetch: rank %d, query %d", rank, q);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const unsigned char *slab = ctx->gather_buf + (size_t)rank * b->slab_bytes;
    const size_t Q = (size_t)b->Q, S = (size_t)b->node_stride;
    int32_t meta[4];
    HIPCHK(ctx, hipMemcpyAsync(meta, slab + Q * S * 16 + (size_t)q * 16, sizeof meta, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    out->status = meta[0];
    out->j = meta[1];
    out->vgoal = meta[2];
    out->found = meta[3];
    const int live = meta[1] + (meta[3] ? 1 : 0);
    if (live < 0 || (size_t)live > S) return fail(ctx, RRT_E_COMM, "rrt_gather_fetch: rank %d query %d carries %d rows", rank, q, live);
    if ((out->pts || out->vcost || out->parent) && live > out->rows)
        return fail(ctx, RRT_E_ARG, "rrt_gather_fetch: rank %d query %d has %d rows, the caller's arrays hold %d (set out->rows to their capacity)", rank, q, live, out->rows);
    out->rows = live;
    if (out->pts) {
        std::vector<uint32_t> tmp((size_t)live);
        HIPCHK(ctx, hipMemcpyAsync(tmp.data(), slab + Q * S * 8 + ((size_t)q * S) * 4, (size_t)live * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < live; ++k) {
            out->pts[2 * k] = (int32_t)(tmp[(size_t)k] & 0xffffu);
            out->pts[2 * k + 1] = (int32_t)(tmp[(size_t)k] >> 16);
        }
    }
    if (out->vcost) HIPCHK(ctx, hipMemcpyAsync(out->vcost, slab + ((size_t)q * S) * 8, (size_t)live * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out->parent)
        HIPCHK(ctx, hipMemcpyAsync(out->parent, slab + Q * S * 12 + ((size_t)q * S) * 4, (size_t)live * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}

// ---- one-shot wrappers -------------------------------------------------------------------

static int ensure_single(rrt_ctx *ctx, int32_t n, uint32_t flags) {
    rrt_batch *s = ctx->single;
    if (s && (s->n_cap < n || s->flags != flags || s->gridW != ctx->W || s->gridH != ctx->H)) {
        rrt_batch_destroy(s);
        s = nullptr;
    }
    if (!s) {
        int rc = rrt_batch_create(ctx, 1, n, flags, &s);
        if (rc != RRT_OK) return rc;
        ctx->single = s;
    }
    return RRT_OK;
}

static int run_single(rrt_ctx *ctx, rrt_result *out) {
    rrt_batch *s = ctx->single;
    int rc = rrt_batch_launch(s);
    if (rc != RRT_OK) return rc;
    rc = rrt_batch_sync(s);
    if (rc != RRT_OK) return rc;
    rc = rrt_batch_get_result(s, 0, out);
    if (rc != RRT_OK) return rc;
    return out->status;
}

extern "C" int rrt_plan(rrt_ctx *ctx, const rrt_query *query, uint32_t flags, rrt_result *out) {
    if (!ctx || !query || !out) return fail(ctx, RRT_E_ARG, "rrt_plan: NULL");
    if (!ctx->og) return fail(ctx, RRT_E_NOGRID, "rrt_plan: call rrt_set_grid first");
    if (query->alg >= RRT_ALG_DUBINS) flags |= RRT_FLAG_DUBINS;
    int rc = ensure_single(ctx, query->n, flags);
    if (rc != RRT_OK) return rc;
    rc = rrt_batch_set_query(ctx->single, 0, query);
    if (rc != RRT_OK) return rc;
    return run_single(ctx, out);
}

extern "C" int rrt_plan_resume(rrt_ctx *ctx, const double *unitball, int32_t count, rrt_result *out) {
    if (!ctx || !unitball || !out) return fail(ctx, RRT_E_ARG, "rrt_plan_resume: NULL");
    rrt_batch *s = ctx->single;
    if (!s || s->h_desc[0].status != ST_NEED_UB) return fail(ctx, RRT_E_ARG, "rrt_plan_resume: no query is waiting");
    int rc = rrt_batch_set_unitball(s, 0, unitball, count, s->h_desc[0].i);
    if (rc != RRT_OK) return rc;
    return run_single(ctx, out);
}

extern "C" int rrt_plan_batch(rrt_ctx *ctx, int32_t Q, const rrt_query *queries, rrt_result *out) {
    if (!ctx || !queries || !out || Q < 1) return fail(ctx, RRT_E_ARG, "rrt_plan_batch: bad argument");
    int32_t n_cap = 0;
    for (int q = 0; q < Q; ++q) n_cap = queries[q].n > n_cap ? queries[q].n : n_cap;
    rrt_batch *b = nullptr;
    int rc = rrt_batch_create(ctx, Q, n_cap, queries[0].alg >= RRT_ALG_DUBINS ? RRT_FLAG_DUBINS : 0u, &b);
    if (rc != RRT_OK) return rc;
    for (int q = 0; q < Q && rc == RRT_OK; ++q) rc = rrt_batch_set_query(b, q, &queries[q]);
    if (rc == RRT_OK) rc = rrt_batch_launch(b);
    if (rc == RRT_OK) rc = rrt_batch_sync(b);
    int worst = RRT_OK;
    for (int q = 0; q < Q && rc == RRT_OK; ++q) {
        int r = rrt_batch_get_result(b, q, &out[q]);
        if (r == RRT_E_HIP || r == RRT_E_ARG) rc = r;
        else if (out[q].status != RRT_OK && worst == RRT_OK) worst = out[q].status;
    }
    rrt_batch_destroy(b);
    return rc != RRT_OK ? rc : worst;
}

// ---- host-driven planners (custom cost functions): the device-resident vertex list and its per-iteration query --------------
struct rrt_tree {
    rrt_ctx *ctx = nullptr;
    int32_t cap = 0, j = 0;
    int32_t W = 0, H = 0;  // the grid the vertices were checked against (first append after create / reset); a query on another shape is refused
    uint32_t *d_nodes = nullptr;
    uint32_t *h_nodes = nullptr;  // page-locked mirror: an append is one 4-byte copy in stream order
    int32_t *d_out = nullptr;     // [2 + cap]
    uint8_t *d_los = nullptr;     // [1 + cap]
    int32_t *h_out = nullptr;     // page-locked [2 + cap]
    uint8_t *h_los = nullptr;     // page-locked [1 + cap]
};

extern "C" int rrt_tree_destroy(rrt_tree *t) {
    if (!t) return RRT_OK;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    if (t->d_nodes) (void)hipFree(t->d_nodes);
    if (t->d_out) (void)hipFree(t->d_out);
    if (t->d_los) (void)hipFree(t->d_los);
    if (t->h_nodes) (void)hipHostFree(t->h_nodes);
    if (t->h_out) (void)hipHostFree(t->h_out);
    if (t->h_los) (void)hipHostFree(t->h_los);
    delete t;
    return RRT_OK;
}

extern "C" int rrt_tree_create(rrt_ctx *ctx, int32_t capacity, rrt_tree **out) {
    if (!ctx || !out || capacity < 1) return fail(ctx, RRT_E_ARG, "rrt_tree_create: bad argument");
    if (capacity > (1 << 24)) return fail(ctx, RRT_E_UNSUPPORTED, "rrt_tree_create: capacity %d exceeds %d vertices", capacity, 1 << 24);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rrt_tree *t = new rrt_tree();
    t->ctx = ctx;
    t->cap = capacity;
    const size_t c = (size_t)capacity;
    hipError_t e = hipMalloc((void **)&t->d_nodes, c * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_out, (c + 2) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_los, c + 1);
    if (e == hipSuccess) e = hipHostMalloc((void **)&t->h_nodes, c * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&t->h_out, (c + 2) * sizeof(int32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&t->h_los, c + 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        rrt_tree_destroy(t);
        return fail(ctx, RRT_E_HIP, "rrt_tree_create: %s", hipGetErrorString(e));
    }
    *out = t;
    return RRT_OK;
}

extern "C" int rrt_tree_reset(rrt_tree *t) {
    if (!t) return fail(nullptr, RRT_E_ARG, "rrt_tree_reset: NULL");
    HIPCHK(t->ctx, hipSetDevice(t->ctx->device));
    HIPCHK(t->ctx, hipStreamSynchronize(t->ctx->stream));  // an append still in flight reads the mirror
    t->j = 0;
    t->W = t->H = 0;
    return RRT_OK;
}

extern "C" int rrt_tree_append(rrt_tree *t, int32_t x, int32_t y, int32_t *index) {
    if (!t) return fail(nullptr, RRT_E_ARG, "rrt_tree_append: NULL");
    rrt_ctx *ctx = t->ctx;
    if (t->j >= t->cap) return fail(ctx, RRT_E_ARG, "rrt_tree_append: the tree holds its %d vertices", t->cap);
    // every stored vertex later starts a line-of-sight walk over the context's grid (tree_query_kernel): it must lie inside THAT grid
    if (!ctx->og) return fail(ctx, RRT_E_NOGRID, "rrt_tree_append: no grid");
    if (t->j == 0) {
        t->W = ctx->W;
        t->H = ctx->H;
    } else if (t->W != ctx->W || t->H != ctx->H) {
        return fail(ctx, RRT_E_ARG, "rrt_tree_append: the grid changed shape (%dx%d -> %dx%d) since the tree's first vertex: rrt_tree_reset first", t->W, t->H, ctx->W, ctx->H);
    }
    if (x < 0 || x >= ctx->W || y < 0 || y >= ctx->H || x >= RRT_GRID_MAX || y >= RRT_GRID_MAX)
        return fail(ctx, RRT_E_ARG, "rrt_tree_append: (%d, %d) outside the %dx%d grid", x, y, ctx->W, ctx->H);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    t->h_nodes[t->j] = ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16);
    HIPCHK(ctx, hipMemcpyAsync(t->d_nodes + t->j, t->h_nodes + t->j, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (index) *index = t->j;
    t->j += 1;
    return RRT_OK;
}

extern "C" int rrt_tree_query(rrt_tree *t, int32_t x, int32_t y, int64_t r2, int32_t *nearest, int32_t *within_count, int32_t *within_idx,
                              uint8_t *los_free, int32_t cap) {
    if (!t || !nearest || !within_count || cap < 0 || (cap > 0 && (!within_idx || !los_free)) || !los_free)
        return fail(t ? t->ctx : nullptr, RRT_E_ARG, "rrt_tree_query: bad argument");
    rrt_ctx *ctx = t->ctx;
    if (!ctx->og) return fail(ctx, RRT_E_NOGRID, "rrt_tree_query: no grid");
    if (t->j < 1) return fail(ctx, RRT_E_ARG, "rrt_tree_query: the tree is empty");
    if (t->W != ctx->W || t->H != ctx->H)
        return fail(ctx, RRT_E_ARG, "rrt_tree_query: the tree's vertices were checked against a %dx%d grid, the context now holds %dx%d", t->W, t->H, ctx->W, ctx->H);
    if (x < 0 || x >= ctx->W || y < 0 || y >= ctx->H) return fail(ctx, RRT_E_ARG, "rrt_tree_query: (%d, %d) outside the %dx%d grid", x, y, ctx->W, ctx->H);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int dcap = cap < t->cap ? cap : t->cap;
    const int64_t capd2 = 0x7fffffff;  // any squared distance of 15-bit coordinates is below 2^31
    const uint32_t r2c = (uint32_t)(r2 < 0 ? 0 : (r2 > capd2 ? capd2 : r2));
    const uint32_t xq = ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16);
    if (ctx->W > RRT_GRID_FAST || ctx->H > RRT_GRID_FAST)
        hipLaunchKernelGGL(tree_query_kernel<true>, dim3(1), dim3(TPB), 0, ctx->stream, ctx->og, ctx->H, t->d_nodes, t->j, xq, r2c, dcap, t->d_out, t->d_los);
    else
        hipLaunchKernelGGL(tree_query_kernel<false>, dim3(1), dim3(TPB), 0, ctx->stream, ctx->og, ctx->H, t->d_nodes, t->j, xq, r2c, dcap, t->d_out, t->d_los);
    HIPCHK(ctx, hipGetLastError());
    // the usual answer (a few dozen rows) comes back in one copy of each array; a longer list in a second pair
    const int first = dcap < 256 ? dcap : 256;
    HIPCHK(ctx, hipMemcpyAsync(t->h_out, t->d_out, (size_t)(2 + first) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(t->h_los, t->d_los, (size_t)(1 + first), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, wait_stream_spin(ctx->stream));
    const int total = t->h_out[1];
    const int have = total < dcap ? total : dcap;
    if (have > first) {
        HIPCHK(ctx, hipMemcpyAsync(t->h_out + 2 + first, t->d_out + 2 + first, (size_t)(have - first) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(t->h_los + 1 + first, t->d_los + 1 + first, (size_t)(have - first), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, wait_stream_spin(ctx->stream));
    }
    *nearest = t->h_out[0];
    *within_count = total;
    los_free[0] = t->h_los[0];
    for (int k = 0; k < have; ++k) {
        within_idx[k] = t->h_out[2 + k];
        los_free[1 + k] = t->h_los[1 + k];
    }
    return RRT_OK;
}

// ---- primitives ---------------------------------------------------------------------------

extern "C" int rrt_prim_collisionfree(rrt_ctx *ctx, const int32_t *ab, int32_t m, uint8_t *out_free, int32_t *out_cells) {
    if (!ctx || !ab || m < 0 || !out_free) return fail(ctx, RRT_E_ARG, "rrt_prim_collisionfree: bad argument");
    if (!ctx->og) return fail(ctx, RRT_E_NOGRID, "rrt_prim_collisionfree: no grid");
    if (m == 0) return RRT_OK;
    for (int k = 0; k < m; ++k)
        if (ab[4 * k] < 0 || ab[4 * k] >= ctx->W || ab[4 * k + 2] < 0 || ab[4 * k + 2] >= ctx->W || ab[4 * k + 1] < 0 ||
            ab[4 * k + 1] >= ctx->H || ab[4 * k + 3] < 0 || ab[4 * k + 3] >= ctx->H)
            return fail(ctx, RRT_E_ARG, "rrt_prim_collisionfree: segment %d outside the grid", k);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevTmp tmp;
    int32_t *d_ab = nullptr, *d_cells = nullptr;
    uint8_t *d_free = nullptr;
    HIPCHK(ctx, tmp.alloc(&d_ab, (size_t)m * 4 * sizeof(int32_t)));
    HIPCHK(ctx, tmp.alloc(&d_cells, (size_t)m * sizeof(int32_t)));
    HIPCHK(ctx, tmp.alloc(&d_free, (size_t)m));
    HIPCHK(ctx, hipMemcpyAsync(d_ab, ab, (size_t)m * 4 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const int waves_per_block = 4;
    if (ctx->W > RRT_GRID_FAST || ctx->H > RRT_GRID_FAST)
        hipLaunchKernelGGL(prim_los_kernel<true>, dim3((unsigned)((m + waves_per_block - 1) / waves_per_block)), dim3(64 * waves_per_block), 0,
                           ctx->stream, ctx->og, ctx->H, d_ab, m, d_free, d_cells);
    else
        hipLaunchKernelGGL(prim_los_kernel<false>, dim3((unsigned)((m + waves_per_block - 1) / waves_per_block)), dim3(64 * waves_per_block), 0,
                           ctx->stream, ctx->og, ctx->H, d_ab, m, d_free, d_cells);
    HIPCHK(ctx, hipMemcpyAsync(out_free, d_free, (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    if (out_cells) HIPCHK(ctx, hipMemcpyAsync(out_cells, d_cells, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}

extern "C" int rrt_prim_nearest_within(rrt_ctx *ctx, const int32_t *pts, int32_t j, const int32_t *xq, int32_t m, int64_t r2,
                                       int32_t *out_nearest, int32_t *out_within_count, int64_t *out_within_idxsum) {
    if (!ctx || !pts || !xq || j < 1 || m < 0 || !out_nearest) return fail(ctx, RRT_E_ARG, "rrt_prim_nearest_within: bad argument");
    if ((long long)j > 64LL * CHUNK) return fail(ctx, RRT_E_UNSUPPORTED, "rrt_prim_nearest_within: j too large");
    if (m == 0) return RRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevTmp tmp;
    std::vector<uint32_t> hp((size_t)((j + 3) & ~3), 0), hq((size_t)m);
    for (int k = 0; k < j; ++k) {
        if (pts[2 * k] < 0 || pts[2 * k] >= 2048 || pts[2 * k + 1] < 0 || pts[2 * k + 1] >= 2048)
            return fail(ctx, RRT_E_ARG, "rrt_prim_nearest_within: node %d outside [0,2048)^2", k);
        hp[(size_t)k] = ((uint32_t)pts[2 * k] & 0xffffu) | ((uint32_t)pts[2 * k + 1] << 16);
    }
    for (int k = 0; k < m; ++k) {
        if (xq[2 * k] < 0 || xq[2 * k] >= 2048 || xq[2 * k + 1] < 0 || xq[2 * k + 1] >= 2048)
            return fail(ctx, RRT_E_ARG, "rrt_prim_nearest_within: query %d outside [0,2048)^2", k);
        hq[(size_t)k] = ((uint32_t)xq[2 * k] & 0xffffu) | ((uint32_t)xq[2 * k + 1] << 16);
    }
    uint32_t *d_p = nullptr, *d_q = nullptr;
    int32_t *d_nn = nullptr, *d_cnt = nullptr;
    unsigned long long *d_sum = nullptr;
    uint2 *d_spill = nullptr;
    HIPCHK(ctx, tmp.alloc(&d_sum, (size_t)m * sizeof(unsigned long long)));
    const int spill_stride = ((j + CHUNK - 1) / CHUNK) * CHUNK;
    HIPCHK(ctx, tmp.alloc(&d_spill, (size_t)m * (size_t)spill_stride * sizeof(uint2)));
    HIPCHK(ctx, tmp.alloc(&d_p, hp.size() * sizeof(uint32_t)));
    HIPCHK(ctx, tmp.alloc(&d_q, hq.size() * sizeof(uint32_t)));
    HIPCHK(ctx, tmp.alloc(&d_nn, (size_t)m * sizeof(int32_t)));
    HIPCHK(ctx, tmp.alloc(&d_cnt, (size_t)m * sizeof(int32_t)));
    HIPCHK(ctx, hipMemcpyAsync(d_p, hp.data(), hp.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_q, hq.data(), hq.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    const int64_t cap = 1 << 24;
    const uint32_t r2c = (uint32_t)(r2 < 0 ? 0 : (r2 > cap ? cap : r2));
    hipLaunchKernelGGL(prim_nn_kernel, dim3((unsigned)m), dim3(TPB), 0, ctx->stream, d_p, j, d_q, r2c, d_nn, d_cnt, d_sum, d_spill, spill_stride);
    HIPCHK(ctx, hipMemcpyAsync(out_nearest, d_nn, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (out_within_count)
        HIPCHK(ctx, hipMemcpyAsync(out_within_count, d_cnt, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (out_within_idxsum)
        HIPCHK(ctx, hipMemcpyAsync(out_within_idxsum, d_sum, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}

extern "C" int rrt_prim_sqrt_u32(rrt_ctx *ctx, uint32_t lo, uint32_t count, double *out) {
    if (!ctx || !out) return fail(ctx, RRT_E_ARG, "rrt_prim_sqrt_u32: NULL");
    if (count == 0) return RRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevTmp tmp;
    double *d = nullptr;
    HIPCHK(ctx, tmp.alloc(&d, (size_t)count * sizeof(double)));
    hipLaunchKernelGGL(prim_sqrt_kernel, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, lo, count, d);
    HIPCHK(ctx, hipMemcpyAsync(out, d, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}

extern "C" int rrt_prim_sqrt_u24(rrt_ctx *ctx, uint32_t lo, uint32_t count, double *out) {
    if (!ctx || !out) return fail(ctx, RRT_E_ARG, "rrt_prim_sqrt_u24: NULL");
    if ((unsigned long long)lo + count > (1ull << 24)) return fail(ctx, RRT_E_ARG, "rrt_prim_sqrt_u24: radicand >= 2^24");
    if (count == 0) return RRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevTmp tmp;
    double *d = nullptr;
    HIPCHK(ctx, tmp.alloc(&d, (size_t)count * sizeof(double)));
    hipLaunchKernelGGL(prim_sqrt_u24_kernel, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, lo, count, d);
    HIPCHK(ctx, hipMemcpyAsync(out, d, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}

extern "C" int rrt_prim_sqrt_f64(rrt_ctx *ctx, const double *in, uint32_t count, double *out) {
    if (!ctx || !in || !out) return fail(ctx, RRT_E_ARG, "rrt_prim_sqrt_f64: NULL");
    if (count == 0) return RRT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevTmp tmp;
    double *d_in = nullptr, *d_out = nullptr;
    HIPCHK(ctx, tmp.alloc(&d_in, (size_t)count * sizeof(double)));
    HIPCHK(ctx, tmp.alloc(&d_out, (size_t)count * sizeof(double)));
    HIPCHK(ctx, hipMemcpyAsync(d_in, in, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(prim_sqrt_f64_kernel, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, d_in, count, d_out);
    HIPCHK(ctx, hipMemcpyAsync(out, d_out, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RRT_OK;
}
