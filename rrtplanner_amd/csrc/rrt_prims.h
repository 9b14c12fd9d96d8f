// rrt_prims.h -- the path's primitives as stand-alone kernels (parity tests, micro-benchmarks).
// They call the same device functions as rrt_expand_kernel.
#pragma once

#include "rrt_kernels.h"

namespace rrtdev {

// RRT.collisionfree (rrt.py:183-229): one wavefront per segment.
__global__ void prim_los_kernel(const uint8_t *og, int H, const int32_t *ab, int m, uint8_t *out_free, int32_t *out_cells) {
    const int lane = (int)(threadIdx.x & 63);
    const int seg = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (seg >= m) return;  // whole wave
    const uint32_t a = pack_xy(ab[4 * seg], ab[4 * seg + 1]), b = pack_xy(ab[4 * seg + 2], ab[4 * seg + 3]);
    int cells = 0;
    const bool ok = los_wave(og, H, a, b, lane, cells);
    if (lane == 0) {
        out_free[seg] = (uint8_t)ok;
        out_cells[seg] = cells;
    }
}

// near()[0] and within() (rrt.py:150-155, :176-181) of one query point per workgroup: the scan
// of rrt_expand_kernel phase A/B over nodes in HBM.
__global__ __launch_bounds__(TPB) void prim_nn_kernel(const uint32_t *nodes, int j, const uint32_t *queries, uint32_t r2,
                                                      int32_t *out_nearest, int32_t *out_count, unsigned long long *out_idxsum,
                                                      uint2 *spill_all) {
    __shared__ uint2 cand[CANDCAP];
    __shared__ uint32_t cnt;
    __shared__ uint2 nnslots[NWAVE];
    __shared__ unsigned long long idxsum;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t xq = queries[blockIdx.x];
    uint2 *spill = spill_all + (size_t)blockIdx.x * (size_t)j;
    if (t == 0) {
        cnt = 0;
        idxsum = 0;
    }
    __syncthreads();
    NearList nl{cand, &cnt, spill};
    uint32_t best = NONE;
    const int nfull = j / CHUNK;
    for (int c = 0; c < nfull; ++c) {
        uint4 v = reinterpret_cast<const uint4 *>(nodes)[c * TPB + t];
        eval4<true>(v, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, nl, lane);
    }
    {
        const int c = nfull, idx0 = c * CHUNK + 4 * t;
        if (idx0 < j) {
            uint4 v = reinterpret_cast<const uint4 *>(nodes)[c * TPB + t];
            uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (idx0 + e < j) {
                    uint32_t d = dist2(pv[e], xq);
                    best = min(best, (d << 8) + ((uint32_t)c << 2) + (uint32_t)e);
                    if (d < r2) {
                        uint32_t pos = atomicAdd(nl.count, 1u);
                        if (pos < (uint32_t)CANDCAP)
                            nl.list[pos] = make_uint2((uint32_t)(idx0 + e), d);
                        else
                            nl.spill[pos - CANDCAP] = make_uint2((uint32_t)(idx0 + e), d);
                    }
                }
            }
        }
    }
    uint32_t kd = best >> 8, tag = best & 0xffu;
    uint32_t ki = (best == NONE) ? NONE : (tag >> 2) * (uint32_t)CHUNK + 4u * (uint32_t)t + (tag & 3u);
    wave_min_key_idx(kd, ki);
    if (lane == 0) nnslots[wave] = make_uint2(kd, ki);
    __syncthreads();
    uint32_t d2n = NONE, vn = NONE;
    if (lane < NWAVE) {
        d2n = nnslots[lane].x;
        vn = nnslots[lane].y;
    }
    wave_min_key_idx(d2n, vn);
    const uint32_t m = cnt;
    unsigned long long s = 0;
    for (uint32_t c = (uint32_t)t; c < m; c += TPB) {
        const uint2 e = (c < (uint32_t)CANDCAP) ? cand[c] : spill[c - CANDCAP];
        s += e.x;
    }
    if (s) atomicAdd(&idxsum, s);
    __syncthreads();
    if (t == 0) {
        out_nearest[blockIdx.x] = (int32_t)vn;
        out_count[blockIdx.x] = (int32_t)m;
        out_idxsum[blockIdx.x] = idxsum;
    }
}

// r2norm on an integer radicand (rrt.py:24) exactly as the kernels evaluate it.
__global__ void prim_sqrt_kernel(uint32_t lo, uint32_t count, double *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = sqrt_u32(lo + k);
}

// sqrt of arbitrary doubles (the ellipse minor axis, rrt.py:622).
__global__ void prim_sqrt_f64_kernel(const double *in, uint32_t count, double *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = sqrt(in[k]);
}

}  // namespace rrtdev
