// rrt_prims.h -- the path's primitives as stand-alone kernels (parity tests, micro-benchmarks).
// They call the same device functions as rrt_expand_kernel.
#pragma once

#include "rrt_block.h"
#include "rrt_kernels.h"

namespace rrtdev {

// RRT.collisionfree (rrt.py:183-229): one wavefront per segment.  WIDE: grids beyond 2048 x 2048 (see los_wave).
template <bool WIDE>
__global__ void prim_los_kernel(const uint8_t *og, int H, const int32_t *ab, int m, uint8_t *out_free, int32_t *out_cells) {
    const int lane = (int)(threadIdx.x & 63);
    const int seg = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (seg >= m) return;  // whole wave
    const uint32_t a = pack_xy(ab[4 * seg], ab[4 * seg + 1]), b = pack_xy(ab[4 * seg + 2], ab[4 * seg + 3]);
    int cells = 0;
    const bool ok = los_wave<WIDE>(og, H, a, b, lane, cells);
    if (lane == 0) {
        out_free[seg] = (uint8_t)ok;
        out_cells[seg] = cells;
    }
}

// near()[0] and within() (rrt.py:150-155, :176-181) of one query point per workgroup: the scan
// of rrt_expand_kernel phase A/B over nodes in HBM, with the same per-wave near-set lists.
__global__ __launch_bounds__(TPB) void prim_nn_kernel(const uint32_t *nodes, int j, const uint32_t *queries, uint32_t r2,
                                                      int32_t *out_nearest, int32_t *out_count, unsigned long long *out_idxsum,
                                                      uint2 *spill_all, int spill_stride) {
    __shared__ __attribute__((aligned(16))) u32x2 wlists[NWAVE * WCAP];
    __shared__ uint4 nnslots[NWAVE];
    __shared__ unsigned long long idxsum;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t xq = queries[blockIdx.x];
    uint2 *spill = spill_all + (size_t)blockIdx.x * (size_t)spill_stride;
    const WaveList wl{(RRT_LDS u32x2 *)wlists + wave * WCAP,
                      reinterpret_cast<u32x2 *>(spill) + (size_t)wave * (size_t)(spill_stride / NWAVE)};
    if (t == 0) idxsum = 0;
    __syncthreads();
    uint32_t best = NONE, wcnt = 0;
    const int nfull = j / CHUNK;
    for (int c = 0; c < nfull; ++c) {
        u32x4 v = reinterpret_cast<const u32x4 *>(nodes)[c * TPB + t];
        eval4<true>(v, xq, (uint32_t)c << 2, (uint32_t)(c * CHUNK + 4 * t), r2, best, wl, wcnt, lane);
    }
    {
        const int c = nfull, idx0 = c * CHUNK + 4 * t;
        uint32_t dd[4] = {NONE, NONE, NONE, NONE};
        if (idx0 < j) {
            u32x4 v = reinterpret_cast<const u32x4 *>(nodes)[c * TPB + t];
            uint32_t pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (idx0 + e < j) {
                    dd[e] = dist2(pv[e], xq);
                    best = min(best, (dd[e] << 8) + ((uint32_t)c << 2) + (uint32_t)e);
                }
            }
        }
        wl_append4(wl, wcnt, dd[0] < r2, dd[1] < r2, dd[2] < r2, dd[3] < r2, (uint32_t)idx0, dd[0], dd[1], dd[2], dd[3], lane);
    }
    uint32_t kd = best >> 8, tag = best & 0xffu;
    uint32_t ki = (best == NONE) ? NONE : (tag >> 2) * (uint32_t)CHUNK + 4u * (uint32_t)t + (tag & 3u);
    wave_min_key_idx(kd, ki);
    if (lane == 0) nnslots[wave] = make_uint4(kd, ki, wcnt, 0u);
    unsigned long long s = 0;
    for (uint32_t c = (uint32_t)lane; c < wcnt; c += 64) s += (c < (uint32_t)WCAP) ? wl.list[c].x : wl.spill[c - WCAP].x;
    if (s) atomicAdd(&idxsum, s);
    __syncthreads();
    uint32_t d2n = NONE, vn = NONE, m = 0;
    if (lane < NWAVE) {
        d2n = nnslots[lane].x;
        vn = nnslots[lane].y;
        m = nnslots[lane].z;
    }
    wave_min_key_idx(d2n, vn);
    m = wave_sum_u32(m);
    if (t == 0) {
        out_nearest[blockIdx.x] = (int32_t)vn;
        out_count[blockIdx.x] = (int32_t)m;
        out_idxsum[blockIdx.x] = idxsum;
    }
}

// One iteration's worth of primitives for a HOST-DRIVEN planner (a custom Python cost function keeps the loop of
// rrt.py:498-548 on the host; the device answers what the loop asks of the tree and the grid):
//   out[0]      near()[0]   (rrt.py:150-155, :503): nearest vertex of the j live rows, lowest index among equal distance
//   out[1]      |within()|  (rrt.py:176-181, :513): live rows with d2 < r2
//   out[2 + k]  their indices, ascending, the first `cap` of them
//   los[0]      collisionfree(nearest -> x)        (rrt.py:506)
//   los[1 + k]  collisionfree(within[k] -> x)      (rrt.py:519, :537: the same direction in both loops)
// One workgroup; thread t owns the contiguous rows [t * per, (t + 1) * per), so that its hits are ascending and the
// hits of the threads concatenate in ascending order (exclusive prefix sum of the per-thread counts).
template <bool WIDE>
__global__ __launch_bounds__(TPB) void tree_query_kernel(const uint8_t *og, int H, const uint32_t *nodes, int j, uint32_t xq, uint32_t r2, int cap,
                                                         int32_t *out, uint8_t *los) {
    __shared__ uint32_t wtot[NWAVE];
    __shared__ uint2 wnn[NWAVE];
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (j + TPB - 1) / TPB;
    const int k0 = t * per, k1 = (k0 + per < j) ? k0 + per : j;
    uint32_t bd = NONE, bi = NONE, cnt = 0;
    for (int k = k0; k < k1; ++k) {
        const uint32_t d2 = dist2(nodes[k], xq);
        if (d2 < bd) {  // ascending k: strict < keeps the lowest index among equals
            bd = d2;
            bi = (uint32_t)k;
        }
        cnt += d2 < r2 ? 1u : 0u;
    }
    wave_min_key_idx(bd, bi);
    uint32_t incl = cnt;  // inclusive prefix over the wave's lanes
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
    if (lane == 63) wtot[wave] = incl;
    if (lane == 0) wnn[wave] = make_uint2(bd, bi);
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < NWAVE; ++w) {
        base += w < wave ? wtot[w] : 0u;
        total += wtot[w];
    }
    uint32_t nd = NONE, ni = NONE;
    if (lane < NWAVE) {
        nd = wnn[lane].x;
        ni = wnn[lane].y;
    }
    wave_min_key_idx(nd, ni);
    uint32_t pos = base + incl - cnt;
    for (int k = k0; k < k1; ++k)
        if (dist2(nodes[k], xq) < r2) {
            if (pos < (uint32_t)cap) out[2 + pos] = k;
            ++pos;
        }
    if (t == 0) {
        out[0] = (int32_t)ni;
        out[1] = (int32_t)total;
    }
    __syncthreads();  // the index list is complete (and visible to this workgroup)
    const int nseg = 1 + (int)(total < (uint32_t)cap ? total : (uint32_t)cap);
    for (int s = wave; s < nseg; s += NWAVE) {
        const uint32_t a = nodes[s == 0 ? (int)ni : out[2 + s - 1]];
        int cells = 0;
        const bool ok = los_wave<WIDE>(og, H, a, xq, lane, cells);
        if (lane == 0) los[s] = (uint8_t)ok;
    }
}

// r2norm on an integer radicand (rrt.py:24) exactly as the kernels evaluate it.
__global__ void prim_sqrt_kernel(uint32_t lo, uint32_t count, double *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = sqrt_u32(lo + k);
}

// the block kernel's short sqrt for radicands below 2^24
__global__ void prim_sqrt_u24_kernel(uint32_t lo, uint32_t count, double *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = sqrt_u24(lo + k);
}

// sqrt of arbitrary doubles (the ellipse minor axis, rrt.py:622).
__global__ void prim_sqrt_f64_kernel(const double *in, uint32_t count, double *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) out[k] = sqrt(in[k]);
}

}  // namespace rrtdev
