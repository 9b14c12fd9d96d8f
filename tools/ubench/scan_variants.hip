// Micro-benchmark: per-node cost of the nearest-neighbour scan inner loop on gfx950 (LDS-resident nodes).
// hipcc --offload-arch=gfx950 -O3 -o scan_variants scan_variants.hip && ./scan_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short short2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDSP __attribute__((address_space(3)))

__device__ __forceinline__ uint32_t d2_dot(uint32_t a, uint32_t b) {
    short2_t d = __builtin_bit_cast(short2_t, a) - __builtin_bit_cast(short2_t, b);
    return (uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false);
}
__device__ __forceinline__ uint32_t d2_mad(uint32_t a, uint32_t b) {
    int dx = (int)(a & 0xffff) - (int)(b & 0xffff), dy = (int)(a >> 16) - (int)(b >> 16);
    return (uint32_t)(__mul24(dx, dx) + __mul24(dy, dy));
}
__device__ __forceinline__ uint32_t d2_f32(uint32_t a, float qx, float qy) {
    float dx = (float)(a & 0xffff) - qx, dy = (float)(a >> 16) - qy;
    return (uint32_t)(dx * dx + dy * dy);
}

template <int VAR>
__global__ __launch_bounds__(1024) void k(const uint32_t* nodes, int nchunks, int reps, uint32_t q0, uint32_t* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDSP uint32_t* l = (LDSP uint32_t*)smem;
    const LDSP u32x4* l4 = (const LDSP u32x4*)smem;
    int t = threadIdx.x;
    for (int k = t; k < nchunks * 4096; k += 1024) l[k] = nodes[k];
    __syncthreads();
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        uint32_t q = q0 + r * 0x00030005u;
        q &= 0x03ff03ffu;
        float qx = (float)(q & 0xffff), qy = (float)(q >> 16);
        uint32_t best = 0xffffffffu;
        u32x4 cur = l4[t];
        for (int c = 0; c < nchunks; ++c) {
            u32x4 nxt = cur;
            if (c + 1 < nchunks) nxt = l4[(c + 1) * 1024 + t];
            uint32_t d0, d1, d2, d3;
            if (VAR == 0) { d0 = d2_dot(cur.x, q); d1 = d2_dot(cur.y, q); d2 = d2_dot(cur.z, q); d3 = d2_dot(cur.w, q); }
            else if (VAR == 1) { d0 = d2_mad(cur.x, q); d1 = d2_mad(cur.y, q); d2 = d2_mad(cur.z, q); d3 = d2_mad(cur.w, q); }
            else { d0 = d2_f32(cur.x, qx, qy); d1 = d2_f32(cur.y, qx, qy); d2 = d2_f32(cur.z, qx, qy); d3 = d2_f32(cur.w, qx, qy); }
            uint32_t tag = c << 2;
            best = min(best, (d0 << 8) + tag);
            best = min(best, (d1 << 8) + tag + 1);
            best = min(best, (d2 << 8) + tag + 2);
            best = min(best, (d3 << 8) + tag + 3);
            cur = nxt;
        }
        acc ^= best;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + t] = acc;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int nchunks = 6, reps = 2000;
    std::vector<uint32_t> h(nchunks * 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((uint32_t)(i * 2654435761u) & 0x03ff) | ((((uint32_t)(i * 40503u) >> 3) & 0x03ff) << 16);
    uint32_t *d, *out; unsigned long long* cyc;
    hipMalloc(&d, h.size() * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    size_t lds = nchunks * 16384;
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int blocks : {1, 256}) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, d, nchunks, reps, 0x00110022u, out, cyc);
            hipDeviceSynchronize();
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            double per_scan = (double)c / reps;
            printf("%-8s blocks=%3d: %8.1f cyc per scan of %d nodes = %.3f cyc/node/CU (%.2f nodes/cyc/CU)\n", name, blocks, per_scan,
                   nchunks * 4096, per_scan / (nchunks * 4096), nchunks * 4096 / per_scan);
        }
    };
    run(k<0>, "dot2");
    run(k<1>, "mad24");
    run(k<2>, "f32");
    return 0;
}
