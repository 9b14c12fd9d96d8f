// rrt_device.h -- device-side building blocks shared by the kernels of rrt_engine.hip.
// gfx950 only (wave64, DPP row_bcast, v_pk_sub_i16 / v_dot2c_i32_i16).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rrt_line.h"

namespace rrtdev {

constexpr int TPB = 1024;          // threads of one query workgroup (16 waves, one CU)
constexpr int NWAVE = TPB / 64;
constexpr int CHUNK = TPB * 4;     // nodes per scan chunk: one 16-byte load per thread
constexpr int MAX_LDS_CHUNKS = 8;  // node chunks cached in LDS (8 * 16 KiB = 128 KiB)
constexpr int WSLOTS = 3;          // near-set entries a lane prices in registers
constexpr int WCAP = 64 * WSLOTS;  // near-set entries per wave in LDS (16 waves * 1.5 KiB)
constexpr uint32_t NONE = 0xffffffffu;

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// explicit address spaces: LDS accesses must be ds_* instructions, never flat_* (a pointer
// select between LDS and HBM silently degrades to flat loads with full waits)
#define RRT_LDS __attribute__((address_space(3)))
#define RRT_GLB __attribute__((address_space(1)))

#define RRT_INF_BITS 0x7ff0000000000000ll
__device__ __forceinline__ double f64_inf() { return __longlong_as_double(RRT_INF_BITS); }

// Nodes and samples are packed int16x2: x in the low half, y in the high half.  Grids are
// at most 2048 x 2048 on this path, so x,y < 2^11 and d2 < 2^23.
__device__ __forceinline__ uint32_t pack_xy(int x, int y) { return ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16); }
__device__ __forceinline__ int ux(uint32_t p) { return (int)(p & 0xffffu); }
__device__ __forceinline__ int uy(uint32_t p) { return (int)(p >> 16); }

// Squared distance of two packed points: v_pk_sub_i16 + v_dot2c_i32_i16.
__device__ __forceinline__ uint32_t dist2(uint32_t a, uint32_t b) {
    short2_t d = __builtin_bit_cast(short2_t, a) - __builtin_bit_cast(short2_t, b);
    return (uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false);
}

// Wave-wide unsigned minimum, result uniform in every lane.  DPP row_shr 1/2/4/8 leaves each
// row's minimum in its lane 15; row_bcast:15 / row_bcast:31 fold the four rows into lane 63.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// `old` operand of a DPP step whose result goes into a signed max: the identity of that max (a lane without a source keeps it), which is
// what lets the compiler fold the step into ONE v_max_i32_dpp -- with -1 ("no cell") it was a move of the constant, a DPP move and the max.
constexpr int DPP_SMAX_ID = (int)0x80000000;

// Wave-wide unsigned sum, result uniform in every lane (same DPP ladder, additive).
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Wave-wide minimum of non-negative floats (their bit patterns order like unsigned integers).
__device__ __forceinline__ float wave_min_f32_nonneg(float v) { return __uint_as_float(wave_min_u32(__float_as_uint(v))); }

// Lexicographic wave minimum of (key, idx): smallest key, lowest idx among equal keys.
__device__ __forceinline__ void wave_min_key_idx(uint32_t &key, uint32_t &idx) {
    uint32_t mk = wave_min_u32(key);
    uint32_t mi = wave_min_u32(key == mk ? idx : NONE);
    key = mk;
    idx = mi;
}

// Lexicographic wave minimum of (c, idx) for non-negative doubles (their bit patterns order
// like unsigned integers); +inf / NONE means "nothing".
__device__ __forceinline__ void wave_min_f64_idx(double &c, uint32_t &idx) {
    unsigned long long b = (unsigned long long)__double_as_longlong(c);
    uint32_t hi = (uint32_t)(b >> 32), lo = (uint32_t)b;
    uint32_t mh = wave_min_u32(hi);
    uint32_t ml = wave_min_u32(hi == mh ? lo : NONE);
    uint32_t mi = wave_min_u32((hi == mh && lo == ml) ? idx : NONE);
    c = __longlong_as_double((long long)(((unsigned long long)mh << 32) | ml));
    idx = mi;
}

// (c1,i1) < (c2,i2) in (cost, index) order
__device__ __forceinline__ bool key_lt(double c1, uint32_t i1, double c2, uint32_t i2) {
    return c1 < c2 || (c1 == c2 && i1 < i2);
}

// r2norm of an integer difference (rrt.py:24): sqrt of an exact integer < 2^53.
__device__ __forceinline__ double sqrt_u32(uint32_t d2) { return sqrt((double)d2); }

// A SHORT segment a -> b (fewer than 64 steps) for one cell per lane: the closed form of include/rrt_line.h with the segment's data in
// scalar registers (a and b are wave-uniform), the cell as one byte offset into the grid, and the quotient without the fix-ups:
//     m_k = floor((2 minor k + major) / (2 major));  the quotient's fractional part is a multiple of 1 / (2 major), so biased by
//     half of that step it stays 1 / (4 major) >= 1/252 clear of the integers on both sides, and
//         m_k = (int)(float(num) * rcp + 0.5 rcp)
//     is exact for any rcp within a few ulp of 1 / (2 major) (num <= 8001; checked exhaustively for major <= 63, every minor and k,
//     rcp off by +-3 ulp, fused or not: tools/check_short_line.py) -- seven vector instructions a cell instead of eighteen, which is
//     what a worker group that tests hundreds of candidates of one sample is bound by.
struct ShortLine {
    int base, smaj, smin, major, minor;
    float rcp, half;
};
__device__ __forceinline__ ShortLine short_line(uint32_t a, uint32_t b, int H) {
    const int x0 = ux(a), y0 = uy(a), x1 = ux(b), y1 = uy(b);
    const int dx = x1 - x0, dy = y1 - y0, adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    const int stx = x0 < x1 ? H : -H, sty = y0 < y1 ? 1 : -1;  // (the steps of rrt_line_setup, as byte strides of the row-major grid)
    const bool xm = adx >= ady;
    ShortLine l;
    l.base = x0 * H + y0;
    l.smaj = xm ? stx : sty;
    l.smin = xm ? sty : stx;
    l.major = xm ? adx : ady;
    l.minor = xm ? ady : adx;
    l.rcp = __builtin_amdgcn_rcpf((float)(l.major > 0 ? 2 * l.major : 1));
    l.half = 0.5f * l.rcp;
    return l;
}
__device__ __forceinline__ uint32_t short_line_cell(const ShortLine &l, int k) {
    const int num = 2 * l.minor * k + l.major;
    const int m = (int)__builtin_fmaf((float)num, l.rcp, l.half);
    return (uint32_t)(l.base + l.smaj * k + l.smin * m);
}

// Line of sight a -> b (RRT.collisionfree, rrt.py:202-229) evaluated by one wavefront:
// lane l tests cell k0+l of the closed-form walk (rrt_line.h); the ballot gives any-hit and
// the first blocked cell.  `cells` = grid cells the reference's serial walk reads
// before returning (first blocked cell + 1, or L+1).  Result uniform.  Long segments keep four
// 64-cell groups of loads in flight.
// WIDE: grids beyond 2048 x 2048 (host-driven planners only): the cell of the walk by 64-bit division (rrt_line_cell_wide).
template <bool WIDE = false>
__device__ __forceinline__ bool los_wave(const uint8_t *__restrict__ og, int H, uint32_t a, uint32_t b, int lane,
                                         int &cells) {
    rrt_line_t l = rrt_line_setup(ux(a), uy(a), ux(b), uy(b));
    const int L = l.major;
    if (WIDE) {
        for (int k0 = 0; k0 <= L; k0 += 64) {
            bool occ = false;
            const int k = k0 + lane;
            if (k <= L) {
                int x, y;
                rrt_line_cell_wide(&l, k, &x, &y);
                occ = og[(uint32_t)x * (uint32_t)H + (uint32_t)y] != 0;
            }
            const unsigned long long m = __ballot(occ);
            if (m) {
                cells = k0 + (int)__builtin_ctzll(m) + 1;
                return false;
            }
        }
        cells = L + 1;
        return true;
    }
    if (L < 64) {
        bool occ = false;
        if (lane <= L) occ = og[short_line_cell(short_line(a, b, H), lane)] != 0;
        unsigned long long m = __ballot(occ);
        cells = m ? (int)__builtin_ctzll(m) + 1 : L + 1;
        return m == 0;
    }
    for (int k0 = 0; k0 <= L; k0 += 256) {
        uint8_t v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k = k0 + 64 * g + lane;
            if (k <= L) {
                int x, y;
                rrt_line_cell(&l, k, &x, &y);
                v[g] = og[(uint32_t)(x * H + y)];
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned long long m = __ballot(v[g] != 0);
            if (m) {
                cells = k0 + 64 * g + (int)__builtin_ctzll(m) + 1;
                return false;
            }
        }
    }
    cells = L + 1;
    return true;
}

// The same test in two halves for segments shorter than 64 cells: los_issue starts the cell loads, los_finish turns them into
// the answer.  A lone wave puts independent work (its share of the near-set stream) between the two so that the memory round
// trip of the test is not on its critical path.  Longer segments are tested in los_finish.
struct LosPending {
    int major;
    uint8_t v;
};
__device__ __forceinline__ LosPending los_issue(const uint8_t *__restrict__ og, int H, uint32_t a, uint32_t b, int lane) {
    const ShortLine l = short_line(a, b, H);  // (major and minor as in rrt_line_setup for any length; the cell only below 64 steps)
    LosPending p;
    p.major = l.major;
    p.v = 0;
    if (l.major < 64 && lane <= l.major) p.v = og[short_line_cell(l, lane)];
    return p;
}
__device__ __forceinline__ bool los_finish(const LosPending &p, const uint8_t *__restrict__ og, int H, uint32_t a, uint32_t b, int lane,
                                           int &cells) {
    if (p.major >= 64) return los_wave(og, H, a, b, lane, cells);
    const unsigned long long m = __ballot(p.v != 0);
    cells = m ? (int)__builtin_ctzll(m) + 1 : p.major + 1;
    return m == 0;
}

// Two lines of sight a0 -> b and a1 -> b at once (the second only if `has1`): for segments shorter than 64 cells both
// cell loads are in flight together, so a failed first test does not cost a second memory round trip.
__device__ __forceinline__ void los_wave2(const uint8_t *__restrict__ og, int H, uint32_t a0, uint32_t a1, bool has1, uint32_t b, int lane,
                                          bool &ok0, int &cells0, bool &ok1, int &cells1) {
    const ShortLine l0 = short_line(a0, b, H), l1 = short_line(a1, b, H);
    ok1 = false;
    cells1 = 0;
    if (l0.major < 64 && (!has1 || l1.major < 64)) {
        uint8_t v0 = 0, v1 = 0;
        if (lane <= l0.major) v0 = og[short_line_cell(l0, lane)];
        if (has1 && lane <= l1.major) v1 = og[short_line_cell(l1, lane)];
        const unsigned long long m0 = __ballot(v0 != 0), m1 = __ballot(v1 != 0);
        ok0 = m0 == 0;
        cells0 = m0 ? (int)__builtin_ctzll(m0) + 1 : l0.major + 1;
        if (has1) {
            ok1 = m1 == 0;
            cells1 = m1 ? (int)__builtin_ctzll(m1) + 1 : l1.major + 1;
        }
        return;
    }
    ok0 = los_wave(og, H, a0, b, lane, cells0);
    if (has1) ok1 = los_wave(og, H, a1, b, lane, cells1);
}

// Up to LOSB lines of sight a[c] -> b at once (c < nc, wave-uniform operands), every segment shorter than 64 cells
// (one ballot each): all cell loads are in flight together.  cells[c] as in los_wave.
constexpr int LOSB = 8;
template <int N>
__device__ __forceinline__ void los_batch_n(const uint8_t *__restrict__ og, int H, const uint32_t (&a)[N], int nc, uint32_t b, int lane,
                                            bool (&ok)[N], int (&cells)[N]) {
    int major[N];
    uint8_t v[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const ShortLine l = short_line(a[c], b, H);
        major[c] = l.major;
        v[c] = 0;
        if (c < nc && lane <= l.major) v[c] = og[short_line_cell(l, lane)];
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const unsigned long long mb = __ballot(v[c] != 0);
        ok[c] = mb == 0;
        cells[c] = mb ? (int)__builtin_ctzll(mb) + 1 : major[c] + 1;
    }
}
__device__ __forceinline__ void los_batch(const uint8_t *__restrict__ og, int H, const uint32_t (&a)[LOSB], int nc, uint32_t b, int lane,
                                          bool (&ok)[LOSB], int (&cells)[LOSB]) {
    los_batch_n<LOSB>(og, H, a, nc, b, lane, ok, cells);
}

}  // namespace rrtdev
