// rrt_dubins_dev.h -- device side of the Dubins planners (BASELINE.json configs[4]; no reference counterpart, see
// include/rrt_dubins.h): the shortest word between two poses and the wave-wide collision sweep of a word.
#pragma once

#include "rrt_device.h"
#include "rrt_dubins.h"

namespace rrtdev {

struct DubCfg {
    double rho;  // turning radius, cells
    int nh;      // discrete headings
    int W, H;
    // optional (LDS, may be null): angle, sine and cosine of every discrete heading, [3][256] -- exactly dub_heading(h, nh) and
    // dub_sincos of it, evaluated once per launch instead of a division per pose and a series per sweep
    const RRT_LDS double *htab = nullptr;
};

__device__ __forceinline__ double dub_heading_dev(int h, const DubCfg &c) { return c.htab ? c.htab[h] : dub_heading(h, c.nh); }

__device__ __forceinline__ dub_path_t dub_between_dev(uint32_t a, int ha, uint32_t b, int hb, const DubCfg &c) {
    if (c.htab)
        return dub_shortest_sc((double)ux(a), (double)uy(a), c.htab[ha], c.htab[256 + ha], c.htab[512 + ha], (double)ux(b), (double)uy(b), c.htab[hb],
                               c.htab[256 + hb], c.htab[512 + hb], c.rho);
    return dub_shortest((double)ux(a), (double)uy(a), dub_heading(ha, c.nh), (double)ux(b), (double)uy(b), dub_heading(hb, c.nh), c.rho);
}

// Sweep of `path` from pose (a, ha) to cell b by one wavefront: lane l evaluates sample k0 + l of the arc-length grid
// (rrt_dubins.h: k * DUB_DS), the ballot gives any hit and the first blocked sample; a sample outside the grid blocks.  Two
// groups of 64 samples are in flight per round.  `cells` = samples the serial sweep reads before it returns.  Result uniform.
__device__ __forceinline__ bool dub_sweep_wave(const uint8_t *__restrict__ og, const DubCfg &c, uint32_t a, int ha, uint32_t b, const dub_path_t &path,
                                               int lane, int &cells) {
    if (path.word == DUB_NONE) {
        cells = 0;
        return false;
    }
    const dub_sweep_t s = c.htab ? dub_sweep_setup_sc((double)ux(a), (double)uy(a), c.htab[ha], c.htab[256 + ha], c.htab[512 + ha], &path, c.rho)
                                 : dub_sweep_setup((double)ux(a), (double)uy(a), dub_heading(ha, c.nh), &path, c.rho);
    for (int k0 = 0; k0 < s.nsamples; k0 += 128) {
        bool occ[2] = {false, false};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int k = k0 + 64 * g + lane;
            if (k < s.nsamples) {
                int32_t cx, cy;
                dub_sweep_cell(&s, k, &cx, &cy);
                occ[g] = (cx < 0 || cx >= c.W || cy < 0 || cy >= c.H) ? true : og[(uint32_t)(cx * c.H + cy)] != 0;
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const unsigned long long m = __ballot(occ[g]);
            if (m) {
                cells = k0 + 64 * g + (int)__builtin_ctzll(m) + 1;
                return false;
            }
        }
    }
    cells = s.nsamples + 1;
    return og[(uint32_t)(ux(b) * c.H + uy(b))] == 0;  // the end pose
}

}  // namespace rrtdev
