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
    double inv_rho = 0.0;  // 1.0 / rho, once per launch (the division dub_sweep_setup_sc makes per sweep: the same operation, the same bits)
};

__device__ __forceinline__ double dub_lane_f64(double v, int l) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// dub_sweep_setup_sc for a wavefront (uniform operands, uniform result; needs DubCfg::inv_rho): the same values field by field.  The
// header's set-up runs two sine / cosine series one behind the other (the headings at the two junctions) and every lane computes
// both alike; the junction headings are known before either series (th0 +- t, then +- p), so here lane 0 evaluates the first and
// lane 1 the second in ONE pass of dub_sincos, and the division by rho is the launch's.
__device__ __forceinline__ dub_sweep_t dub_sweep_setup_wave(double x0, double y0, double th0, double sn0, double cs0, const dub_path_t &path, const DubCfg &c,
                                                            int lane) {
    dub_sweep_t s;
    s.x0 = x0;
    s.y0 = y0;
    s.th0 = th0;
    s.rho = c.rho;
    s.inv_rho = c.inv_rho;
    s.t = path.t;
    s.p = path.p;
    s.q = path.q;
    s.k0 = dub_seg_kind(path.word, 0);
    s.k1 = dub_seg_kind(path.word, 1);
    s.k2 = dub_seg_kind(path.word, 2);
    s.sn0 = sn0;
    s.cs0 = cs0;
    // the headings behind the first and the second segment, as dub_advance_sc forms them
    s.th1 = s.k0 == 0 ? th0 : (s.k0 > 0 ? th0 + s.t : th0 - s.t);
    s.th2 = s.k1 == 0 ? s.th1 : (s.k1 > 0 ? s.th1 + s.p : s.th1 - s.p);
    double sv, cv;
    dub_sincos((lane & 1) ? s.th2 : s.th1, &sv, &cv);
    const double s1 = dub_lane_f64(sv, 0), c1 = dub_lane_f64(cv, 0), s2 = dub_lane_f64(sv, 1), c2 = dub_lane_f64(cv, 1);
    // dub_advance_sc(0, 0, th0, sn0, cs0, k0, t, ...)
    if (s.k0 == 0) {
        s.x1 = DUB_FMA(cs0, s.t, 0.0);
        s.y1 = DUB_FMA(sn0, s.t, 0.0);
        s.sn1 = sn0;
        s.cs1 = cs0;
    } else if (s.k0 > 0) {
        s.x1 = 0.0 + (s1 - sn0);
        s.y1 = 0.0 - (c1 - cs0);
        s.sn1 = s1;
        s.cs1 = c1;
    } else {
        s.x1 = 0.0 - (s1 - sn0);
        s.y1 = 0.0 + (c1 - cs0);
        s.sn1 = s1;
        s.cs1 = c1;
    }
    // dub_advance_sc(x1, y1, th1, sn1, cs1, k1, p, ...)
    if (s.k1 == 0) {
        s.x2 = DUB_FMA(s.cs1, s.p, s.x1);
        s.y2 = DUB_FMA(s.sn1, s.p, s.y1);
        s.sn2 = s.sn1;
        s.cs2 = s.cs1;
    } else if (s.k1 > 0) {
        s.x2 = s.x1 + (s2 - s.sn1);
        s.y2 = s.y1 - (c2 - s.cs1);
        s.sn2 = s2;
        s.cs2 = c2;
    } else {
        s.x2 = s.x1 - (s2 - s.sn1);
        s.y2 = s.y1 + (c2 - s.cs1);
        s.sn2 = s2;
        s.cs2 = c2;
    }
    s.nsamples = (int32_t)floor(path.len / DUB_DS) + 1;
    return s;
}

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
    const dub_sweep_t s = (c.htab && c.inv_rho != 0.0)
                              ? dub_sweep_setup_wave((double)ux(a), (double)uy(a), c.htab[ha], c.htab[256 + ha], c.htab[512 + ha], path, c, lane)
                              : (c.htab ? dub_sweep_setup_sc((double)ux(a), (double)uy(a), c.htab[ha], c.htab[256 + ha], c.htab[512 + ha], &path, c.rho)
                                        : dub_sweep_setup((double)ux(a), (double)uy(a), dub_heading(ha, c.nh), &path, c.rho));
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
