// aai_rot_quad.hpp -- second formulation of the rotated-lattice area average (K2), built for fp32 issue rates.
//
// Replaces Source.cpp:413-579 + 986-1431 of the reference for every dst pixel that does not sit on one of the
// reference's DBL_EPSILON knife edges (those are redone by the strict replay, aai_strict.hpp).  Shared between
// the HIP kernel (aai_rotated_quad.hip) and the host-side replay the CPU test-suite uses
// (tests/emulation/host_emulation.cpp): AAI_HD, templated on the arithmetic type F (float in production,
// double in the tests that compare the formulas with the older ones to 1e-12).
//
// Geometry, in the frame of the dst square (a along its top edge, b along its left edge; the square is
// |a| <= h, |b| <= h, h = L/2 > sqrt(2)/2): a unit source pixel is a square turned by the reduced angle whose
// corners lie at distance k = (c+s)/2 along either axis.  With A = h - |a| and B = h - |b| the inside-distances
// of the pixel centre from the NEARER left/right and top/bottom edge line:
//   * min(A,B) <= -k            the pixel is outside                                        (type 0)
//   * min(A,B) >=  k            inside, area 1                                              (type 1)
//   * max(A,B) >=  k            ONE edge line cuts it ("single"): the area is a function of t = min(A,B)+k
//                               alone -- corner triangle, trapezoid or their complements   (types 2, 3, 4)
//   * otherwise both near lines cut it ("double"): the pixel meets the quadrant the two near edges span at
//     their common vertex V.  Either V lies inside the pixel -- exactly four pixels of every dst pixel, handled
//     one vertex at a time by quad_vertex_area (types 7, 8, 9) -- or each edge, as a SEGMENT ending at V,
//     crosses the whole pixel or misses it, so that area = G(u) S1 + G(v) S2 - S1 S2 with G the single-line
//     area and S1, S2 two sign tests on V's position relative to the pixel (types 2-6).
// The reference multiplies the complementary legs when a lone left/right edge cuts off exactly one pixel corner
// (Source.cpp:1055-1062); policy REFERENCE reproduces that in the single-line term.
//
// Per dst pixel (quad_pixel): one uniform pass classifies the (at most 8 x 8) window positions into 64-bit position
// masks -- four compares each, no area math -- while the window's pixel values are being fetched; the areas are
// then evaluated class by class over the set bits, so that every lane of a wave runs the same formula at the same
// time, reading the values from the staged window.
//
// Precision: everything is relative to the dst pixel's own centre, so fp32 carries absolute errors of a few
// 1e-7 pixel.  The reference-policy areas are discontinuous where an edge passes through a pixel corner or a
// vertex crosses a pixel side, so fp32 must not DECIDE such cases: SCAN mode runs the same code without pixel
// loads and reports every dst pixel with a decision closer than QuadConsts::margin to its threshold, or with too
// little total area for fp32 weights; the plan stores that per wave next to the knife-edge flags and the fix-up
// pass redoes those waves in double precision (aai_rotated_kernel<STRICT>).
#pragma once

#include "aai_rot_math.hpp"

// a vote over the lanes of the wave on the GPU; the CPU replay evaluates one pixel / cell at a time
#if defined(__HIP_DEVICE_COMPILE__)
#define AAI_WAVE_ALL(x) (__all(x) != 0)
#else
#define AAI_WAVE_ALL(x) (x)
#endif

namespace aai {

constexpr int kQuadMaxWin = 8;        // window positions per axis that the 64-bit position masks can hold

template <typename F>
struct QuadConsts {
    F c, s, h, k, k2;                 // cos, sin of the reduced angle (both > 0); L/2; (c+s)/2; c+s
    F hmk, hpk;                       // h - k, h + k
    F lo, hi;                         // min(c,s), max(c,s)
    F r2cs, rhi, trapOff;             // 1/(2cs), 1/hi, lo/(2 hi)
    F rc, rs, hrc;                    // 1/c, 1/s, 1/(2c)
    F m1, im1;                        // s/c, c/s
    F hmkRc, hpkRc, hmkRs, hpkRs;     // (h - k) / c, (h + k) / c, (h - k) / s, (h + k) / s: the thresholds of |a|, |b| in window columns (QuadRows)
    F ox[4], oy[4];                   // vertex i = centre + (ox[i], oy[i]); 0 left/top, 1 right/top, 2 left/bottom, 3 right/bottom
    F hbm;                            // h(c+s) - 1/2 + guard: half extent of the window of pixel CENTRES
    F margin;                         // SCAN: decisions closer than this to their threshold are reported
    F minArea;                        // SCAN: pixels whose total area is below this (and not zero) are reported
    int ref;                          // 1 = AAI_POLICY_REFERENCE
    int win;                          // window positions per axis, <= kQuadMaxWin (wide footprints: of ONE part of the window)
    int parts, winFull;               // parts per axis the window is split into (1: a single window; 2 or 4: quad_wide_parts) and its whole extent
    // fast mode only looks at pixel CENTRES inside the closed square: they lie within h (c + s) of its centre along either
    // lattice axis, i.e. at most floor(2 h (c + s)) + 1 positions per axis -- one fewer than the area mode's window, which
    // also holds the pixels the square merely touches (config 3: 4 x 4 instead of 5 x 5)
    F hbf;                            // h (c + s) + guard
    int winFast;                      // (wide footprints: of ONE part of the window, like `win`)
    int partsFast, winFastFull;       // parts per axis of fast mode's window (1, 2 or 4) and its whole extent
    // Close to an axis (min(c,s) small) the reference's corner-triangle rule has slope ~1/(2 min(c,s)) in t, far too
    // steep for fp32 coordinates, and under either policy the thin corner triangles (area t^2 / (2 c s), t tiny) need t to
    // a RELATIVE accuracy fp32 differences of numbers near 1 do not have: hiPrec evaluates both edges' t = h + k - |a|,
    // h + k - |b| in double precision from the centre's double-precision fraction (three fp64 operations each per pair)
    // and only then rounds them to F.
    int hiPrec;
    int steep;                        // hiPrec because of an edge slope ~1 / min(c, s): the vertex area formula in double precision too
    F marginT;                        // SCAN, hiPrec: margin of the t thresholds lo / hi taken on the precise t
    double cD, sD, hpkD, hmkD;
    double oxD[4], oyD[4], m1D, im1D;     // hiPrec: the vertex offsets and edge slopes in double precision
};

// fp32 error budget of a coordinate relative to the dst pixel's centre: the constants' rounding times an index of at
// most the window size, two fused multiply-adds on magnitudes <= hb + 1, the centre's own fraction (see quad_pixel)
AAI_HD double quad_coord_eps(double side, double c, double s) { return 1.1920929e-7 * (2.0 + 0.5 * side * (c + s)); }

// Does fp32 carry the edges' t?  The steepest area formula -- the reference's corner-triangle rule, slope
// (1/c + 1/s)/2 in t -- must keep a coordinate error of quad_coord_eps below ~1.5e-7 of the dst value (weights sum to about
// L^2; pixel values differ from their mean by a few tenths).  Closer to the axes than that (reduced angle within a few
// degrees of 0 or 90) t is taken in double precision (QuadConsts::hiPrec).
AAI_HD bool quad_needs_hiprec(double side, double c, double s)
{
    const double amp = 0.5 * (1.0 / c + 1.0 / s);
    return amp * quad_coord_eps(side, c, s) * 0.5 > 1.5e-7 * side * side;
}

// Can the quad formulation serve this geometry?
//   * the window of source pixels fits the 8 x 8 position masks;
//   * h - k is well away from zero, so that the sign of a, b is never in doubt for a pixel both near lines cut;
//   * the reduced angle is not within ~0.006 degrees of an axis (sin or cos below 1e-4: the double-precision kernel's
//     territory, where even 1/sin overflows fp32 products).
AAI_HD bool quad_supported(double side, double c, double s)
{
    if (!(c > 1e-4 && s > 1e-4)) return false;
    const double h = 0.5 * side, k = 0.5 * (c + s);
    const double m = h * (c + s) - 0.5 + 1e-5;
    return (int)floor(2.0 * m) + 3 <= kQuadMaxWin && h - k > 1e-3;
}

// Wide footprints (dst pixels of more than ~5 source pixels a side): the window of up to 16 x 16 / 32 x 32 positions is split
// into 2 x 2 / 4 x 4 PARTS of at most kQuadMaxWin positions a side, each evaluated like a window of its own (quad_pixel, PART)
// by a lane of its own, the sums added afterwards (quad_parts_sum).  Returns the parts per axis, 0: not a wide footprint the
// formulation serves.  The coordinate error grows with the window (quad_coord_eps) but the areas it perturbs are a thin ring
// of the L^2 the weights sum to.
AAI_HD int quad_wide_parts(double side, double c, double s)
{
    if (!(c > 1e-4 && s > 1e-4)) return 0;
    const double h = 0.5 * side, k = 0.5 * (c + s);
    if (!(h - k > 1e-3)) return 0;
    const double m = h * (c + s) - 0.5 + 1e-5;
    if (!(m < 64.0)) return 0;
    const int win = (int)floor(2.0 * m) + 3;
    if (win <= kQuadMaxWin) return 0;
    return win <= 2 * kQuadMaxWin ? 2 : (win <= 4 * kQuadMaxWin ? 4 : 0);
}

// Fast mode looks at pixel centres only, so its window is two positions narrower: 1 = one window of at most 8 x 8 (aai_quad_fast_kernel),
// 2 / 4 = parts per axis (aai_wide_fast_kernel), 0 = the double-precision line-walking kernel.
AAI_HD int quad_fast_parts(double side, double c, double s)
{
    if (!(c > 1e-4 && s > 1e-4)) return 0;
    const double h = 0.5 * side, k = 0.5 * (c + s);
    if (!(h - k > 1e-3)) return 0;
    const double hb = h * (c + s) + 1e-5;
    if (!(hb < 64.0)) return 0;
    const int win = (int)floor(2.0 * hb) + 1;
    return win <= kQuadMaxWin ? 1 : (win <= 2 * kQuadMaxWin ? 2 : (win <= 4 * kQuadMaxWin ? 4 : 0));
}

// the sum of the parts' partial sums, in the order the lanes of a dst pixel exchange them (a butterfly over lane distance 1, 2,
// 4, ...): v[0] afterwards
template <typename F>
AAI_HD F quad_parts_sum(F *v, int n)
{
    for (int o = 1; o < n; o <<= 1)
        for (int i = 0; i + o < n; i += 2 * o) v[i] = v[i] + v[i + o];
    return v[0];
}

template <typename F>
AAI_HD QuadConsts<F> make_quad_consts(double side, double c, double s, int policy, int scale = 1)
{
    QuadConsts<F> q;
    const double h = 0.5 * side, k = 0.5 * (c + s);
    const double lo = c < s ? c : s, hi = c < s ? s : c;
    q.c = (F)c; q.s = (F)s; q.h = (F)h; q.k = (F)k; q.k2 = (F)(c + s);
    q.hmk = (F)(h - k); q.hpk = (F)(h + k);
    q.lo = (F)lo; q.hi = (F)hi;
    q.r2cs = (F)(1.0 / (2.0 * c * s)); q.rhi = (F)(1.0 / hi); q.trapOff = (F)(lo / (2.0 * hi));
    q.rc = (F)(1.0 / c); q.rs = (F)(1.0 / s); q.hrc = (F)(0.5 / c);
    q.m1 = (F)(s / c); q.im1 = (F)(c / s);
    q.hmkRc = (F)((h - k) / c); q.hpkRc = (F)((h + k) / c); q.hmkRs = (F)((h - k) / s); q.hpkRs = (F)((h + k) / s);
    const double o0x = -h * (c + s), o0y = h * (s - c), o1x = h * (c - s), o1y = -h * (s + c);
    q.ox[0] = (F)o0x; q.oy[0] = (F)o0y; q.ox[1] = (F)o1x; q.oy[1] = (F)o1y;
    q.ox[2] = (F)-o1x; q.oy[2] = (F)-o1y; q.ox[3] = (F)-o0x; q.oy[3] = (F)-o0y;
    const double hb = h * (c + s);
    const double eps = quad_coord_eps(side, c, s);
    q.hbm = (F)(hb - 0.5 + 1e-5);
    q.margin = (F)(3.0 * eps);
    q.minArea = (F)(side * side < 4.0 ? 0.25 * side * side : 1.0);
    q.ref = policy == AAI_POLICY_REFERENCE ? 1 : 0;
    q.winFull = (int)floor(2.0 * (hb - 0.5 + 1e-5)) + 3;
    q.parts = q.winFull <= kQuadMaxWin ? 1 : (q.winFull <= 2 * kQuadMaxWin ? 2 : 4);
    q.win = (q.winFull + q.parts - 1) / q.parts;
    q.hbf = (F)(hb + 1e-5);
    q.winFastFull = (int)floor(2.0 * (hb + 1e-5)) + 1;
    q.partsFast = q.winFastFull <= kQuadMaxWin ? 1 : (q.winFastFull <= 2 * kQuadMaxWin ? 2 : 4);
    q.winFast = (q.winFastFull + q.partsFast - 1) / q.partsFast;
    // ... and with replicated source pixels (up-sampling): a dst pixel then covers one or two source pixels and a dst value is
    // nearly a copy of a source value, so on noisy data it can be a hundred times smaller than its neighbours -- where the
    // 1e-7 absolute error of an fp32 area shows as several 1e-6 relative (7e-6 on 8-bit noise; 4e-7 with hiPrec)
    q.steep = quad_needs_hiprec(side, c, s) ? 1 : 0;
    q.hiPrec = (q.steep || scale > 1) ? 1 : 0;
    q.marginT = (F)(1e-6 * lo);
    q.cD = c; q.sD = s; q.hpkD = h + k; q.hmkD = h - k;
    q.oxD[0] = o0x; q.oyD[0] = o0y; q.oxD[1] = o1x; q.oyD[1] = o1y;
    q.oxD[2] = -o1x; q.oyD[2] = -o1y; q.oxD[3] = -o0x; q.oyD[3] = -o0y;
    q.m1D = s / c; q.im1D = c / s;
    return q;
}

// How far from a dst pixel's centre, along either lattice axis, the window kernels fetch at most (rotated_band_source_rows sizes a
// row band's source rows from these; tests/emulation replays the windows themselves against the rows it reports).  The window of
// quad_pixel starts at Xc + floor(fpx - hbm) with Xc the lattice point nearest the centre and fpx = centre - Xc, and holds parts x
// win positions (a wide footprint's parts may overhang winFull); quad_fast_pixel's starts at Xc + ceil(fpx - hbf).
template <typename F>
AAI_HD double quad_window_reach(const QuadConsts<F> &q)
{
    const double lo = (double)q.hbm + 1.0, hi = (double)(q.parts * q.win) - 1.0 - (double)q.hbm;
    return (lo > hi ? lo : hi) + 1e-3;
}
template <typename F>
AAI_HD double quad_fast_window_reach(const QuadConsts<F> &q)
{
    const double lo = (double)q.hbf, hi = (double)(q.partsFast * q.winFast) - (double)q.hbf;
    return (lo > hi ? lo : hi) + 1e-3;
}

// ---- the source footprint of a 16 x 16 dst tile (aai_quad_fast_lds_kernel stages it through LDS) ------------------------------------
// Pixel centres are affine in (dx, dy) (quad_centre) and every window reaches quad_fast_window_reach beyond its pixel's centre, so the
// windows of a tile lie in a box whose corners sit at fixed offsets from the centre of the tile's first pixel.
struct FastTile {
    double x0off, x1off, y0off, y1off;       // the box relative to the centre of the tile's first pixel, virtual frame
    int maxSide;                              // upper bound of the box's side (lattice points): bounds the LDS pitch
};
constexpr int kFastLdsBytes = 64 * 1024;
// false: the box does not fit the LDS budget (or a side exceeds two elements per lane of a wave)
template <typename F>
AAI_HD bool make_fast_tile(const RotLaunch &r, const QuadConsts<F> &q, FastTile &ft)
{
    const double reach = quad_fast_window_reach(q) + 0.01;
    const double xa = 15.0 * r.cXa, xb = 15.0 * r.cXb, ya = 15.0 * r.cYa, yb = 15.0 * r.cYb;
    ft.x0off = (xa < 0.0 ? xa : 0.0) + (xb < 0.0 ? xb : 0.0) - reach; ft.x1off = (xa > 0.0 ? xa : 0.0) + (xb > 0.0 ? xb : 0.0) + reach;
    ft.y0off = (ya < 0.0 ? ya : 0.0) + (yb < 0.0 ? yb : 0.0) - reach; ft.y1off = (ya > 0.0 ? ya : 0.0) + (yb > 0.0 ? yb : 0.0) + reach;
    const double sx = ft.x1off - ft.x0off, sy = ft.y1off - ft.y0off, side = sx > sy ? sx : sy;
    if (!(side < 126.0)) return false;
    ft.maxSide = (int)ceil(side) + 2;
    return (size_t)(ft.maxSide | 1) * (size_t)ft.maxSide * sizeof(float) <= (size_t)kFastLdsBytes;
}
// the box [X0, X1] x [Y0, Y1] of the tile whose first pixel is (dx0, dy0), clipped to the lattice; false: it misses the lattice
AAI_HD bool fast_tile_box(const RotLaunch &r, const FastTile &ft, int dx0, int dy0, int &X0, int &X1, int &Y0, int &Y1)
{
    double px0, py0;
    quad_centre(r, dx0, dy0, px0, py0);
    const double bx0 = floor(px0 + ft.x0off), bx1 = ceil(px0 + ft.x1off), by0 = floor(py0 + ft.y0off), by1 = ceil(py0 + ft.y1off);
    const double wMax = (double)(r.mW - 1), hMax = (double)(r.mH - 1);
    if (!(bx0 <= wMax && bx1 >= 0.0 && by0 <= hMax && by1 >= 0.0)) return false;
    X0 = (int)(bx0 > 0.0 ? bx0 : 0.0); X1 = (int)(bx1 < wMax ? bx1 : wMax);
    Y0 = (int)(by0 > 0.0 ? by0 : 0.0); Y1 = (int)(by1 < hMax ? by1 : hMax);
    return true;
}

// minimum / maximum of two numbers that are never NaN (coordinates, areas): ONE instruction on the GPU (v_min_f32 / v_max_f32),
// where "a < b ? a : b" is a compare and a select because it must hand a NaN through
AAI_HD float qmin(float a, float b) { return __builtin_fminf(a, b); }
AAI_HD float qmax(float a, float b) { return __builtin_fmaxf(a, b); }
AAI_HD double qmin(double a, double b) { return __builtin_fmin(a, b); }
AAI_HD double qmax(double a, double b) { return __builtin_fmax(a, b); }
AAI_HD float qabs(float a) { return __builtin_fabsf(a); }       // a source modifier on the GPU, not an instruction
AAI_HD double qabs(double a) { return __builtin_fabs(a); }
// Every multiply-add of this header is an EXPLICIT fused multiply-add and its translation units are compiled without
// contraction: the kernels' instantiations (plain / interleaved, scan / production) and the CPU replay of the test-suite
// then execute the same IEEE operations and agree bit for bit, instead of differing in the last place wherever a
// compiler chose to fuse differently.
AAI_HD float qfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
AAI_HD double qfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// two independent fused multiply-adds: r0 = a0 b0 + c0, r1 = a1 b1 + c1 -- ONE packed instruction on the GPU (v_pk_fma_f32), the
// same two IEEE operations on the CPU
AAI_HD void qfma2(float a0, float a1, float b0, float b1, float c0, float c1, float &r0, float &r1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 a = {a0, a1}, b = {b0, b1}, c = {c0, c1};
    const f2 r = __builtin_elementwise_fma(a, b, c);
    r0 = r.x; r1 = r.y;
#else
    r0 = __builtin_fmaf(a0, b0, c0); r1 = __builtin_fmaf(a1, b1, c1);
#endif
}
AAI_HD void qfma2(double a0, double a1, double b0, double b1, double c0, double c1, double &r0, double &r1)
{
    r0 = __builtin_fma(a0, b0, c0); r1 = __builtin_fma(a1, b1, c1);
}

// Area of the part of a unit pixel on the inner side of ONE edge line that has entered it by t (0 <= t <= c+s,
// measured from the pixel's extreme corner along the line's normal).  substitute: the line is a left/right edge
// under policy REFERENCE -- the two corner-triangle cases take the reference's complementary legs.
template <typename F>
AAI_HD F quad_cut_tp(const QuadConsts<F> &q, F tp, bool flip, bool substitute)
{
    // tp = min(t, c + s - t): the line's distance from the NEARER extreme corner (the cases mirror at t = k); flip: t > k
    const F trap = qfma(tp, q.rhi, -q.trapOff);           // (tp - lo/2) / hi
    const F triExact = (tp * tp) * q.r2cs;                // legs tp/c, tp/s
    const F triRef = qfma(-q.hrc, tp, F(0.5)) * qfma(-q.rs, tp, F(1));
    const F tri = substitute ? triRef : triExact;
    const F g = tp <= q.lo ? tri : trap;
    return flip ? F(1) - g : g;
}
template <typename F>
AAI_HD F quad_cut(const QuadConsts<F> &q, F t, bool substitute)
{
    return quad_cut_tp(q, qmin(t, q.k2 - t), t > q.k, substitute);
}

// A pixel cut by both near edge lines, the vertex V NOT inside it (a pixel that holds V gets 0 here and its area
// from quad_vertex_area).  A, B as above; sameSign: a and b have the same sign, which makes edge 1 of the
// canonical orientation (u, v) the left/right edge.  nearS (SCAN): distance of the closest live sign test from
// its threshold.
template <typename F, bool SCAN>
AAI_HD F quad_double(const QuadConsts<F> &q, F A, F B, F tpA, bool flipA, F tpB, bool flipB, bool sameSign, F &nearS)
{
    // (tpA, flipA), (tpB, flipB) = the left/right and the top/bottom edge's t = A + k, B + k in mirrored form
    // (quad_cut_tp; from double precision under hiPrec)
    const F u = sameSign ? A : B, v = sameSign ? B : A;
    // V relative to the pixel centre along the pixel's own axes, in the orientation where the square lies
    // towards -x, -y of V: edge 1 runs from V towards -y, edge 2 towards -x
    const F cx = qfma(u, q.c, v * q.s), cy = qfma(v, q.c, -(u * q.s));
    const bool r = cx >= F(0.5);
    const bool S1 = r || cy >= F(0.5);                    // edge 1 crosses the pixel (V beyond its +x or +y side)
    const bool S2 = r || cy <= F(-0.5);                   // edge 2 crosses the pixel (V beyond its +x or -y side)
    if (SCAN) {
        const F dx = qabs(cx - F(0.5));
        nearS = dx;
        if (cx < F(0.5) + q.margin) nearS = qmin(dx, qmin(qabs(cy - F(0.5)), qabs(cy + F(0.5))));
    }
    // the reference's corner rule applies to a left/right edge that crosses the pixel ALONE (edge 1 iff sameSign)
    const F gA = quad_cut_tp(q, tpA, flipA, q.ref != 0 && (sameSign ? !S2 : !S1));
    const F gB = quad_cut_tp(q, tpB, flipB, false);
    const F g1 = sameSign ? gA : gB, g2 = sameSign ? gB : gA;
    const F both = qmax(g1 + g2 - F(1), F(0));
    return S1 ? (S2 ? both : g1) : (S2 ? g2 : F(0));
}

// The pixel that holds vertex `vidx` of the dst square: area of the pixel inside the right-angled wedge the two
// edges span at the vertex.  (fx, fy) = the vertex relative to the pixel centre, both in [-1/2, 1/2].  Half the
// sum over the pixel's sides of (distance of the vertex from the side) x (length of the side inside the wedge);
// vertices 1-3 are vertex 0 seen through a quarter turn.  Exact for both policies (Source.cpp:1276-1401).
template <typename F>
AAI_HD F quad_vertex_area(const QuadConsts<F> &q, F fx, F fy, int vidx);
template <typename F>
AAI_HD F quad_vertex_area(F m1, F im1, F fx, F fy, int vidx)
{
    // (selects, not a switch: the wide kernel's lanes each evaluate their own vertex)
    const bool straight = vidx == 0 || vidx == 3;
    const F ax = straight ? fx : fy, ay = straight ? fy : fx;
    const F x = vidx >= 2 ? -ax : ax, y = (vidx & 1) ? -ay : ay;      // 0: (fx, fy)  1: (fy, -fx)  2: (-fy, fx)  3: (-fx, -fy)
    // vertex 0: the wedge opens towards +x between the rays (c,-s) and (s,c)
    const F dR = F(0.5) - x, dT = y + F(0.5), dB = F(0.5) - y;
    const F y1 = qfma(-dR, m1, y), y2 = qfma(dR, im1, y);          // where the two rays meet the line x = 1/2
    const F lenR = qmin(y2, F(0.5)) - qmax(y1, F(-0.5));
    const F lenT = qmax(F(0), F(0.5) - qfma(dT, im1, x));    // ray 1 leaves through the top side
    const F lenB = qmax(F(0), F(0.5) - qfma(dB, m1, x));     // ray 2 leaves through the bottom side
    return F(0.5) * qfma(dR, lenR, qfma(dT, lenT, dB * lenB));
}
template <typename F>
AAI_HD F quad_vertex_area(const QuadConsts<F> &q, F fx, F fy, int vidx) { return quad_vertex_area<F>(q.m1, q.im1, fx, fy, vidx); }

AAI_HD int quad_ctz(unsigned long long m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)m) - 1;
#else
    return __builtin_ctzll(m);
#endif
}
AAI_HD int quad_ctz(unsigned m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)m) - 1;
#else
    return __builtin_ctz(m);
#endif
}
AAI_HD int quad_popcount(unsigned long long m) { return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32)); }
AAI_HD int quad_popcount(unsigned m) { return __builtin_popcount(m); }
// v where keep is all ones, +0 where it is zero
AAI_HD float quad_keep_bits(float v, int keep)
{
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    u &= (unsigned)keep;
    __builtin_memcpy(&v, &u, 4);
    return v;
}
AAI_HD double quad_keep_bits(double v, int keep) { return keep ? v : 0.0; }
// position masks: one bit per window slot -- 32 bits are enough up to 5 x 5
template <int WIN, bool SMALL = (WIN * WIN <= 32)> struct QuadMask { typedef unsigned long long type; };
template <int WIN> struct QuadMask<WIN, true> { typedef unsigned type; };

// A plane of window-slot bits, filled from the LAST slot down to slot 0: p = 2 p + bit is ONE instruction per plane and position
// on the GPU (v_addc_co_u32 with the compare's lane mask as carry-in) where "if (c) p |= bit" is a select and an or.
AAI_HD void quad_push_bit(unsigned &p, bool bit)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(bit);      // the lane mask the compare produced (an SGPR pair)
    asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(p) : "s"(lanes) : "vcc");
#else
    p = p + p + (bit ? 1u : 0u);
#endif
}
template <int WIN>
struct QuadPlane {
    unsigned lo = 0, hi = 0;
    AAI_HD void push(int slot, bool bit) { if (slot >= 32) quad_push_bit(hi, bit); else quad_push_bit(lo, bit); }      // slots in DESCENDING order
    AAI_HD typename QuadMask<WIN>::type mask() const
    {
        typedef typename QuadMask<WIN>::type mask_t;
        return WIN * WIN <= 32 ? (mask_t)lo : (mask_t)(((unsigned long long)hi << 32) | lo);
    }
};

// Source indices of the WIN window positions g0 ... g0 + WIN - 1 along one axis of a REPLICATED lattice (mN = n * scale virtual
// pixels): every position is clamped onto the lattice first, then divided by the scale relative to the FIRST clamped position -- a
// non-negative coordinate whose floor division is exact in double precision, and non-negative steps (rem + i + 0.5) / scale with rem
// + i < scale + WIN, far from every integer compared with fp32 rounding.  Any g0: the cell kernel evaluates cells whose window misses
// the lattice altogether instead of branching around them (tests/emulation: aai_emu_replicated_indices).
template <int WIN>
AAI_HD void replicated_indices(int g0, int mN, int scale, double invScaleD, float invScale, int (&q)[WIN])
{
    const int o = g0 < 0 ? 0 : (g0 > mN - 1 ? mN - 1 : g0);
    const int q0 = (int)(((double)o + 0.5) * invScaleD);
    const float rem = (float)(o - q0 * scale) + 0.5f;
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
        const int g = g0 + i, c = g < 0 ? 0 : (g > mN - 1 ? mN - 1 : g);
        q[i] = q0 + (int)((rem + (float)(c - o)) * invScale);          // (c - o in [0, WIN))
    }
}

// ---- classification by intervals ------------------------------------------------------------------------------------------------
// The dst-frame coordinates of the window positions of one row grow with the column (cos, sin > 0), so a threshold is crossed at ONE
// column, and "which positions of this row lie below X" is a run of bits from column 0 on: the crossing column (one addition per row
// and threshold), a clamp, a rounding and a bit-run instruction -- where testing the positions one by one costs a multiply-add, a
// compare and a shift-in EACH.  The helpers below are shared by the quad, wide and cell formulations.
// bits [OFF, OFF + n) of a word, n = the low five bits of `count`: ONE instruction on the GPU (v_bfm_b32)
template <int OFF>
AAI_HD unsigned quad_bit_run(unsigned count)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm("v_bfm_b32 %0, %1, %2" : "=v"(r) : "v"(count), "n"(OFF));
    return r;
#else
    return ((1u << (count & 31u)) - 1u) << OFF;
#endif
}
// How many of the columns 0 .. WIN - 1 lie below e, as the low bits of a word; eHalf = e + 1/2.  Clamped to [0, WIN], then rounded to
// the nearest integer by adding 1.5 x 2^23 (whose low bits are zero): ceil(e) for every e that is not an integer -- and a threshold
// crossed exactly AT a column is a decision the scan leaves to the fix-up pass anyway.  Two instructions.
template <int WIN>
AAI_HD unsigned quad_columns_below(double eHalf)        // (the CPU checks' double-precision instantiation)
{
    return (unsigned)__builtin_nearbyint(qmin(qmax(eHalf, 0.0), (double)WIN));
}
template <int WIN>
AAI_HD unsigned quad_columns_below(float eHalf)
{
    const float t = qmin(qmax(eHalf, 0.f), (float)WIN) + 12582912.f;
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__float_as_int(t);
#else
    unsigned u;
    __builtin_memcpy(&u, &t, 4);
    return u;
#endif
}
// a plane of window-slot bits built row by row
struct RowPlane {
    unsigned lo = 0, hi = 0;
    template <int WIN>
    AAI_HD typename QuadMask<WIN>::type whole() const
    {
        typedef typename QuadMask<WIN>::type mask_t;
        return WIN * WIN <= 32 ? (mask_t)lo : (mask_t)(((unsigned long long)hi << 32) | lo);
    }
};
// the bits of window row J (slots J WIN ... J WIN + WIN - 1) below column count `n`, into a plane
template <int WIN, int J>
AAI_HD void quad_row_bits(unsigned n, RowPlane &p)
{
    if (J * WIN + WIN <= 32) p.lo |= quad_bit_run<(J * WIN + WIN <= 32 ? J * WIN : 0)>(n);
    else if (J * WIN >= 32) p.hi |= quad_bit_run<(J * WIN >= 32 ? J * WIN - 32 : 0)>(n);
    else {
        const unsigned long long b = (unsigned long long)quad_bit_run<0>(n) << (J * WIN);
        p.lo |= (unsigned)b; p.hi |= (unsigned)(b >> 32);
    }
}
// rows J ... WIN - 1 of quad_pixel's window: (a0, b0) = the columns (+ 1/2) where a and b cross zero in this row; |a| <= h - k between
// the columns a0 -/+ hmkRc, |a| < h + k between a0 -/+ hpkRc, the same for b with 1 / s
template <int WIN, int J>
struct QuadRows {
    template <typename F>
    static AAI_HD void run(const QuadConsts<F> &q, F a0, F b0, RowPlane &aLo, RowPlane &aHi, RowPlane &bLo, RowPlane &bHi, RowPlane &tLo, RowPlane &tHi)
    {
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(a0 - q.hmkRc), aLo);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(a0 + q.hmkRc), aHi);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(b0 - q.hmkRs), bLo);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(b0 + q.hmkRs), bHi);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(qmax(a0 - q.hpkRc, b0 - q.hpkRs)), tLo);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(qmin(a0 + q.hpkRc, b0 + q.hpkRs)), tHi);
        QuadRows<WIN, J + 1>::run(q, a0 + q.m1, b0 - q.im1, aLo, aHi, bLo, bHi, tLo, tHi);
    }
};
template <int WIN>
struct QuadRows<WIN, WIN> {
    template <typename F>
    static AAI_HD void run(const QuadConsts<F> &, F, F, RowPlane &, RowPlane &, RowPlane &, RowPlane &, RowPlane &, RowPlane &) {}
};

// One dst pixel.  WIN = window positions per axis (QuadConsts::win, a compile-time constant so that the window pass
// unrolls and the staged window has a fixed size); window position (i, j) is bit / slot j * WIN + i.
// (Xc, Yc) = the virtual pixel nearest the centre, (fpx, fpy) = centre - (Xc, Yc), both in [-1/2, 1/2].
// Source protocol:
//   src.issue(xg0, yg0, valid)  start fetching the values of the window whose position (0, 0) is virtual pixel
//                               (xg0, yg0); `valid` has the bits of the positions inside the mW x mH lattice
//   src.commit()                make them addressable (the GPU parks them in LDS, one column per lane)
//   src.at(slot, vals)          the NC channel values of a position whose valid bit is set
// so that the loads are in flight while the window is classified and every later pass reads at LDS latency.
// Returns the sums; the dst value of channel c is sumVA[c] / sumA, or 0 when sumA is 0 (Source.cpp:577).  NC = 1 for
// a plain image; interleaved channels share every area (NC = 4 accumulators, unused ones stay 0).
// SCAN: src is never touched, every value counts as 1 and the return value says whether this pixel must be left to
// the double-precision pass.
// HP: QuadConsts::hiPrec as a compile-time switch (the double-precision code costs registers even where it never runs)
// PART: this call evaluates part (partI, partJ) of a wide window (quad_wide_parts) -- WIN x WIN positions from (partI, partJ) *
// WIN on; a vertex pixel belongs to the part that holds it; the caller adds the parts and (SCAN) tests the total area
template <typename F, int WIN, bool SCAN, bool HP, int NC, typename Src, bool PART = false>
AAI_HD bool quad_pixel(const QuadConsts<F> &q, int Xc, int Yc, double dfx, double dfy, int mW, int mH, Src &src, F &sumA, F (&sumVA)[NC],
                       int partI = 0, int partJ = 0)
{
    typedef typename QuadMask<WIN>::type u64;
    const F fpx = (F)dfx, fpy = (F)dfy;          // (32 bits for windows up to 5 x 5)
    static_assert(WIN >= 2 && WIN <= kQuadMaxWin, "window size");
    sumA = F(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) sumVA[c] = F(0);
    // sumVA += w * (the values of one window slot); SCAN counts every value as 1
    auto accumulate = [&](F w, int slot) {
        F vals[NC];
        if (SCAN) {
#pragma unroll
            for (int c = 0; c < NC; ++c) vals[c] = F(1);
        } else src.at(slot, vals);
#pragma unroll
        for (int c = 0; c < NC; ++c) sumVA[c] = qfma(w, vals[c], sumVA[c]);
    };
    // window origin: first pixel centre the square's bounding box can reach
    const F fi0 = floor(fpx - q.hbm) + (PART ? (F)(partI * WIN) : F(0)), fj0 = floor(fpy - q.hbm) + (PART ? (F)(partJ * WIN) : F(0));
    const int i0 = (int)fi0, j0 = (int)fj0;
    const int xg0 = Xc + i0, yg0 = Yc + j0;

    // positions inside the lattice
    u64 valid;
    {
        const int ia = xg0 < 0 ? -xg0 : 0, ib = (mW - 1 - xg0 < WIN - 1) ? mW - 1 - xg0 : WIN - 1;
        const int ja = yg0 < 0 ? -yg0 : 0, jb = (mH - 1 - yg0 < WIN - 1) ? mH - 1 - yg0 : WIN - 1;
        if (ia > ib || ja > jb) return false;                                 // the whole window misses the image
        const unsigned cols = (2u << ib) - (1u << ia);                         // bits ia..ib of one row
        valid = 0;
#pragma unroll
        for (int j = 0; j < WIN; ++j)
            if (j >= ja && j <= jb) valid |= (u64)cols << (j * WIN);
    }
    if (!SCAN) src.issue(xg0, yg0, valid);

    // dst-frame coordinates of pixel (Xc, Yc)'s centre: (ex, ey) = -(fpx, fpy)
    const F ac = qfma(fpy, q.s, -(fpx * q.c)), bc = -qfma(fpx, q.s, fpy * q.c);
    // hiPrec: both dst-frame coordinates of (Xc, Yc) in double precision
    const double acD = qfma(dfy, q.sD, -(dfx * q.cD)), bcD = -qfma(dfx, q.sD, dfy * q.cD);
    // an edge's t = h + k - |coordinate| for window position (fi, fj) in double precision (lr: the left/right edge, else
    // the top/bottom edge), clamped to [0, c + s] and mirrored about k there as well -- near an axis both t and c + s - t
    // can be tiny differences of numbers near 1, and the thin corner triangles they describe (area t^2 / (2 c s)) are
    // where a dst value far below its neighbours gets its relative accuracy from
    auto precise_tp = [&](F fi, F fj, bool lr, bool &flip) -> F {
        const double ad = lr ? qfma((double)fi, q.cD, qfma(-(double)fj, q.sD, acD)) : qfma((double)fi, q.sD, qfma((double)fj, q.cD, bcD));
        const double m = ad < 0.0 ? -ad : ad;
        // t = (h + k) - |coordinate| and its mirror image c + s - t = |coordinate| - (h - k), each as ONE difference in double
        // precision; the nearer one, clamped at 0, is the mirrored parameter
        const F t1 = (F)(q.hpkD - m), t2 = (F)(m - q.hmkD);
        flip = t1 > t2;
        return qmax(qmin(t1, t2), F(0));
    };
    bool uncertain = false;

    // ---- pass 1: classify every window position --------------------------------------------------------------
    // a = ac + (fi0 + i) c - (fj0 + j) s and b = bc + (fi0 + i) s + (fj0 + j) c grow with the column i, so in row j
    //   a < X  <=>  i < X / c + iA + j s / c,  iA = -ac / c + fj0 s / c - fi0;      b < X  <=>  i < X / s + iB - j c / s
    // and the planes |a| <= h - k, |b| <= h - k, touched (|a| < h + k and |b| < h + k) are runs of columns: QuadRows, ~30 instructions
    // per row where the positions one by one took ~12 each.  The crossing columns carry fp32 rounding of their own: the scan
    // (below) also classifies position by position and reports every pixel where the two disagree.
    u64 mIn, mSingle, mDouble;
    {
        const F iA = qfma(-ac, q.rc, qfma(fj0, q.m1, -fi0)), iB = qfma(-bc, q.rs, qfma(-fj0, q.im1, -fi0));
        RowPlane aLo, aHi, bLo, bHi, tLo, tHi;
        QuadRows<WIN, 0>::run(q, iA + F(0.5), iB + F(0.5), aLo, aHi, bLo, bHi, tLo, tHi);
        const u64 pA = aHi.template whole<WIN>() & ~aLo.template whole<WIN>(), pB = bHi.template whole<WIN>() & ~bLo.template whole<WIN>();
        const u64 pT = tHi.template whole<WIN>() & ~tLo.template whole<WIN>() & valid;
        mIn = pA & pB & valid;                 // wholly inside (inside implies touched: h - k < h + k)
        mSingle = pT & (pA ^ pB);              // one near line clear of the pixel, the other cuts it
        mDouble = pT & ~(pA | pB);             // both near lines cut it
    }
    if (SCAN) {
        // bit planes position by position: |a| <= h - k, |b| <= h - k, touched
        QuadPlane<WIN> plA, plB, plT;
#pragma unroll
        for (int jj = 0; jj < WIN; ++jj) {
            const int j = WIN - 1 - jj;                                  // (last slot first: QuadPlane)
            const F fj = fj0 + (F)j;
            F rowA, rowB;
            qfma2(-fj, fj, q.s, q.c, ac, bc, rowA, rowB);
#pragma unroll
            for (int ii = 0; ii < WIN; ++ii) {
                const int i = WIN - 1 - ii;
                const F fi = fi0 + (F)i;
                F sa, sb;
                qfma2(fi, fi, q.c, q.s, rowA, rowB, sa, sb);
                const F a = qabs(sa), b = qabs(sb);
                const u64 bit = (u64)1 << (j * WIN + i);
                plA.push(j * WIN + i, a <= q.hmk);
                plB.push(j * WIN + i, b <= q.hmk);
                plT.push(j * WIN + i, a < q.hpk && b < q.hpk);
                // thresholds of |a| (the left/right line also switches formula at t = lo, hi under policy REFERENCE)
                // and of |b|; one axis' thresholds only matter while the other axis does not already say "outside"
                F na = qmin(qabs(a - q.hmk), qabs(a - q.hpk));
                if (q.ref && !HP) na = qmin(na, qmin(qabs(a - (q.hpk - q.lo)), qabs(a - (q.hpk - q.hi))));
                const F nb = qmin(qabs(b - q.hmk), qabs(b - q.hpk));
                const bool live = (valid & bit) != 0;
                if (live && ((na < q.margin && b < q.hpk + q.margin) || (nb < q.margin && a < q.hpk + q.margin))) uncertain = true;
            }
        }
        const u64 pA = plA.mask(), pB = plB.mask(), pT = plT.mask() & valid;
        if ((pA & pB & valid) != mIn || (pT & (pA ^ pB)) != mSingle || (pT & ~(pA | pB)) != mDouble) uncertain = true;
    }
    if (!SCAN) src.commit();

    // ---- the four pixels that hold a vertex ----------------------------------------------------------------------
    if (PART && !HP) {
        // A part of a wide window holds ONE of the four vertices as a rule (two or none where the window's split passes beside them).
        // Every lane first finds its vertices (position tests only), then evaluates them in a loop of its own -- one turn in almost
        // every wave -- where four blocks in a row made all lanes sit through the vertices of the other three parts.
        unsigned held = 0;
        F hx[4], hy[4];
        int hs[4];
#pragma unroll
        for (int vtx = 0; vtx < 4; ++vtx) {
            const F wx = fpx + q.ox[vtx], wy = fpy + q.oy[vtx];       // vertex relative to (Xc, Yc)
            const F rx = floor(wx + F(0.5)), ry = floor(wy + F(0.5));
            hx[vtx] = wx - rx; hy[vtx] = wy - ry;
            const int i = (int)rx - i0, j = (int)ry - j0;
            hs[vtx] = j * WIN + i;
            const bool mine = i >= 0 && i < WIN && j >= 0 && j < WIN;      // (else another part's)
            if (mine) held |= 1u << vtx;
            if (SCAN && mine && (qabs(hx[vtx]) > F(0.5) - q.margin || qabs(hy[vtx]) > F(0.5) - q.margin)) uncertain = true;
        }
        while (held) {
            const int vtx = quad_ctz(held);
            held &= held - 1;
            const bool v0 = vtx == 0, v1 = vtx == 1, v2 = vtx == 2;
            const int slot = v0 ? hs[0] : (v1 ? hs[1] : (v2 ? hs[2] : hs[3]));
            const F fx = v0 ? hx[0] : (v1 ? hx[1] : (v2 ? hx[2] : hx[3])), fy = v0 ? hy[0] : (v1 ? hy[1] : (v2 ? hy[2] : hy[3]));
            const u64 bit = (u64)1 << slot;
            mDouble &= ~bit;
            if (valid & bit) {
                const F area = quad_vertex_area(q, fx, fy, vtx);
                sumA += area;
                accumulate(area, slot);
            }
        }
    } else {
#pragma unroll
    for (int vtx = 0; vtx < 4; ++vtx) {
        const F wx = fpx + q.ox[vtx], wy = fpy + q.oy[vtx];       // vertex relative to (Xc, Yc)
        const F rx = floor(wx + F(0.5)), ry = floor(wy + F(0.5));
        const F fx = wx - rx, fy = wy - ry;
        const int i = (int)rx - i0, j = (int)ry - j0;
        if (i < 0 || i >= WIN || j < 0 || j >= WIN) { if (SCAN && !PART) uncertain = true; continue; }   // another part's; else cannot happen (the box holds the vertices)
        const int slot = j * WIN + i;
        const u64 bit = (u64)1 << slot;
        if (SCAN && (qabs(fx) > F(0.5) - q.margin || qabs(fy) > F(0.5) - q.margin)) uncertain = true;
        mDouble &= ~bit;
        if (valid & bit) {
            F area;
            if (HP) {
                // near an axis one of the edge slopes is ~1 / min(c, s): the vertex's position inside its pixel in double
                // precision (from the centre's double-precision fraction), the area formula in double as well
                const double fxD = (dfx + q.oxD[vtx]) - (double)rx, fyD = (dfy + q.oyD[vtx]) - (double)ry;
                if (q.steep) area = (F)quad_vertex_area<double>(q.m1D, q.im1D, fxD, fyD, vtx);
                else area = quad_vertex_area(q, (F)fxD, (F)fyD, vtx);
            } else area = quad_vertex_area(q, fx, fy, vtx);
            sumA += area;
            accumulate(area, slot);
        }
    }
    }

    // ---- pixels wholly inside: area 1 ------------------------------------------------------------------------------
    if (WIN >= 6) {
        // Large windows (the parts of a wide footprint: most of their 36 ... 64 positions are interior): one straight pass over the
        // positions -- a value from a fixed address, a bit test and an add each, in slot order -- instead of a loop over the set
        // bits, whose every turn finds the bit, clears it and computes an address (~14 instructions with 64-bit masks) and which runs
        // as long as the fullest lane of the wave
        sumA += (F)quad_popcount(mIn);
#pragma unroll
        for (int slot = 0; slot < WIN * WIN; ++slot) {
            F vals[NC];
            if (SCAN) {
#pragma unroll
                for (int c = 0; c < NC; ++c) vals[c] = F(1);
            } else src.at(slot, vals);
            // (the position's bit spread over a word and ANDed onto the value's bits -- a select, not a product: a NaN outside stays outside)
            const unsigned word = slot < 32 ? (unsigned)mIn : (unsigned)((unsigned long long)mIn >> 32);
            const int keep = (int)(word << (31 - (slot & 31))) >> 31;
#pragma unroll
            for (int c = 0; c < NC; ++c) sumVA[c] += quad_keep_bits(vals[c], keep);
        }
    } else {
        while (mIn) {
            const int slot = quad_ctz(mIn);
            mIn &= mIn - 1;
            sumA += F(1);
            accumulate(F(1), slot);
        }
    }
    // ---- pixels cut by one edge line ---------------------------------------------------------------------------------
    while (mSingle) {
        const int slot = quad_ctz(mSingle);
        mSingle &= mSingle - 1;
        const int j = slot / WIN, i = slot - j * WIN;
        const F fj = fj0 + (F)j, fi = fi0 + (F)i;
        const F a = qabs(qfma(fi, q.c, qfma(-fj, q.s, ac))), b = qabs(qfma(fi, q.s, qfma(fj, q.c, bc)));
        const bool isLR = a > b;                              // the line nearer the pixel centre is the cutting one
        const F t = qmin(qmax(q.hpk - qmax(a, b), F(0)), q.k2);      // its inside-distance + k
        F tp = qmin(t, q.k2 - t);
        bool flip = t > q.k;
        if (HP) {
            tp = precise_tp(fi, fj, isLR, flip);
            if (SCAN && isLR && q.ref != 0 && qabs(tp - q.lo) < q.marginT) uncertain = true;
        }
        const F area = quad_cut_tp(q, tp, flip, isLR && q.ref != 0);
        sumA += area;
        accumulate(area, slot);
    }
    // ---- pixels cut by both near edge lines --------------------------------------------------------------------------
    while (mDouble) {
        const int slot = quad_ctz(mDouble);
        mDouble &= mDouble - 1;
        const int j = slot / WIN, i = slot - j * WIN;
        const F fj = fj0 + (F)j, fi = fi0 + (F)i;
        const F a = qfma(fi, q.c, qfma(-fj, q.s, ac)), b = qfma(fi, q.s, qfma(fj, q.c, bc));
        F nearS = F(1);
        const F A = q.h - qabs(a), B = q.h - qabs(b);
        const F tA = qmin(qmax(A + q.k, F(0)), q.k2), tB = qmin(qmax(B + q.k, F(0)), q.k2);
        F tpA = qmin(tA, q.k2 - tA), tpB = qmin(tB, q.k2 - tB);
        bool flipA = tA > q.k, flipB = tB > q.k;
        if (HP) {
            tpA = precise_tp(fi, fj, true, flipA);
            tpB = precise_tp(fi, fj, false, flipB);
            if (SCAN && q.ref != 0 && qabs(tpA - q.lo) < q.marginT) uncertain = true;
        }
        const F area = quad_double<F, SCAN>(q, A, B, tpA, flipA, tpB, flipB, (a < F(0)) == (b < F(0)), nearS);
        if (SCAN && nearS < q.margin) uncertain = true;
        sumA += area;
        accumulate(area, slot);
    }
    if (SCAN && !PART && sumA > F(0) && sumA < q.minArea) uncertain = true;
    return uncertain;
}

// Fast mode (Source.cpp:868-907 + 837-864): the mean of the virtual pixels whose CENTRES lie in the closed dst square
// (SURVEY.md B.3).  WIN = QuadConsts::winFast; the same up-front fetch as quad_pixel; membership is two compares per position,
// and the values are summed straight from the registers they were fetched into -- no LDS, no passes.
//   src.issue(xg0, yg0, valid), src.reg(slot) = the fetched value of a (compile-time) slot.
// Returns sum and count; the dst value is sum / count, or 0 when count is 0 (Source.cpp:905).
// SCAN: nothing is fetched; the return value says whether a centre lies within QuadConsts::margin of an edge, in which
// case the double-precision pass (with the reference's ray cast at knife edges) owns this pixel.
// (partI, partJ): part of a wide window (quad_fast_parts), WIN x WIN positions from (partI, partJ) * WIN on; the caller adds the parts
template <typename F, int WIN, bool SCAN, typename Src>
AAI_HD bool quad_fast_pixel(const QuadConsts<F> &q, int Xc, int Yc, double dfx, double dfy, int mW, int mH, Src &src, F &sum, int &count,
                            int partI = 0, int partJ = 0)
{
    typedef typename QuadMask<WIN>::type mask_t;
    const F fpx = (F)dfx, fpy = (F)dfy;
    sum = F(0); count = 0;
    const F fi0 = ceil(fpx - q.hbf) + (F)(partI * WIN), fj0 = ceil(fpy - q.hbf) + (F)(partJ * WIN);      // first lattice point a centre of the square can be
    const int i0 = (int)fi0, j0 = (int)fj0;
    const int xg0 = Xc + i0, yg0 = Yc + j0;
    // Away from the image border every window of the wave lies inside the lattice: one vote replaces the per-column / per-row
    // validity and the clamps in the loads
    const bool interior = AAI_WAVE_ALL(xg0 >= 0 && xg0 + WIN <= mW && yg0 >= 0 && yg0 + WIN <= mH);
    // window columns ia..ib and rows ja..jb lie inside the lattice
    int ia = 0, ib = WIN - 1, ja = 0, jb = WIN - 1;
    if (!interior) {
        ia = xg0 < 0 ? -xg0 : 0; ib = (mW - 1 - xg0 < WIN - 1) ? mW - 1 - xg0 : WIN - 1;
        ja = yg0 < 0 ? -yg0 : 0; jb = (mH - 1 - yg0 < WIN - 1) ? mH - 1 - yg0 : WIN - 1;
        if (ia > ib || ja > jb) return false;
    }
    if (!SCAN) {
        mask_t valid = ~(mask_t)0;
        if (!interior) {
            const unsigned cols = (2u << ib) - (1u << ia);
            valid = 0;
#pragma unroll
            for (int j = 0; j < WIN; ++j)
                if (j >= ja && j <= jb) valid |= (mask_t)cols << (j * WIN);
        }
        src.issue(xg0, yg0, valid, interior);
    }
    // Columns and rows off the lattice get a coordinate far away: a = far * c (or far * s, or far * (c - s) with
    // b = far * (c + s) when both are off) fails the membership test by itself, c and s being > 1e-4 -- one select per
    // column and row instead of a validity test per position.
    const F far = F(1e30);
    F fis[WIN], fjs[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
        fis[i] = (i >= ia && i <= ib) ? fi0 + (F)i : far;
        fjs[i] = (i >= ja && i <= jb) ? fj0 + (F)i : far;
    }
    const F ac = qfma(fpy, q.s, -(fpx * q.c)), bc = -qfma(fpx, q.s, fpy * q.c);
    bool uncertain = false;
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
        F rowA, rowB;
        qfma2(-fjs[j], fjs[j], q.s, q.c, ac, bc, rowA, rowB);
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            F sa, sb;
            qfma2(fis[i], fis[i], q.c, q.s, rowA, rowB, sa, sb);
            const F a = qabs(sa), b = qabs(sb);
            const bool in = a <= q.h && b <= q.h;
            if (SCAN) {
                if ((qabs(a - q.h) < q.margin && b < q.h + q.margin) || (qabs(b - q.h) < q.margin && a < q.h + q.margin)) uncertain = true;
            } else {
                sum += in ? (F)src.reg(j * WIN + i) : F(0);       // (a select, not a product: values outside the square may be anything)
                count += in ? 1 : 0;
            }
        }
    }
    return uncertain;
}

// run-time window size and precision switch -> the matching instantiation
template <typename F, bool SCAN, int NC, typename Src>
AAI_HD bool quad_pixel_any(const QuadConsts<F> &q, int Xc, int Yc, double fpx, double fpy, int mW, int mH, Src &src, F &sumA, F (&sumVA)[NC])
{
#define AAI_QUAD_CASE(W) case W: return q.hiPrec ? quad_pixel<F, W, SCAN, true, NC>(q, Xc, Yc, fpx, fpy, mW, mH, src, sumA, sumVA) \
                                                 : quad_pixel<F, W, SCAN, false, NC>(q, Xc, Yc, fpx, fpy, mW, mH, src, sumA, sumVA);
    switch (q.win) {
    AAI_QUAD_CASE(3) AAI_QUAD_CASE(4) AAI_QUAD_CASE(5) AAI_QUAD_CASE(6) AAI_QUAD_CASE(7)
    default: return q.hiPrec ? quad_pixel<F, 8, SCAN, true, NC>(q, Xc, Yc, fpx, fpy, mW, mH, src, sumA, sumVA)
                             : quad_pixel<F, 8, SCAN, false, NC>(q, Xc, Yc, fpx, fpy, mW, mH, src, sumA, sumVA);
    }
#undef AAI_QUAD_CASE
}

}  // namespace aai
