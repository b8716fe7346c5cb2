// aai_rot_math.hpp -- per-pair geometry of the rotated-lattice kernels (K2/K3), shared between the HIP
// kernels (aai_rotated.hip) and the host-side emulation that the CPU test-suite uses to check this math
// against the golden vectors without a GPU (tests/host_emulation.cpp).  The functions are pure and
// double-precision; AAI_HD makes them __host__ __device__ under hipcc and plain inline under g++.
//
// What is computed and why: see the header comment of aai_rotated.hip; reference citations:
// Source.cpp:413-579 (loop), 962-1431 (classifier + area table), SURVEY.md Appendix A/B.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>

#include "../../include/aai.h"

#if defined(__HIPCC__)
#define AAI_HD __host__ __device__ __forceinline__
#else
#define AAI_HD inline
#endif

namespace aai {

// Geometry block handed to the per-output-pixel kernels, all double.
struct RotLaunch {
    double fracX, fracY, side, isoX, isoY, offX, offY, sn, cs;
    double reach;        // side*sqrt(2)/2 + 1, the reference's search-window half width (Source.cpp:426-429)
    int dW, dH, mW, mH, W, H, scale, quadrant;
    int mode, policy;
};

struct Frame {                      // per-dst-pixel constants, all in virtual-source units
    double px, py;                  // centre
    double c, s, h;                 // cos, sin of the reduced angle; half side
    double o0x, o0y, o1x, o1y;      // v0 = P + o0, v1 = P + o1, v2 = P - o1, v3 = P - o0
    double m1, im1;                 // dx/dy of the left/right edges (s/c) and its reciprocal
    double m2, im2;                 // dx/dy of the top/bottom edges (-c/s) and its reciprocal
    double rLc, rLs;                // 1/(L c), 1/(L s)
    double Lc, Ls;
};

AAI_HD double clamp01(double v) { return fmin(fmax(v, 0.0), 1.0); }

// integral over eta in [e0,e1] intersect [0,1] of clamp01(x(eta)), x(eta) = x0 + (eta-e0)*m
template <bool INCREASING>
AAI_HD double clamp_integral(double e0, double e1, double x0, double m, double im)
{
    const double a = fmax(e0, 0.0), b = fmin(e1, 1.0);
    if (!(a < b)) return 0.0;
    const double t0 = e0 - x0 * im;           // eta where x == 0
    const double t1 = e0 + (1.0 - x0) * im;   // eta where x == 1
    double ones, p, q;
    if (INCREASING) { ones = fmax(b - fmax(a, t1), 0.0); p = fmax(a, t0); q = fmin(b, t1); }
    else            { ones = fmax(fmin(b, t1) - a, 0.0); p = fmax(a, t1); q = fmin(b, t0); }
    double ramp = 0.0;
    if (q > p) {
        const double xp = clamp01(x0 + (p - e0) * m), xq = clamp01(x0 + (q - e0) * m);
        ramp = (q - p) * 0.5 * (xp + xq);
    }
    return ones + ramp;
}

// Overlap area of the dst square with the unit source pixel whose top-left corner is the local origin.
// lx,ly = dst centre in local coordinates.  policy REFERENCE applies the Appendix-B.2 substitution.
AAI_HD double pair_area(const Frame &f, double lx, double ly, int policy)
{
    const double v0x = lx + f.o0x, v0y = ly + f.o0y;
    const double v1x = lx + f.o1x, v1y = ly + f.o1y;
    const double v2x = lx - f.o1x, v2y = ly - f.o1y;
    const double v3x = lx - f.o0x, v3y = ly - f.o0y;

    // exact area: right boundary (v1 -> v3 -> v2) minus left boundary (v1 -> v0 -> v2), clamped to [0,1]
    double area = clamp_integral<true>(v1y, v3y, v1x, f.m1, f.im1)     // right edge
                + clamp_integral<false>(v3y, v2y, v3x, f.m2, f.im2)    // bottom edge
                - clamp_integral<false>(v1y, v0y, v1x, f.m2, f.im2)    // top edge
                - clamp_integral<true>(v0y, v2y, v0x, f.m1, f.im1);    // left edge
    area = clamp01(area);
    if (policy != AAI_POLICY_REFERENCE) return area;

    // Which dst edges (as segments) pass through the pixel?  Slab clip of A + t*D, t in [0,1].
    // top/bottom edges run along (c,-s): they enter through the left or bottom side.
    {
        const double txa = -v0x * f.rLc, txb = (1.0 - v0x) * f.rLc;
        const double tya = (v0y - 1.0) * f.rLs, tyb = v0y * f.rLs;
        if (fmax(fmax(txa, tya), 0.0) < fmin(fmin(txb, tyb), 1.0)) return area;   // top edge crosses
    }
    {
        const double txa = -v2x * f.rLc, txb = (1.0 - v2x) * f.rLc;
        const double tya = (v2y - 1.0) * f.rLs, tyb = v2y * f.rLs;
        if (fmax(fmax(txa, tya), 0.0) < fmin(fmin(txb, tyb), 1.0)) return area;   // bottom edge crosses
    }
    // left/right edges run along (s,c): they enter through the top or left side and leave through the
    // bottom or right side.  At most one of them can reach the pixel (they are L > sqrt 2 apart).
    double ax = v0x, ay = v0y;
    bool isLeft = true;
    {
        const double gl = fabs((0.5 - v0x) * f.c - (0.5 - v0y) * f.s);   // distance of the pixel centre to the left edge line
        const double gr = fabs((0.5 - v1x) * f.c - (0.5 - v1y) * f.s);
        if (gr < gl) { ax = v1x; ay = v1y; isLeft = false; }
    }
    const double txa = -ax * f.rLs, txb = (1.0 - ax) * f.rLs;      // x = 0, x = 1
    const double tya = -ay * f.rLc, tyb = (1.0 - ay) * f.rLc;      // y = 0, y = 1
    const double tin = fmax(txa, tya), tout = fmin(txb, tyb);
    if (!(tin < tout) || !(tin > 0.0) || !(tout < 1.0)) return area;   // misses, or a dst vertex lies inside
    const bool inTop = tya > txa, outRight = txb < tyb;
    if (inTop != outRight) return area;                            // opposite sides: a straight cut, exact
    double tri;
    if (inTop) {   // cuts the top-right corner: reference legs xa and 1-yb
        const double xin = ax + tin * f.Ls, yout = ay + tout * f.Lc;
        tri = 0.5 * xin * (1.0 - yout);
    } else {       // cuts the bottom-left corner: reference legs 1-xb and ya
        const double yin = ay + tin * f.Lc, xout = ax + tout * f.Ls;
        tri = 0.5 * (1.0 - xout) * yin;
    }
    // left edge + top-right corner, or right edge + bottom-left corner: the corner is the inside part
    return (isLeft == inTop) ? tri : 1.0 - tri;
}

// virtual pixel (X,Y) -> element offset in the original image (Source.cpp:164-167)
AAI_HD int64_t virt_offset(const RotLaunch &r, int X, int Y, int64_t rowStride)
{
    int sx, sy;
    switch (r.quadrant) {
    default:
    case 0: sx = X;            sy = Y;            break;
    case 1: sx = Y;            sy = r.mW - 1 - X; break;
    case 2: sx = r.mW - 1 - X; sy = r.mH - 1 - Y; break;
    case 3: sx = r.mH - 1 - Y; sy = X;            break;
    }
    if (r.scale > 1) { sx /= r.scale; sy /= r.scale; }
    return (int64_t)sy * rowStride + sx;
}

AAI_HD void pixel_centre(const RotLaunch &r, int dx, int dy, double &px, double &py)
{
    const double u = (dx + r.fracX) * r.side - r.isoX + r.offX;
    const double v = (dy + r.fracY) * r.side - r.isoY + r.offY;
    px = u * r.cs + v * r.sn + r.isoX;
    py = -u * r.sn + v * r.cs + r.isoY;
}

// Fills the per-dst-pixel constants for the area path.
AAI_HD void frame_init(Frame &f, const RotLaunch &r)
{
    f.c = r.cs; f.s = r.sn; f.h = 0.5 * r.side;
    f.o0x = -f.h * (f.c + f.s); f.o0y = f.h * (f.s - f.c);
    f.o1x = f.h * (f.c - f.s);  f.o1y = -f.h * (f.s + f.c);
    f.m1 = f.s / f.c;  f.im1 = f.c / f.s;
    f.m2 = -f.c / f.s; f.im2 = -f.s / f.c;
    f.Lc = r.side * f.c; f.Ls = r.side * f.s;
    f.rLc = 1.0 / f.Lc; f.rLs = 1.0 / f.Ls;
}

}  // namespace aai
