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
#include <hip/hip_runtime.h>
#define AAI_HD __host__ __device__ __forceinline__
#else
#define AAI_HD inline
#endif

namespace aai {

// Geometry block handed to the per-output-pixel kernels.  Everything here is
// uniform over the launch and computed once on the host in double precision (make_rot_launch in
// aai_plan.cpp), so the divisions below never run on the device and the values live in scalar registers.
struct RotLaunch {
    // affine map dst pixel -> virtual source (Source.cpp:173-219)
    double fracX, fracY, side, isoX, isoY, offX, offY, sn, cs;
    double reach;        // side*sqrt(2)/2 + 1, the reference's search-window half width (Source.cpp:426-429)
    int dW, dH, mW, mH, W, H, scale, quadrant;
    int mode, policy;
    int dyBase, dyEnd;   // dst rows [dyBase, dyEnd) are computed (a band; the whole image by default); dyBase % 16 == 0
    int srcRow0;         // the source pointer addresses source row srcRow0 (band buffers hold only the rows needed)
    double invScale;     // 1/scale: virtual index -> original index is (int)((X + 0.5) * invScale), exact
    // the dst square in the reduced frame: c = cos, s = sin of the reduced angle (both > 0), h = L/2
    double c, s, h;
    double o0x, o0y, o1x, o1y;      // v0 = P + o0, v1 = P + o1, v2 = P - o1, v3 = P - o0
    double m1, im1;                 // dx/dy of the left/right edges (s/c) and its reciprocal
    double m2, im2;                 // dx/dy of the top/bottom edges (-c/s) and its reciprocal
    double Lc, Ls, rLc, rLs;        // L*c, L*s and their reciprocals
    double k;                       // (c+s)/2: half extent of a unit pixel along either dst axis
    double lo, hi;                  // min(c,s), max(c,s)
    double rc, rs, rhi, r2cs;       // 1/c, 1/s, 1/hi, 1/(2 c s)
    // the reference's edge-line parametrisation (Source.cpp:229-240), for the strict replay only
    int runs;                       // area mode: walk each source row as boundary / interior / boundary runs (large footprints)
    int quad;                       // area / fast mode: the fp32 quad formulation serves this geometry (aai_rot_quad.hpp)
    int cell;                       // area mode: ... and so does the cell formulation (aai_rot_cell.hpp), which then takes plain images
    int wide;                       // area mode, wide footprints: parts per axis of the fp32 window (quad_wide_parts: 2 or 4), 0 = none
    int chan;                       // interleaved channels per pixel (1 = a plain image): element = pixel offset * chan + channel
    int lt45;                       // reduced angle < 45 degrees
    double tsn, tcs, ttn;           // tmpSin, tmpCos, tmpTan (tan snapped to 0 below DBL_EPSILON)
    // samplers (K4/K5): the dst pixel centre in continuous ORIGINAL-image coordinates (pixel centres at integers) is
    // affine in (dx, dy): sx = sAx dx + sBx dy + sCx, sy alike -- pixel_centre, the quadrant's pre-rotation
    // (Source.cpp:164-167) and the 1/scale of the replication composed once on the host in double precision
    double sAx, sBx, sCx, sAy, sBy, sCy;
    // the fp32 window kernels (aai_rotated_quad.hip): the dst pixel centre in virtual-lattice coordinates as two fused
    // multiply-adds per coordinate, X = cXa dx + cXb dy + cX0, Y alike (quad_centre)
    double cXa, cXb, cX0, cYa, cYb, cY0;
    // request hints (include/aai.h): AAI_POLICY_PREFER_CELL, AAI_POLICY_DIAG_NO_FIXUP
    int preferCell, noFixup;
    int srcRow1;         // ... and the buffer ends before source row srcRow1 (the image height unless a row band)
};

// Decisions closer than this (in virtual-source pixels) to their threshold are "knife edges": the fast
// path still answers, but reports the pair so that the strict replay (aai_strict.hpp) can overrule it.
// fp64 noise on coordinates up to ~3e4 is ~1e-11, generic geometry keeps margins >> 1e-9.
#define AAI_KNIFE_GUARD 1e-9

AAI_HD double clamp01(double v) { return fmin(fmax(v, 0.0), 1.0); }

// Overlap area when exactly ONE edge line of the dst square cuts the unit pixel and the pixel lies inside
// the other three half-planes.  d = signed distance of the pixel centre from that edge (positive inside),
// |d| < k.  The cut part is a corner triangle, a trapezoid or the complement of a corner triangle; for a
// left/right edge (isLR) the reference takes the complementary legs for the two corner cases
// (Source.cpp:1055-1062), which policy REFERENCE reproduces.
template <bool KNIFE>
AAI_HD double single_cut_area(const RotLaunch &r, double d, bool isLR, int policy, bool &edgy)
{
    const double t = d + r.k;                 // how far the line has entered the pixel, in (0, 2k)
    // the line passes through a second pixel corner at t == lo and t == hi
    if (KNIFE) edgy = fabs(t - r.lo) < AAI_KNIFE_GUARD || fabs(t - r.hi) < AAI_KNIFE_GUARD;
    const bool substitute = isLR && policy == AAI_POLICY_REFERENCE;
    if (t <= r.lo) {                          // inside part = corner triangle with legs t/c, t/s
        if (substitute) return 0.5 * (1.0 - t * r.rc) * (1.0 - t * r.rs);
        return t * t * r.r2cs;
    }
    if (t < r.hi) return (t - 0.5 * r.lo) * r.rhi;     // trapezoid: exact for every edge
    const double u = 2.0 * r.k - t;           // outside part = corner triangle with legs u/c, u/s
    if (substitute) return 1.0 - 0.5 * (1.0 - u * r.rc) * (1.0 - u * r.rs);
    return 1.0 - u * u * r.r2cs;
}

// integral over eta in [e0,e1] intersect [0,1] of clamp01(x(eta)) for the line x(eta) = xr + (eta-er)*m
// through the reference point (xr, er); e0 may be -infinity and e1 +infinity (a ray from a vertex).
template <bool INCREASING>
AAI_HD double ray_integral(double e0, double e1, double xr, double er, double m, double im)
{
    const double a = fmax(e0, 0.0), b = fmin(e1, 1.0);
    if (!(a < b)) return 0.0;
    const double t0 = er - xr * im;           // eta where x == 0
    const double t1 = er + (1.0 - xr) * im;   // eta where x == 1
    double ones, p, q;
    if (INCREASING) { ones = fmax(b - fmax(a, t1), 0.0); p = fmax(a, t0); q = fmin(b, t1); }
    else            { ones = fmax(fmin(b, t1) - a, 0.0); p = fmax(a, t1); q = fmin(b, t0); }
    double ramp = 0.0;
    if (q > p) {
        const double xp = clamp01(xr + (p - er) * m), xq = clamp01(xr + (q - er) * m);
        ramp = (q - p) * 0.5 * (xp + xq);
    }
    return ones + ramp;
}

// Overlap area of the dst square with a unit source pixel near one of its vertices (lx, ly = dst centre relative to the
// pixel's top-left corner).  A unit pixel that is not wholly outside the square lies wholly inside
// the half-planes of the two FAR edges (h = L/2 exceeds the pixel's half extent k), so only the nearer
// left/right edge and the nearer top/bottom edge matter: the overlap is pixel (intersect) the quadrant those two
// edges span at their common vertex V, i.e. two ray integrals instead of four segment integrals, and the
// reference-policy test only has to look at those two edges.  nearLeft / nearTop say which edges are near
// (signs of the pixel centre's dst-frame coordinates a, b).
template <bool KNIFE>
AAI_HD double wedge_pair_area(const RotLaunch &f, double lx, double ly, bool nearLeft, bool nearTop, int policy, bool &edgy)
{
    const double inf = HUGE_VAL;
    if (KNIFE) edgy = false;
    // the near vertex: v0 = left/top, v1 = right/top, v2 = left/bottom, v3 = right/bottom
    const double ox = nearTop ? (nearLeft ? f.o0x : f.o1x) : (nearLeft ? -f.o1x : -f.o0x);
    const double oy = nearTop ? (nearLeft ? f.o0y : f.o1y) : (nearLeft ? -f.o1y : -f.o0y);
    const double vx = lx + ox, vy = ly + oy;
    // m1: dx/dy of the left/right edges (increasing), m2: of the top/bottom edges (decreasing)
    double area;
    if (nearTop) {
        if (nearLeft) area = 1.0 - ray_integral<false>(-inf, vy, vx, vy, f.m2, f.im2) - ray_integral<true>(vy, inf, vx, vy, f.m1, f.im1);
        else          area = ray_integral<true>(vy, inf, vx, vy, f.m1, f.im1) - ray_integral<false>(vy, inf, vx, vy, f.m2, f.im2);
    } else {
        if (nearLeft) area = ray_integral<false>(-inf, vy, vx, vy, f.m2, f.im2) - ray_integral<true>(-inf, vy, vx, vy, f.m1, f.im1);
        else          area = ray_integral<true>(-inf, vy, vx, vy, f.m1, f.im1) + ray_integral<false>(vy, inf, vx, vy, f.m2, f.im2);
    }
    area = clamp01(area);

    if (KNIFE) {   // per-pair knife tests, fix-up pass only
        const double g = AAI_KNIFE_GUARD, L = 2.0 * f.h;
        const double v0x = lx + f.o0x, v0y = ly + f.o0y, v1x = lx + f.o1x, v1y = ly + f.o1y;
        const double v2x = lx - f.o1x, v2y = ly - f.o1y, v3x = lx - f.o0x, v3y = ly - f.o0y;
        auto onSide = [&](double px, double py) {
            const double ex = fmin(fabs(px), fabs(px - 1.0)), ey = fmin(fabs(py), fabs(py - 1.0));
            const bool inx = px > -g && px < 1.0 + g, iny = py > -g && py < 1.0 + g;
            return (ex < g && iny) || (ey < g && inx);
        };
        if (onSide(v0x, v0y) || onSide(v1x, v1y) || onSide(v2x, v2y) || onSide(v3x, v3y)) edgy = true;
        const double dl0 = -v0x * f.c + v0y * f.s, dt0 = -v0x * f.s - v0y * f.c;
        auto onEdge = [&](double dl, double dt) {
            const double dr = L - dl, db = L - dt;
            return (fmin(fabs(dl), fabs(dr)) < g && dt > -g && db > -g) || (fmin(fabs(dt), fabs(db)) < g && dl > -g && dr > -g);
        };
        if (onEdge(dl0, dt0) || onEdge(dl0 + f.c, dt0 + f.s) || onEdge(dl0 - f.s, dt0 + f.c) || onEdge(dl0 + f.c - f.s, dt0 + f.s + f.c)) edgy = true;
    }
    if (policy != AAI_POLICY_REFERENCE) return area;

    // Reference policy: does the near top/bottom edge (as a segment) pass through the pixel?  It starts at
    // its LEFT vertex (v0 for the top edge, v2 for the bottom edge) and runs along (c,-s).
    {
        const double sx = lx + (nearTop ? f.o0x : -f.o1x), sy = ly + (nearTop ? f.o0y : -f.o1y);
        const double txa = -sx * f.rLc, txb = (1.0 - sx) * f.rLc;
        const double tya = (sy - 1.0) * f.rLs, tyb = sy * f.rLs;
        if (fmax(fmax(txa, tya), 0.0) < fmin(fmin(txb, tyb), 1.0)) return area;   // it crosses: two chords or a vertex, exact
    }
    // The near left/right edge starts at its TOP vertex (v0 / v1) and runs along (s,c): it enters through the
    // top or left side and leaves through the bottom or right side.
    const double ax = lx + (nearLeft ? f.o0x : f.o1x), ay = ly + (nearLeft ? f.o0y : f.o1y);
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
    return (nearLeft == inTop) ? tri : 1.0 - tri;
}

// virtual pixel (X,Y) -> element offset in the original image (Source.cpp:164-167)
// (pitch = elements per pixel: the channel count of an interleaved image; rowStride is in elements either way)
AAI_HD int64_t virt_offset(const RotLaunch &r, int X, int Y, int64_t rowStride, int pitch = 1)
{
    int sx, sy;
    switch (r.quadrant) {
    default:
    case 0: sx = X;            sy = Y;            break;
    case 1: sx = Y;            sy = r.mW - 1 - X; break;
    case 2: sx = r.mW - 1 - X; sy = r.mH - 1 - Y; break;
    case 3: sx = r.mH - 1 - Y; sy = X;            break;
    }
    if (r.scale > 1) {   // exact: (v + 0.5)/scale is at least 0.5/scale away from an integer
        sx = (int)((sx + 0.5) * r.invScale);
        sy = (int)((sy + 0.5) * r.invScale);
    }
    return (int64_t)(sy - r.srcRow0) * rowStride + (int64_t)sx * pitch;
}

// With scale == 1 the virtual lattice IS the source image seen through a quarter-turn.  A "line" is the set of virtual
// pixels that share one SOURCE row: virtual row Y = u in quadrants 0 and 2 (inner coordinate w = X), virtual column
// X = u in quadrants 1 and 3 (w = Y).  virt_line gives the element offset of that source row and whether w runs
// against source x; element w of the line is source column (rev ? n - 1 - w : w), n = mW (rows) or mH (columns) = W.
AAI_HD bool virt_lines_are_columns(const RotLaunch &r) { return (r.quadrant & 1) != 0; }
AAI_HD int64_t virt_line(const RotLaunch &r, int u, int64_t rowStride, bool &rev)
{
    int sy;
    switch (r.quadrant) {
    default:
    case 0: sy = u;            rev = false; break;
    case 1: sy = r.mW - 1 - u; rev = false; break;      // (X, Y) -> source (Y, mW-1-X)
    case 2: sy = r.mH - 1 - u; rev = true;  break;      // (X, Y) -> source (mW-1-X, mH-1-Y)
    case 3: sy = u;            rev = true;  break;      // (X, Y) -> source (mH-1-Y, X)
    }
    return (int64_t)(sy - r.srcRow0) * rowStride;
}

// Runs of one line inside the window range [w0, w1] of the dst square centred at (px, py); pIn is the centre's inner
// coordinate and fixedRel the line's outer coordinate relative to the centre (rows: px and Y - py; columns: py and
// X - px):
//   [t0, t1]  pixels the square can touch -- every pixel outside it is PAIR_OUTSIDE for classify_pair;
//   [i0, i1]  pixels wholly inside the square -- every one of them is PAIR_INSIDE (area exactly 1);
// i0 > i1 when the line has no interior pixel, t0 > t1 when it has none at all.  The pixel's half extent along both
// dst axes is k, so "touchable" is |a|, |b| < h + k and "inside" is |a|, |b| <= h - k with a = ex c - ey s,
// b = ex s + ey c: two intervals of the inner coordinate.  Twice the knife guard is taken off both so that rounding in
// the bounds (~1e-12) can never contradict classify_pair, which decides the pixels left in the two boundary runs
// exactly as before.
AAI_HD void line_runs(const RotLaunch &r, bool columns, double pIn, double fixedRel, int w0, int w1, int &t0, int &t1, int &i0, int &i1)
{
    const double g2 = 2.0 * AAI_KNIFE_GUARD;
    // rows:    ex in ((ey s -+ H) / c) and ((-ey c -+ H) / s);   columns: ey in ((ex c -+ H) / s) and ((-ex s -+ H) / c)
    const double e1 = fixedRel * (columns ? r.c : r.s), e2 = fixedRel * (columns ? r.s : r.c);
    const double q1 = columns ? r.rs : r.rc, q2 = columns ? r.rc : r.rs;
    const double ht = r.h + r.k + g2, hi = r.h - r.k - g2;
    const double lot = fmax((e1 - ht) * q1, (-e2 - ht) * q2), hit = fmin((e1 + ht) * q1, (-e2 + ht) * q2);
    // clamp in double before converting: near-axis rotations make the unconstrained bounds astronomically large
    const double ta = fmax(ceil(pIn + lot), (double)w0), tb = fmin(floor(pIn + hit), (double)w1);
    if (!(ta <= tb)) { t0 = 0; t1 = -1; i0 = 0; i1 = -1; return; }
    t0 = (int)ta; t1 = (int)tb;
    const double loi = fmax((e1 - hi) * q1, (-e2 - hi) * q2), hii = fmin((e1 + hi) * q1, (-e2 + hi) * q2);
    const double ia = fmax(ceil(pIn + loi), ta), ib = fmin(floor(pIn + hii), tb);
    if (hi > 0.0 && ia <= ib) { i0 = (int)ia; i1 = (int)ib; }
    else { i0 = t1 + 1; i1 = t1; }          // empty interior
}

// Fast mode: the pixel centres of one line (see virt_line) inside the closed dst square form one interval of the inner
// coordinate relative to the centre, [lo, hi] (two pairs of parallel edges = two interval constraints).
AAI_HD void centre_interval(const RotLaunch &r, bool columns, double fixedRel, double &lo, double &hi)
{
    const double e1 = fixedRel * (columns ? r.c : r.s), e2 = fixedRel * (columns ? r.s : r.c);
    const double q1 = columns ? r.rs : r.rc, q2 = columns ? r.rc : r.rs;
    lo = fmax((e1 - r.h) * q1, (-e2 - r.h) * q2);
    hi = fmin((e1 + r.h) * q1, (-e2 + r.h) * q2);
}

AAI_HD void pixel_centre(const RotLaunch &r, int dx, int dy, double &px, double &py)
{
    const double u = (dx + r.fracX) * r.side - r.isoX + r.offX;
    const double v = (dy + r.fracY) * r.side - r.isoY + r.offY;
    px = u * r.cs + v * r.sn + r.isoX;
    py = -u * r.sn + v * r.cs + r.isoY;
}

// The window of virtual pixels the double-precision kernels (aai_rotated_kernel, aai_rotated_runs_kernel, the strict fix-up pass,
// K1's model check) visit for a dst pixel centred at (px, py): the tight bounding box of its square, clipped to the lattice
// (the reference searches a wider window, Source.cpp:426-429, whose extra pixels all classify as "not included").  Everything
// those kernels FETCH lies in these rows and columns (their 16-byte segment loads stay inside a source row).
AAI_HD void rot_window(const RotLaunch &r, double px, double py, int &x0, int &x1, int &y0, int &y1)
{
    const double hb = r.h * (r.c + r.s);
    const double ax = floor(px - hb + 0.5 - AAI_KNIFE_GUARD), bx = ceil(px + hb - 0.5 + AAI_KNIFE_GUARD);
    const double ay = floor(py - hb + 0.5 - AAI_KNIFE_GUARD), by = ceil(py + hb - 0.5 + AAI_KNIFE_GUARD);
    x0 = (int)(ax > 0.0 ? ax : 0.0); x1 = (int)(bx < (double)(r.mW - 1) ? bx : (double)(r.mW - 1));
    y0 = (int)(ay > 0.0 ? ay : 0.0); y1 = (int)(by < (double)(r.mH - 1) ? by : (double)(r.mH - 1));
}
// ... and how far from the pixel's centre, along either lattice axis, that window reaches at most (rotated_band_source_rows)
AAI_HD double rot_window_reach(double h, double c, double s) { return h * (c + s) + 0.5 + 1e-6; }

// The same centre from host-composed coefficients: 4 fused multiply-adds instead of 14 additions and 6 products in double precision
// (a sixth of the fast-mode window kernel's issue time at config 5).  It differs from pixel_centre -- the reference's own operation
// order, which the double-precision kernels and the knife-edge scan keep -- by ~1e-12 virtual pixels; the kernels that use it work
// in fp32 relative to the nearest lattice point and leave every decision closer than 3 eps_coord(fp32) to its threshold to the
// double-precision pass, and their plan-time scans and CPU replays use this function too.
AAI_HD void quad_centre(const RotLaunch &r, int dx, int dy, double &px, double &py)
{
    const double x = (double)dx, y = (double)dy;
    px = fma(x, r.cXa, fma(y, r.cXb, r.cX0));
    py = fma(x, r.cYa, fma(y, r.cYb, r.cY0));
}

// The dst columns of rows [ya, yb] whose centres lie within rho virtual pixels of the lattice's extent -- a superset, lo > hi: none.
// Further out a dst pixel is 0 in every mode (its footprint reaches h (c + s) + 1/2 < rho - 1; a sample point is outside the image),
// which is 36 % of config 3's canvas and 50 % of config 5's.  Centres are affine in (dx, dy) (pixel_centre); reduced angle strictly
// inside (0, 90) degrees, else everything is reported live.
AAI_HD void rot_live_cols(const RotLaunch &r, int ya, int yb, double rho, int &lo, int &hi)
{
    const double Lc = r.side * r.cs, Ls = r.side * r.sn;
    lo = 0; hi = r.dW - 1;
    if (!(Lc > 1e-9 && Ls > 1e-9)) return;
    // centre of (dx, dy): (A0 + dx Lc + dy Ls, B0 - dx Ls + dy Lc)
    const double u0 = r.fracX * r.side - r.isoX + r.offX, v0 = r.fracY * r.side - r.isoY + r.offY;
    const double A0 = u0 * r.cs + v0 * r.sn + r.isoX, B0 = -u0 * r.sn + v0 * r.cs + r.isoY;
    const double xMax = (double)r.mW - 1.0 + rho, yMax = (double)r.mH - 1.0 + rho;
    double a = (-rho - A0 - yb * Ls) / Lc, b = (xMax - A0 - ya * Ls) / Lc;            // -rho <= X <= mW - 1 + rho for some row of the band
    a = fmax(a, (B0 + ya * Lc - yMax) / Ls); b = fmin(b, (B0 + yb * Lc + rho) / Ls);   // ... and the same for Y
    a = floor(a) - 1.0; b = ceil(b) + 1.0;
    if (a > 0.0) lo = a > 2147483000.0 ? 2147483000 : (int)a;
    if (b < (double)(r.dW - 1)) hi = b < -2147483000.0 ? -2147483000 : (int)b;
}

// How a source pixel relates to the dst square, from its centre's dst-frame coordinates (a along the
// top edge direction, b along the left edge direction; the square is |a| <= h, |b| <= h).
enum PairClass { PAIR_OUTSIDE = 0, PAIR_INSIDE = 1, PAIR_CUT_LR = 2, PAIR_CUT_TB = 3, PAIR_GENERAL = 4 };

template <bool KNIFE>
AAI_HD int classify_pair(const RotLaunch &r, double a, double b, double &d, bool &edgy)
{
    const double guard = AAI_KNIFE_GUARD;     // anything this close to a class boundary takes the general path
    const double ma = r.h - fabs(a), mb = r.h - fabs(b);     // inside-distance from the nearer L/R and T/B edge
    const double mn = fmin(ma, mb);
    if (KNIFE) edgy = false;
    if (mn <= -r.k - guard) return PAIR_OUTSIDE;
    // an edge line through a pixel corner: |m| == k
    if (KNIFE) edgy = fabs(fabs(ma) - r.k) < guard || fabs(fabs(mb) - r.k) < guard;
    if (mn >= r.k + guard) return PAIR_INSIDE;
    if (mb >= r.k + guard && fabs(ma) < r.k - guard) { d = ma; return PAIR_CUT_LR; }
    if (ma >= r.k + guard && fabs(mb) < r.k - guard) { d = mb; return PAIR_CUT_TB; }
    return PAIR_GENERAL;
}

// Does dst pixel with centre (px,py) have ANY knife edge -- a vertex on a pixel-boundary line, or an edge
// through a lattice point (pixel corners at half-integers for the area mode, pixel centres at integers for
// the fast mode)?  A superset of the per-pair tests above at twice their guard, evaluated once per dst pixel
// by the production pass (about 8 operations per lattice line an edge crosses).
AAI_HD bool pixel_on_knife_edge(const RotLaunch &r, double px, double py, bool areaMode)
{
    const double g2 = 2.0 * AAI_KNIFE_GUARD;
    const double off = areaMode ? 0.5 : 0.0;
    auto nearLattice = [&](double v) { const double w = v - off; return fabs(w - floor(w + 0.5)) < 2.0 * g2; };
    const double vx[4] = {px + r.o0x, px + r.o1x, px - r.o1x, px - r.o0x};
    const double vy[4] = {py + r.o0y, py + r.o1y, py - r.o1y, py - r.o0y};
    if (areaMode)
        for (int i = 0; i < 4; ++i)
            if (nearLattice(vx[i]) || nearLattice(vy[i])) return true;
    // left/right edges run along (s,c) from v0 / v1; top/bottom edges along (c,-s) from v0 / v2.  Walk the
    // lattice lines each edge crosses most squarely so that the slope used is at most 1 in magnitude.
    const bool steep = r.c >= r.s;            // left/right edges closer to vertical, top/bottom to horizontal
    for (int e = 0; e < 4; ++e) {
        const bool lr = e >= 2;
        const int ia = lr ? (e == 2 ? 0 : 1) : (e == 0 ? 0 : 2);         // start vertex: LR from v0, v1; TB from v0, v2
        const double ax = vx[ia], ay = vy[ia];
        const double ex = lr ? r.Ls : r.Lc, ey = lr ? r.Lc : -r.Ls;       // edge vector
        const bool alongY = lr ? steep : !steep;                          // iterate lattice lines y = j + off
        if (alongY) {
            const double lo = fmin(ay, ay + ey), hi = fmax(ay, ay + ey);
            const double slope = lr ? r.m1 : r.m2;                         // dx/dy
            for (double y = ceil(lo - off - g2) + off; y <= hi + g2; y += 1.0)
                if (nearLattice(ax + (y - ay) * slope)) return true;
        } else {
            const double lo = fmin(ax, ax + ex), hi = fmax(ax, ax + ex);
            const double slope = lr ? r.im1 : r.im2;                       // dy/dx
            for (double x = ceil(lo - off - g2) + off; x <= hi + g2; x += 1.0)
                if (nearLattice(ay + (x - ax) * slope)) return true;
        }
    }
    return false;
}

}  // namespace aai
