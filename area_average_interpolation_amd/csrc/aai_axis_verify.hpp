// aai_axis_verify.hpp -- does K1's separable model hold for this dst pixel?
//
// At reduced angle 0 the overlap of a dst pixel with a source pixel factors into (x overlap) * (y overlap), and K1
// (aai_axis.hip) is built on that.  The reference, however, reaches the same areas through its general classifier
// (Source.cpp:986-1431), whose DBL_EPSILON decisions give other answers in a few exactly-aligned situations -- a dst
// vertex on the midpoint of a pixel side (an x edge through pixel centres while the y edge runs along a pixel boundary)
// makes it return the whole pixel where half of it is covered.  Such pixels cannot be written as a product and
// change the normalisation of every weight of their dst pixel.
//
// Fast mode has the same problem with centres exactly ON a dst edge or vertex: the reference's ray cast is not the product
// of two closed intervals there.
//
// So, once per geometry, the plan compares the two models pair by pair for every dst pixel that has a knife edge at all
// (area mode under either policy -- they differ in the corner-triangle rule of a slanted edge only -- and fast mode):
// the reference's side is exactly what the fix-up pass (aai_rotated_kernel<..., STRICT>) computes, the other side is
// what K1's tables say (the product of the two clipped extents; two closed intervals).  Dst pixels where any pair
// differs, and dst pixels that only graze the lattice, are recomputed by that fix-up pass behind K1; at 8192^2 -> 2048^2
// (every edge on a pixel boundary, every vertex on a pixel corner) none differs.  Shared by the plan-time scan kernel
// and the CPU replay of the test-suite.
#pragma once
#include "aai_rot_math.hpp"
#include "aai_strict.hpp"

namespace aai {

// K1's fast-mode membership along one axis: is the centre of virtual pixel X inside the closed interval [lo, hi] of a
// dst pixel?  The reference decides with ray / edge parameters compared against +-DBL_EPSILON (Source.cpp:857): the
// parameter along a dst edge is s = (X - lo) / (hi - lo) and must satisfy -eps < s < 1 + eps; the ray parameter
// r = distance / 100 must satisfy r > -eps.  (Used by the table builder in aai_plan.cpp and by the scan below.)
AAI_HD bool axis_centre_inside(double lo, double hi, int X)
{
    const double s = (-100.0 * (lo - X)) / (100.0 * (hi - lo));
    if (!(-DBL_EPSILON < s && s < 1 + DBL_EPSILON)) return false;
    return (X - lo) / 100.0 > -DBL_EPSILON && (hi - X) / 100.0 > -DBL_EPSILON;
}

// the edges K1's tables give dst pixel (dx, dy) (aai_plan.cpp: edge_along_x / edge_along_y)
AAI_HD void axis_pixel_edges(const RotLaunch &r, int dx, int dy, double &lox, double &hix, double &loy, double &hiy)
{
    double px, py, qx, qy;
    pixel_centre(r, dx, 0, px, py);
    lox = px - r.h * (r.tcs + r.tsn);
    if (dx + 1 < r.dW) { pixel_centre(r, dx + 1, 0, qx, qy); hix = qx - r.h * (r.tcs + r.tsn); }
    else hix = px + r.h * (r.tcs - r.tsn);
    pixel_centre(r, 0, dy, px, py);
    loy = py - r.h * (r.tcs - r.tsn);
    if (dy + 1 < r.dH) { pixel_centre(r, 0, dy + 1, qx, qy); hiy = qy - r.h * (r.tcs - r.tsn); }
    else hiy = py + r.h * (r.tcs + r.tsn);
}

// area mode
AAI_HD bool axis_pixel_differs(const RotLaunch &r, int dx, int dy)
{
    double px, py;
    pixel_centre(r, dx, dy, px, py);
    if (!pixel_on_knife_edge(r, px, py, true)) return false;       // generic pixels: the closed forms ARE the products
    int x0, x1, y0, y1;
    rot_window(r, px, py, x0, x1, y0, y1);
    SVec sv4[4];
    bool haveVertices = false;
    double sumStrict = 0.0, sumProduct = 0.0;
    for (int Y = y0; Y <= y1; ++Y) {
        const double oy = fmax(0.0, fmin(py + r.h, Y + 0.5) - fmax(py - r.h, Y - 0.5));
        for (int X = x0; X <= x1; ++X) {
            // the fix-up pass's answer for this pair (aai_rotated_kernel.hpp, STRICT)
            const double ex = X - px, ey = Y - py;
            const double a = ex * r.c - ey * r.s, b = ex * r.s + ey * r.c;
            double d = 0.0, area = 0.0;
            bool edgy = false, edgy2 = false;
            const int cls = classify_pair<true>(r, a, b, d, edgy);
            if (cls != PAIR_OUTSIDE) {
                if (cls == PAIR_INSIDE) area = 1.0;
                else if (cls == PAIR_GENERAL) area = wedge_pair_area<true>(r, px - (X - 0.5), py - (Y - 0.5), a < 0.0, b < 0.0, r.policy, edgy2);
                else area = single_cut_area<true>(r, d, cls == PAIR_CUT_LR, r.policy, edgy2);
                if (edgy || edgy2) {
                    if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                    area = strict_pair_area(sv4, X, Y, r.policy);
                }
            }
            const double ox = fmax(0.0, fmin(px + r.h, X + 0.5) - fmax(px - r.h, X - 0.5));
            if (fabs(area - ox * oy) > 1e-9) return true;
            sumStrict += area; sumProduct += ox * oy;
        }
    }
    // A dst pixel that only grazes the lattice (total area ~1e-14 from an extent that should be zero): the reference still
    // divides one rounding residue by another (Source.cpp:577 asks for DBL_EPSILON < sum only) -- leave it to the replay.
    return (sumStrict > 0.0 || sumProduct > 0.0) && (sumStrict < 1e-6 || sumProduct < 1e-6);
}

// fast mode: the membership of every centre in the window, replay of the reference's ray cast (aai_rotated_kernel<fast,
// STRICT>) against the per-axis rule of K1's tables
AAI_HD bool axis_pixel_differs_fast(const RotLaunch &r, int dx, int dy)
{
    double px, py;
    pixel_centre(r, dx, dy, px, py);
    if (!pixel_on_knife_edge(r, px, py, false)) return false;
    double lox, hix, loy, hiy;
    axis_pixel_edges(r, dx, dy, lox, hix, loy, hiy);
    const bool span = hix > lox && hiy > loy;
    int x0, x1, y0, y1;
    rot_window(r, px, py, x0, x1, y0, y1);
    const double lim = r.h + DBL_EPSILON * r.side;
    SVec sv4[4];
    bool haveVertices = false;
    for (int Y = y0; Y <= y1; ++Y)
        for (int X = x0; X <= x1; ++X) {
            const double ex = X - px, ey = Y - py;
            const double a = fabs(ex * r.c - ey * r.s), b = fabs(ex * r.s + ey * r.c);
            bool in = a <= lim && b <= lim;
            const bool edgy = (fabs(a - r.h) < AAI_KNIFE_GUARD && b <= r.h + AAI_KNIFE_GUARD) || (fabs(b - r.h) < AAI_KNIFE_GUARD && a <= r.h + AAI_KNIFE_GUARD);
            if (edgy) {
                if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                SVec pc; pc.x = X; pc.y = Y;
                in = strict_centre_inside(pc, sv4);
            }
            const bool model = span && axis_centre_inside(lox, hix, X) && axis_centre_inside(loy, hiy, Y);
            if (in != model) return true;
        }
    return false;
}

}  // namespace aai
