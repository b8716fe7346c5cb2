// aai_axis_verify.hpp -- does K1's separable model hold for this dst pixel?
//
// At reduced angle 0 the overlap of a dst pixel with a source pixel factors into (x overlap) * (y overlap), and K1
// (aai_axis.hip) is built on that.  The reference, however, reaches the same areas through its general classifier
// (Source.cpp:986-1431), whose DBL_EPSILON decisions give other answers in a few exactly-aligned situations -- a dst
// vertex on the midpoint of a pixel side (an x edge through pixel centres while the y edge runs along a pixel boundary)
// makes it return the whole pixel where half of it is covered.  Such pixels cannot be written as a product and
// change the normalisation of every weight of their dst pixel.
//
// So, once per geometry, the plan compares the two models pair by pair for every dst pixel that has a knife edge at all
// (policy REFERENCE, area mode): the reference's side is exactly what the fix-up pass (aai_rotated_kernel<AREA, STRICT>)
// computes, the other side is the product of the two clipped extents.  Dst pixels where any pair differs are recomputed by
// that fix-up pass behind K1; at 8192^2 -> 2048^2 (every edge on a pixel boundary, every vertex on a pixel corner)
// none differs.  Shared by the plan-time scan kernel and the CPU replay of the test-suite.
#pragma once
#include "aai_rot_math.hpp"
#include "aai_strict.hpp"

namespace aai {

AAI_HD bool axis_pixel_differs(const RotLaunch &r, int dx, int dy)
{
    double px, py;
    pixel_centre(r, dx, dy, px, py);
    if (!pixel_on_knife_edge(r, px, py, true)) return false;       // generic pixels: the closed forms ARE the products
    const double hb = r.h * (r.c + r.s);
    const int x0 = (int)fmax(0.0, floor(px - hb + 0.5 - AAI_KNIFE_GUARD)), x1 = (int)fmin((double)(r.mW - 1), ceil(px + hb - 0.5 + AAI_KNIFE_GUARD));
    const int y0 = (int)fmax(0.0, floor(py - hb + 0.5 - AAI_KNIFE_GUARD)), y1 = (int)fmin((double)(r.mH - 1), ceil(py + hb - 0.5 + AAI_KNIFE_GUARD));
    SVec sv4[4];
    bool haveVertices = false;
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
        }
    }
    return false;
}

}  // namespace aai
