// aai_strict.hpp -- the slow, reference-faithful evaluation of ONE (dst pixel, source pixel) pair, used
// by the rotated-lattice kernels only when the fast path finds a decision sitting on a knife edge.
//
// Why it exists.  The reference decides every pair with segment/segment parameters compared against
// +-DBL_EPSILON (Source.cpp:986-1034), followed by end-point rules (Source.cpp:330-342, 500-564) and a
// 10-type area table (Source.cpp:1035-1431).  When a dst edge passes (in exact arithmetic) through a
// source-pixel corner, or a dst vertex sits on a pixel side -- reduced angles like 30/45/60 degrees with
// commensurate sizes -- the outcome depends on the last bits of the reference's own operation order, and
// its area is discontinuous there.  The fast path (aai_rot_math.hpp) cannot
// agree with that by construction, so it measures how close each of its own decisions is to a threshold
// and hands pairs within kGuard of one to this file, which replays the reference's arithmetic operation by
// operation (same operand order, no fused multiply-adds, IEEE division) and therefore reproduces its
// decisions.  In generic geometry (all BASELINE configurations) no pair ever comes here.
//
// Shared by the HIP kernels and the host-side emulation used by the CPU tests (AAI_HD).
#pragma once

#include "aai_rot_math.hpp"

namespace aai {

struct SVec { double x, y; };
struct SLine { double a, b, c; };      // a*x + b*y + c = 0

#if defined(__clang__)
#define AAI_STRICT_FP _Pragma("clang fp contract(off)")
#else
#define AAI_STRICT_FP
#endif

// Source.cpp:212-219
AAI_HD SVec strict_centre(const RotLaunch &r, unsigned dx, unsigned dy)
{
    AAI_STRICT_FP
    const double u = (dx + r.fracX) * r.side - r.isoX + r.offX;
    const double v = (dy + r.fracY) * r.side - r.isoY + r.offY;
    SVec p;
    p.x = u * r.cs + v * r.sn + r.isoX;
    p.y = -u * r.sn + v * r.cs + r.isoY;
    return p;
}

// Source.cpp:243-273: "horizontal" edge line k of the dst lattice (k = 0..dH)
AAI_HD SLine strict_edge_row(const RotLaunch &g, unsigned k)
{
    AAI_STRICT_FP
    const double h = g.side / 2;
    SLine l;
    const bool last = k >= (unsigned)g.dH;
    const SVec p = strict_centre(g, 0, last ? (unsigned)g.dH - 1 : k);
    if (g.lt45) {
        l.a = g.ttn; l.b = 1;
        if (!last) l.c = -l.a * (p.x - h * (g.tcs + g.tsn)) - (p.y - h * (g.tcs - g.tsn));
        else       l.c = -l.a * (p.x - h * (g.tcs - g.tsn)) - (p.y + h * (g.tcs + g.tsn));
    } else {
        l.a = 1; l.b = -g.ttn;
        if (!last) l.c = -(p.x - h * (g.tcs + g.tsn)) - l.b * (p.y - h * (g.tcs - g.tsn));
        else       l.c = -(p.x + h * (g.tcs - g.tsn)) - l.b * (p.y - h * (g.tcs + g.tsn));
    }
    return l;
}

// Source.cpp:275-305: "vertical" edge line k (k = 0..dW)
AAI_HD SLine strict_edge_col(const RotLaunch &g, unsigned k)
{
    AAI_STRICT_FP
    const double h = g.side / 2;
    SLine l;
    const bool last = k >= (unsigned)g.dW;
    const SVec p = strict_centre(g, last ? (unsigned)g.dW - 1 : k, 0);
    if (g.lt45) {
        l.a = 1; l.b = -g.ttn;
        if (!last) l.c = -(p.x - h * (g.tcs + g.tsn)) - l.b * (p.y - h * (g.tcs - g.tsn));
        else       l.c = -(p.x + h * (g.tcs - g.tsn)) - l.b * (p.y - h * (g.tcs + g.tsn));
    } else {
        l.a = g.ttn; l.b = 1;
        if (!last) l.c = -l.a * (p.x - h * (g.tcs - g.tsn)) - (p.y + h * (g.tcs + g.tsn));
        else       l.c = -l.a * (p.x - h * (g.tcs + g.tsn)) - (p.y - h * (g.tcs - g.tsn));
    }
    return l;
}

// Source.cpp:962-985 (the `/ a2*b1` of line 978 kept as written).  Returns false without touching p when
// the reference bails out; that never happens for the perpendicular lattice lines used here.
AAI_HD bool strict_meet(SLine l1, SLine l2, SVec &p)
{
    AAI_STRICT_FP
    const double E = DBL_EPSILON;
    if ((fabs(l1.a) <= E && fabs(l1.b) <= E) || (fabs(l2.a) <= E && fabs(l2.b) <= E)) return false;
    if (fabs(l1.b) <= E && fabs(l2.b) <= E) return false;
    if (fabs(l1.a) <= E && fabs(l2.a) <= E) return false;
    const double det = l2.a * l1.b - l1.a * l2.b;
    if (fabs(det) <= E) return false;
    if (fabs(l2.b) <= E) {
        p.x = -l2.c / l2.a;
        p.y = (l1.a * l2.c - l2.a * l1.c) / l2.a * l1.b;
    } else {
        p.x = (l2.b * l1.c - l1.b * l2.c) / det;
        p.y = (l1.a * l2.c - l2.a * l1.c) / det;
    }
    return true;
}

// The four vertices of dst pixel (dx,dy) exactly as the reference obtains them (Source.cpp:419-422):
// v[0] top-left, v[1] top-right, v[2] bottom-left, v[3] bottom-right in the dst frame.
AAI_HD void strict_vertices(const RotLaunch &r, int dx, int dy, SVec v[4])
{
    const SLine top = strict_edge_row(r, (unsigned)dy), bot = strict_edge_row(r, (unsigned)dy + 1);
    const SLine lft = strict_edge_col(r, (unsigned)dx), rgt = strict_edge_col(r, (unsigned)dx + 1);
    v[0].x = v[0].y = v[1].x = v[1].y = v[2].x = v[2].y = v[3].x = v[3].y = 0.0;
    strict_meet(top, lft, v[0]);
    strict_meet(top, rgt, v[1]);
    strict_meet(bot, lft, v[2]);
    strict_meet(bot, rgt, v[3]);
}

// Source.cpp:986-1034.  Codes: 1 parallel, 2 overlapping, 3 interior crossing, 4 end-point crossing,
// 5 crossing outside the segments.  rr/ss are written only for codes 3-5.
AAI_HD int strict_seg(SVec p1, SVec p2, double &rr, SVec q1, SVec q2, double &ss)
{
    AAI_STRICT_FP
    const double E = DBL_EPSILON;
    const double den = (p2.x - p1.x) * (q2.y - q1.y) - (p2.y - p1.y) * (q2.x - q1.x);
    const double rn = (q1.x - p1.x) * (q2.y - q1.y) - (q1.y - p1.y) * (q2.x - q1.x);
    const double sn = (p2.y - p1.y) * (q1.x - p1.x) - (p2.x - p1.x) * (q1.y - p1.y);
    if (fabs(den) <= E && fabs(rn) <= E && fabs(sn) <= E) return 2;
    if (fabs(den) <= E) return 1;
    rr = rn / den;
    ss = sn / den;
    if (-E <= rr && rr <= 1.0 + E && -E <= ss && ss <= 1.0 + E) {
        if (fabs(rr) <= E || fabs(rr - 1.0) <= E || fabs(ss) <= E || fabs(ss - 1.0) <= E) return 4;
        return 3;
    }
    return 5;
}

// Source.cpp:368-398 / 837-864: is the source pixel centre inside the dst pixel (four axis rays).
AAI_HD bool strict_centre_inside(SVec c, const SVec v[4])
{
    AAI_STRICT_FP
    const double E = DBL_EPSILON;
    double rr = 0.0, ss = 0.0;               // persist across the 16 tests like tmpr/tmps
    for (int d = 0; d < 4; ++d) {
        SVec far;
        far.x = c.x + (d == 2 ? -100 : (d == 3 ? 100 : 0));
        far.y = c.y + (d == 0 ? -100 : (d == 1 ? 100 : 0));
        int hits = 0;
        for (int i = 0; i < 4; ++i) {
            // clockwise vertex order 0,1,3,2 (Source.cpp:377)
            const int i0 = i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? 3 : 2));
            const int i1 = i == 0 ? 1 : (i == 1 ? 3 : (i == 2 ? 2 : 0));
            strict_seg(c, far, rr, v[i0], v[i1], ss);
            if (-E < rr && -E < ss && ss < 1 + E) ++hits;
        }
        if (!hits) return false;
    }
    return true;
}

AAI_HD void strict_sort(double *a, int n)
{
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && a[j - 1] > a[j]; --j) { const double t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; }
}

AAI_HD void strict_erase(double *a, int &n, int k)
{
    for (int i = k; i + 1 < n; ++i) a[i] = a[i + 1];
    --n;
}

// The reference's overlap area for virtual source pixel (X,Y) and the dst pixel with vertices dv
// (Source.cpp:432-572 + 1035-1431).  policy EXACT swaps in the true corner-triangle legs (types 2/4).
AAI_HD double strict_pair_area(const SVec dv[4], int X, int Y, int policy)
{
    AAI_STRICT_FP
    const double E = DBL_EPSILON;
    SVec sv[4];
    sv[0].x = X - 0.5; sv[0].y = Y - 0.5;
    sv[1].x = X + 0.5; sv[1].y = Y - 0.5;
    sv[2].x = X - 0.5; sv[2].y = Y + 0.5;
    sv[3].x = X + 0.5; sv[3].y = Y + 0.5;

    // side lists as four fixed arrays (kept as separate scalars-of-arrays so they stay in registers)
    double xa[4], ya[4], yb[4], xb[4];
    int nxa = 0, nya = 0, nyb = 0, nxb = 0;
    double r4[4] = {0, 0, 0, 0}, s4[4] = {0, 0, 0, 0};

#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // dst edges in the reference's order: top (0,1), bottom (2,3), left (0,2), right (1,3)
        const SVec e0 = dv[e == 0 ? 0 : (e == 1 ? 2 : (e == 2 ? 0 : 1))];
        const SVec e1 = dv[e == 0 ? 1 : (e == 1 ? 3 : (e == 2 ? 2 : 3))];
        int code[4];
        code[0] = strict_seg(e0, e1, r4[0], sv[0], sv[1], s4[0]);
        code[1] = strict_seg(e0, e1, r4[1], sv[0], sv[2], s4[1]);
        code[2] = strict_seg(e0, e1, r4[2], sv[1], sv[3], s4[2]);
        code[3] = strict_seg(e0, e1, r4[3], sv[2], sv[3], s4[3]);
        bool drop = false;                    // a lone end-point touch discards the whole edge (330-342)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (code[i] != 4 || drop) continue;
            bool other = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j != i && (code[j] == 3 || code[j] == 4)) other = true;
            if (!other) drop = true;
        }
        if (drop) continue;
        if (code[0] == 3 || code[0] == 4) { if (nxa < 4) xa[nxa++] = s4[0]; }
        if (code[1] == 3 || code[1] == 4) { if (nya < 4) ya[nya++] = s4[1]; }
        if (code[2] == 3 || code[2] == 4) { if (nyb < 4) yb[nyb++] = s4[2]; }
        if (code[3] == 3 || code[3] == 4) { if (nxb < 4) xb[nxb++] = s4[3]; }
    }

    SVec pc; pc.x = X; pc.y = Y;
    const bool cin = strict_centre_inside(pc, dv);

    bool vin = false;
    double vx = -1, vy = -1;
    for (int i = 0; i < 4; ++i) {             // Source.cpp:401-408, last match wins
        if (sv[0].x + E < dv[i].x && dv[i].x < sv[1].x - E && sv[0].y + E < dv[i].y && dv[i].y < sv[2].y - E) {
            vin = true;
            vx = dv[i].x - sv[0].x;
            vy = dv[i].y - sv[0].y;
        }
    }

    // end-point clean-up (Source.cpp:496-564)
    strict_sort(xa, nxa); strict_sort(ya, nya); strict_sort(yb, nyb); strict_sort(xb, nxb);
    auto hasLow = [&](const double *a, int n) { for (int i = 0; i < n; ++i) if (a[i] <= E) return true; return false; };
    auto hasHigh = [&](const double *a, int n) { for (int i = 0; i < n; ++i) if (1 - a[i] <= E) return true; return false; };
    for (int i = 0; i < nya;) {
        if (ya[i] <= E)          { if (!hasLow(xa, nxa)) strict_erase(ya, nya, i); else ++i; }
        else if (1 - ya[i] <= E) { if (!hasLow(xb, nxb)) strict_erase(ya, nya, i); else ++i; }
        else ++i;
    }
    for (int i = 0; i < nyb;) {
        if (yb[i] <= E)          { if (!hasHigh(xa, nxa)) strict_erase(yb, nyb, i); else ++i; }
        else if (1 - yb[i] <= E) { if (!hasHigh(xb, nxb)) strict_erase(yb, nyb, i); else ++i; }
        else ++i;
    }
    for (int i = 0; i < nxa;) { if (xa[i] <= E || 1 - xa[i] <= E) strict_erase(xa, nxa, i); else ++i; }
    for (int i = 0; i < nxb;) { if (xb[i] <= E || 1 - xb[i] <= E) strict_erase(xb, nxb, i); else ++i; }

    const int nx = nxa + nxb, ny = nya + nyb;
    auto mn = [](double a, double b) { return a < b ? a : b; };
    auto mx = [](double a, double b) { return a < b ? b : a; };
    auto corner = [&]() {                     // types 2/4 triangle, Source.cpp:1055-1062
        double x, y;
        const bool top = nxa != 0, left = nya != 0;
        if (policy == AAI_POLICY_REFERENCE) {
            x = top ? xa[0] : 1 - xb[0];
            y = left ? ya[0] : 1 - yb[0];
        } else {
            const double cx = top ? xa[0] : xb[0], cy = left ? ya[0] : yb[0];
            x = left ? cx : 1 - cx;
            y = top ? cy : 1 - cy;
        }
        return 0.5 * x * y;
    };

    if (!vin) {
        if (nx == 0 && ny == 0) return cin ? 1 : 0;
        if (nx == 1 && ny == 1 && !cin) return corner();
        if ((nx == 2 && ny == 0) || (nx == 0 && ny == 2)) {
            double s1, s2;
            if (nxa != 0 && nxb != 0) { s1 = xa[0]; s2 = xb[0]; }
            else if (nya != 0 && nyb != 0) { s1 = ya[0]; s2 = yb[0]; }
            else return cin ? 1 : 0;
            const double t = 0.5 * (s1 + s2);
            return cin ? mx(t, 1 - t) : mn(t, 1 - t);
        }
        if (nx == 1 && ny == 1 && cin) return 1 - corner();
        if ((nx == 3 && ny == 1) || (nx == 1 && ny == 3)) {
            double sb, lb, base, height;
            if (nx == 1 && ny == 3) {
                if (nxa == 0) {
                    if (nya == 1) { sb = ya[0]; lb = mn(yb[0], yb[1]); base = 1 - xb[0]; height = 1 - mx(yb[0], yb[1]); }
                    else          { sb = mn(ya[0], ya[1]); lb = yb[0]; base = xb[0]; height = 1 - mx(ya[0], ya[1]); }
                } else {
                    if (nya == 1) { sb = 1 - ya[0]; lb = 1 - mx(yb[0], yb[1]); base = 1 - xa[0]; height = mn(yb[0], yb[1]); }
                    else          { sb = 1 - mx(ya[0], ya[1]); lb = 1 - yb[0]; base = xa[0]; height = mn(ya[0], ya[1]); }
                }
            } else {
                if (nya == 0) {
                    if (nxa == 1) { sb = xa[0]; lb = mn(xb[0], xb[1]); base = 1 - mx(xb[0], xb[1]); height = 1 - yb[0]; }
                    else          { sb = xb[0]; lb = mn(xa[0], xa[1]); base = 1 - mx(xa[0], xa[1]); height = yb[0]; }
                } else {
                    if (nxa == 1) { sb = 1 - xa[0]; lb = 1 - mx(xb[0], xb[1]); base = mn(xb[0], xb[1]); height = 1 - ya[0]; }
                    else          { sb = 1 - xb[0]; lb = 1 - mx(xa[0], xa[1]); base = mn(xa[0], xa[1]); height = ya[0]; }
                }
            }
            const double trapezoid = 0.5 * (sb + lb);
            const double triangle = 0.5 * base * height;
            return 1 - trapezoid - triangle;
        }
        if (nx == 2 && ny == 2) {
            double t1 = 0, t2 = 0;
            if (nxa == 2)      { t1 = 0.5 * mn(xa[0], xa[1]) * ya[0];       t2 = 0.5 * (1 - mx(xa[0], xa[1])) * yb[0]; }
            else if (nxb == 2) { t1 = 0.5 * mn(xb[0], xb[1]) * (1 - ya[0]); t2 = 0.5 * (1 - mx(xb[0], xb[1])) * (1 - yb[0]); }
            else if (nya == 2) { t1 = 0.5 * xa[0] * mn(ya[0], ya[1]);       t2 = 0.5 * xb[0] * (1 - mx(ya[0], ya[1])); }
            else if (nyb == 2) { t1 = 0.5 * (1 - xa[0]) * mn(yb[0], yb[1]); t2 = 0.5 * (1 - xb[0]) * (1 - mx(yb[0], yb[1])); }
            return 1.0 - t1 - t2;
        }
        if (nx == 0 && ny == 1) return cin ? 1 : 0;
    } else {
        if ((nx == 2 && ny == 0) || (nx == 0 && ny == 2)) {
            if (nxa == 2 || nxb == 2 || nya == 2 || nyb == 2) {
                double base = 0, height = 0;   // map order xa, xb, ya, yb: the last side with two hits wins
                if (nxa == 2) { base = fabs(xa[0] - xa[1]); height = vy; }
                if (nxb == 2) { base = fabs(xb[0] - xb[1]); height = 1 - vy; }
                if (nya == 2) { base = fabs(ya[0] - ya[1]); height = vx; }
                if (nyb == 2) { base = fabs(yb[0] - yb[1]); height = 1 - vx; }
                return 0.5 * base * height;
            }
            double t1, t2, t3;
            if (nxa == 1 && nxb == 1) {
                if (mx(xa[0], xb[0]) <= vx) { t1 = 0.5 * xa[0] * vy; t2 = 0.5 * vx; t3 = 0.5 * xb[0] * (1 - vy); }
                else { t1 = 0.5 * (1 - xa[0]) * vy; t2 = 0.5 * (1 - vx); t3 = 0.5 * (1 - xb[0]) * (1 - vy); }
            } else {
                if (mx(ya[0], yb[0]) <= vy) { t1 = 0.5 * ya[0] * vx; t2 = 0.5 * vy; t3 = 0.5 * yb[0] * (1 - vx); }
                else { t1 = 0.5 * (1 - ya[0]) * vx; t2 = 0.5 * (1 - vy); t3 = 0.5 * (1 - yb[0]) * (1 - vx); }
            }
            return t1 + t2 + t3;
        }
        if (nx == 1 && ny == 1) {
            double t1, t2;
            if (nxa == 1 && nya == 1)      { t1 = 0.5 * xa[0] * vy;             t2 = 0.5 * ya[0] * vx; }
            else if (nxa == 1 && nyb == 1) { t1 = 0.5 * (1 - xa[0]) * vy;       t2 = 0.5 * yb[0] * (1 - vx); }
            else if (nxb == 1 && nya == 1) { t1 = 0.5 * xb[0] * (1 - vy);       t2 = 0.5 * (1 - ya[0]) * vx; }
            else                           { t1 = 0.5 * (1 - xb[0]) * (1 - vy); t2 = 0.5 * (1 - yb[0]) * (1 - vx); }
            return t1 + t2;
        }
    }
    return cin ? 1 : 0;
}

}  // namespace aai
