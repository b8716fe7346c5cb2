// TEST INFRASTRUCTURE ONLY -- host-side emulation of the device kernels' arithmetic.
//
// Built by tests/conftest.py with plain g++ (no HIP) into tests/_build/libaai_hostemu.so.  It reuses the
// PRODUCT's host planner (csrc/aai_plan.cpp: geometry, separable tables, strips) and the PRODUCT's
// per-pair math header (csrc/aai_rot_math.hpp) and replays the loop structure of aai_axis.hip /
// aai_rotated.hip serially, so that the CPU test-suite can check tables, strip partitioning, output
// addressing and the clip/substitution math against the golden vectors in a container without a GPU.
// It is not part of the package, is never loaded by it, and is not a fallback for anything.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../area_average_interpolation_amd/csrc/aai_plan.cpp"
#include "../../area_average_interpolation_amd/csrc/aai_rot_math.hpp"
#include "../../area_average_interpolation_amd/csrc/aai_strict.hpp"
#include "../../area_average_interpolation_amd/csrc/aai_rot_quad.hpp"
#include "../../area_average_interpolation_amd/csrc/aai_rot_cell.hpp"
#include "../../area_average_interpolation_amd/csrc/aai_axis_verify.hpp"

using namespace aai;

// ---- the four-edge form of the pair area: a cross-check of the product's closed forms, used by tests only ------------
namespace aai {
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

// Overlap area of the dst square with the unit source pixel whose top-left corner is the local origin,
// for any configuration (two edges, a dst vertex inside, ...).  lx,ly = dst centre in local coordinates.
// policy REFERENCE applies the Appendix-B.2 substitution when a lone left/right edge SEGMENT cuts a corner.
template <bool KNIFE>
AAI_HD double pair_area(const RotLaunch &f, double lx, double ly, int policy, bool &edgy)
{
    if (KNIFE) edgy = false;
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

    // Knife edges, where the reference's answer hangs on its DBL_EPSILON rules (Source.cpp:330-342, 401-408,
    // 500-564, 1430) and its area jumps: (a) a dst vertex on a side of the pixel, (b) a dst edge through a
    // corner of the pixel.  "On" = within AAI_KNIFE_GUARD.  (Per-pair form, used by the fix-up pass only; the
    // production pass tests the dst pixel as a whole with pixel_on_knife_edge below.)
    if (KNIFE) {
        const double g = AAI_KNIFE_GUARD, L = 2.0 * f.h;
        auto onSide = [&](double vx, double vy) {
            const double ex = fmin(fabs(vx), fabs(vx - 1.0)), ey = fmin(fabs(vy), fabs(vy - 1.0));
            const bool inx = vx > -g && vx < 1.0 + g, iny = vy > -g && vy < 1.0 + g;
            return (ex < g && iny) || (ey < g && inx);
        };
        if (onSide(v0x, v0y) || onSide(v1x, v1y) || onSide(v2x, v2y) || onSide(v3x, v3y)) edgy = true;
        // inside-distances of the pixel's top-left corner from the left and the top edge line; the other
        // corners and edges follow by adding c, s and subtracting from L
        const double dl0 = -v0x * f.c + v0y * f.s, dt0 = -v0x * f.s - v0y * f.c;
        auto onEdge = [&](double dl, double dt) {
            const double dr = L - dl, db = L - dt;
            return (fmin(fabs(dl), fabs(dr)) < g && dt > -g && db > -g) || (fmin(fabs(dt), fabs(db)) < g && dl > -g && dr > -g);
        };
        if (onEdge(dl0, dt0) || onEdge(dl0 + f.c, dt0 + f.s) || onEdge(dl0 - f.s, dt0 + f.c) || onEdge(dl0 + f.c - f.s, dt0 + f.s + f.c)) edgy = true;
    }

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
    if (policy != AAI_POLICY_REFERENCE) return area;
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

}  // namespace aai

static float row_w(const AxisEntry &e, int y) { return y == e.s0 ? e.wFirst : (y == e.s1 ? e.wLast : e.wMid); }

static void emu_axis(const Geometry &g, int mode, const float *src, int64_t srcStride, float *dst, int64_t dstStride)
{
    AxisTables t;
    build_axis_tables(g, mode, t);
    const int64_t sa = t.transposed ? dstStride : 1, sb = t.transposed ? 1 : dstStride;
    const int64_t strideA = t.flipA ? -sa : sa, strideB = t.flipB ? -sb : sb;
    const int64_t base = (t.flipA ? (int64_t)(t.nA - 1) * sa : 0) + (t.flipB ? (int64_t)(t.nB - 1) * sb : 0);
    if (t.wide) {
        for (int kb = 0; kb < t.nB; ++kb)
            for (int ka = 0; ka < t.nA; ++ka) {
                const AxisEntry &c = t.lane[ka], &e = t.row[kb];
                float acc = 0.f;
                for (int y = e.s0; y <= e.s1; ++y) {
                    float h = 0.f;
                    for (int x = c.s0; x <= c.s1; ++x) h += row_w(c, x) * src[(int64_t)y * srcStride + x];
                    acc += row_w(e, y) * h;
                }
                dst[base + ka * strideA + kb * strideB] = acc;
            }
        return;
    }
    std::vector<float> line(STRIP_COLS);
    for (const AxisStrip &st : t.strips) {
        for (int kb = 0; kb < t.nB; ++kb) {
            const AxisEntry &e = t.row[kb];
            for (int i = 0; i < STRIP_COLS; ++i) {
                const int col = st.x0 + i;
                float acc = 0.f;
                for (int y = e.s0; y <= e.s1; ++y) acc += row_w(e, y) * (col < g.W ? src[(int64_t)y * srcStride + col] : 0.f);
                line[i] = acc;
            }
            for (int k = st.k0; k < st.k1; ++k) {
                const AxisEntry &c = t.lane[k];
                const int off = c.s0 - st.x0, span = c.s1 - c.s0;
                float s = c.wFirst * line[off];
                if (span > 0) {
                    float mid = 0.f;
                    for (int i = 1; i < span; ++i) mid += line[off + i];
                    s += c.wMid * mid + c.wLast * line[off + span];
                }
                dst[base + k * strideA + kb * strideB] = s;
            }
        }
    }
}

// Interleaved channels through the axis-aligned path (mirrors enqueue() in csrc/aai_engine.cpp and the CH = true kernel):
// lane entries are (pixel, channel) pairs over the source row's elements, taps `C` elements apart.
static int emu_axis_channels(const Geometry &g, int mode, int C, const float *src, float *dst)
{
    AxisTables t;
    build_axis_tables(g, mode, t, C);
    if (t.channels != C || t.nA != (t.transposed ? g.dH : g.dW) * C) return 1;
    const int64_t srcStride = (int64_t)g.W * C, dstStride = (int64_t)g.dW * C;
    const int nApix = t.nA / C;
    const int64_t sa = t.transposed ? dstStride : C, sb = t.transposed ? C : dstStride;
    int64_t strideA = t.flipA ? -sa : sa;
    const int64_t strideB = t.flipB ? -sb : sb;
    const int64_t base = (t.flipA ? (int64_t)(nApix - 1) * sa : 0) + (t.flipB ? (int64_t)(t.nB - 1) * sb : 0);
    int outChan = C;
    if (C > 1 && !t.transposed && !t.flipA) { strideA = 1; outChan = 1; }
    auto out_off = [&](int ka) { return base + (int64_t)(ka / outChan) * strideA + ka % outChan; };
    // strips: in order, <= 256 entries, windows inside (unless the per-pixel fallback serves the request)
    int next = 0;
    for (const auto &s : t.strips) {
        if (s.k0 != next || s.k1 <= s.k0 || s.k1 - s.k0 > 256) return 2;
        for (int k = s.k0; k < s.k1; ++k) {
            if ((t.lane[k].s1 - t.lane[k].s0) % C != 0 || t.lane[k].s0 % C != k % C) return 3;
            if (!t.wide && (t.lane[k].s0 < s.x0 || t.lane[k].s1 >= s.x0 + STRIP_COLS)) return 4;
        }
        next = s.k1;
    }
    if (next != t.nA) return 5;
    std::vector<float> line(STRIP_COLS);
    for (const AxisStrip &st : t.strips)
        for (int kb = 0; kb < t.nB; ++kb) {
            const AxisEntry &e = t.row[kb];
            if (!t.wide)
                for (int i = 0; i < STRIP_COLS; ++i) {
                    const int col = st.x0 + i;
                    float acc = 0.f;
                    for (int y = e.s0; y <= e.s1; ++y) acc += row_w(e, y) * (col < g.W * C ? src[(int64_t)y * srcStride + col] : 0.f);
                    line[i] = acc;
                }
            for (int k = st.k0; k < st.k1; ++k) {
                const AxisEntry &c = t.lane[k];
                float s;
                if (t.wide) {
                    s = 0.f;
                    for (int y = e.s0; y <= e.s1; ++y) {
                        float h = 0.f;
                        for (int x = c.s0; x <= c.s1; x += C) h += row_w(c, x) * src[(int64_t)y * srcStride + x];
                        s += row_w(e, y) * h;
                    }
                } else {
                    const int off = c.s0 - st.x0, span = (c.s1 - c.s0) / C;
                    s = c.wFirst * line[off];
                    if (span > 0) {
                        float mid = 0.f;
                        for (int i = 1; i < span; ++i) mid += line[off + i * C];
                        s += c.wMid * mid + c.wLast * line[off + span * C];
                    }
                }
                dst[out_off(k) + (int64_t)kb * strideB] = s;
            }
        }
    return 0;
}

static int g_forceGeneral = 0;   // test hook: route every cut pair through pair_area (cross-checks the closed form)
static int g_strict = 1;         // test hook: 0 = production pass only, 1 = production + knife-edge fix-up pass
static long g_knifePairs = 0, g_knifePixels = 0, g_missedPairs = 0;
static int g_skipAxisFixup = 0;  // test hook: K1's separable weights alone, without the fix-up pass behind them
static int g_forceRotated = 0;   // test hook: axis-aligned requests take the per-pixel path (production pass + knife-edge fix-up) too
static int g_useQuad = 0;        // test hook: 1 = unflagged area-mode pixels take the fp32 quad formulation (aai_rot_quad.hpp) like the GPU does
static long g_quadPixels = 0, g_quadUncertain = 0;   // pixels answered by the quad path / left to the double-precision path by its scan
static int g_useCell = 0;        // test hook: 1 = unflagged area-mode pixels take the fp32 cell formulation (aai_rot_cell.hpp) like aai_cell_kernel does

// source access of the quad formulation: window slot -> virtual pixel -> image element (staged like the GPU does)
template <int WIN>
struct EmuQuadSrc {
    const RotLaunch *r; const float *img; int64_t stride;
    float v[WIN * WIN];
    void issue(int xg0, int yg0, unsigned long long valid, bool = false)
    {
        for (int j = 0; j < WIN; ++j)
            for (int i = 0; i < WIN; ++i)
                v[j * WIN + i] = ((valid >> (j * WIN + i)) & 1) ? img[virt_offset(*r, xg0 + i, yg0 + j, stride)] : -1e30f;   // poison: must never be read
    }
    void commit() {}
    void at(int slot, float (&vals)[1]) const { vals[0] = v[slot]; }
    float reg(int slot) const { return v[slot]; }
};

template <int WIN>
static bool emu_quad_pixel(const QuadConsts<float> &qc, const RotLaunch &r, const float *img, int64_t stride, double px, double py, float &value)
{
    const double cxr = std::floor(px + 0.5), cyr = std::floor(py + 0.5);
    if (!(std::fabs(cxr) < 1e9 && std::fabs(cyr) < 1e9)) return false;
    const int Xc = (int)cxr, Yc = (int)cyr;
    const double fpx = px - cxr, fpy = py - cyr;
    EmuQuadSrc<WIN> qs{&r, img, stride, {}};
    float sA, sVA[1];
    if (qc.hiPrec) {
        if (quad_pixel<float, WIN, true, true, 1>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA)) return false;      // the scan leaves it to double precision
        quad_pixel<float, WIN, false, true, 1>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA);
    } else {
        if (quad_pixel<float, WIN, true, false, 1>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA)) return false;
        quad_pixel<float, WIN, false, false, 1>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA);
    }
    value = sA > 0.f ? sVA[0] / sA : 0.f;
    return true;
}

// wide footprints (aai_wide_kernel): the window in PARTS x PARTS parts, each a quad_pixel of its own, summed in the lanes' order
template <int WIN>
static bool emu_wide_pixel(const QuadConsts<float> &qc, const RotLaunch &r, const float *img, int64_t stride, double px, double py, float &value)
{
    const double cxr = std::floor(px + 0.5), cyr = std::floor(py + 0.5);
    if (!(cxr > -40.0 && cxr < (double)r.mW + 40.0 && cyr > -40.0 && cyr < (double)r.mH + 40.0)) { value = 0.f; return true; }
    const int Xc = (int)cxr, Yc = (int)cyr;
    const double fpx = px - cxr, fpy = py - cyr;
    const int n = qc.parts * qc.parts;
    float a[16], va[16];
    bool uncertain = false;
    for (int pass = 0; pass < 2; ++pass)          // the plan's scan first, then the production arithmetic
        for (int part = 0; part < n; ++part) {
            EmuQuadSrc<WIN> qs{&r, img, stride, {}};
            float sA, sVA[1];
            const int pi = part % qc.parts, pj = part / qc.parts;
            if (pass == 0) {
                if (qc.hiPrec) uncertain |= quad_pixel<float, WIN, true, true, 1, EmuQuadSrc<WIN>, true>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA, pi, pj);
                else uncertain |= quad_pixel<float, WIN, true, false, 1, EmuQuadSrc<WIN>, true>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA, pi, pj);
            } else {
                if (qc.hiPrec) quad_pixel<float, WIN, false, true, 1, EmuQuadSrc<WIN>, true>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA, pi, pj);
                else quad_pixel<float, WIN, false, false, 1, EmuQuadSrc<WIN>, true>(qc, Xc, Yc, fpx, fpy, r.mW, r.mH, qs, sA, sVA, pi, pj);
            }
            a[part] = sA; va[part] = sVA[0];
            if (pass == 0 && part == n - 1) {
                float t[16];
                for (int k = 0; k < n; ++k) t[k] = a[k];
                const float A = quad_parts_sum(t, n);
                if (uncertain || (A > 0.f && A < qc.minArea)) return false;
            }
        }
    const float A = quad_parts_sum(a, n), VA = quad_parts_sum(va, n);
    value = A > 0.f ? VA / A : 0.f;
    return true;
}

// fast mode through the fp32 formulation (aai_quad_fast_kernel)
template <int WIN>
static bool emu_quad_fast_pixel(const QuadConsts<float> &qc, const RotLaunch &r, const float *img, int64_t stride, double px, double py, float &value)
{
    const double cxr = std::floor(px + 0.5), cyr = std::floor(py + 0.5);
    if (!(std::fabs(cxr) < 1e9 && std::fabs(cyr) < 1e9)) return false;
    EmuQuadSrc<WIN> qs{&r, img, stride, {}};
    float sum;
    int count;
    if (quad_fast_pixel<float, WIN, true>(qc, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, sum, count)) return false;
    quad_fast_pixel<float, WIN, false>(qc, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, sum, count);
    value = count > 0 ? sum / (float)count : 0.f;
    return true;
}

// fast mode over a wide footprint (aai_wide_fast_kernel): the window of centres in parts, sums and counts added in the lanes' order
template <int WIN>
static bool emu_wide_fast_pixel(const QuadConsts<float> &qc, const RotLaunch &r, const float *img, int64_t stride, double px, double py, float &value)
{
    const double cxr = std::floor(px + 0.5), cyr = std::floor(py + 0.5);
    if (!(cxr > -40.0 && cxr < (double)r.mW + 40.0 && cyr > -40.0 && cyr < (double)r.mH + 40.0)) { value = 0.f; return true; }
    const int n = qc.partsFast * qc.partsFast;
    float sums[16];
    int total = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int part = 0; part < n; ++part) {
            EmuQuadSrc<WIN> qs{&r, img, stride, {}};
            float sum;
            int count;
            const int pi = part % qc.partsFast, pj = part / qc.partsFast;
            if (pass == 0) {
                if (quad_fast_pixel<float, WIN, true>(qc, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, sum, count, pi, pj)) return false;
            } else {
                quad_fast_pixel<float, WIN, false>(qc, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, sum, count, pi, pj);
                sums[part] = sum; total += count;
            }
        }
    const float S = quad_parts_sum(sums, n);
    value = total > 0 ? S / (float)total : 0.f;
    return true;
}

// The cell formulation (aai_rot_cell.hpp) over a whole image, as aai_cell_kernel runs it: every cell (x, y), x in [0, dW],
// y in [0, dH], evaluated once (scan first: a cell with a decision too close to its threshold marks itself), its four parts
// kept; emu_rotated combines them per dst pixel in the kernel's order.
struct EmuCells {
    int W1 = 0, H1 = 0;
    std::vector<float> a[4], va[4];
    std::vector<unsigned char> unc;
    bool ok = false;
};
template <int WIN>
static void emu_cells_win(const RotLaunch &r, const QuadConsts<float> &qc, const CellConsts<float> &zc, const float *img, int64_t stride, EmuCells &out)
{
    for (int y = 0; y <= r.dH; ++y)
        for (int x = 0; x <= r.dW; ++x) {
            const size_t at = (size_t)y * out.W1 + x;
            int Zx, Zy; double dfx, dfy;
            if (!cell_anchor(r, cell_column(r, zc, x), y, Zx, Zy, dfx, dfy)) continue;
            EmuQuadSrc<WIN> qs{&r, img, stride, {}};
            float sA[4], sVA[4];
            bool u;
            if (qc.hiPrec) {
                u = cell_eval<float, WIN, true, true>(qc, zc, Zx, Zy, dfx, dfy, r.mW, r.mH, qs, sA, sVA);
                cell_eval<float, WIN, false, true>(qc, zc, Zx, Zy, dfx, dfy, r.mW, r.mH, qs, sA, sVA);
            } else {
                u = cell_eval<float, WIN, true, false>(qc, zc, Zx, Zy, dfx, dfy, r.mW, r.mH, qs, sA, sVA);
                cell_eval<float, WIN, false, false>(qc, zc, Zx, Zy, dfx, dfy, r.mW, r.mH, qs, sA, sVA);
            }
            out.unc[at] = u ? 1 : 0;
            for (int t = 0; t < 4; ++t) { out.a[t][at] = sA[t]; out.va[t][at] = sVA[t]; }
        }
}
static void emu_cells(const RotLaunch &r, const float *img, int64_t stride, EmuCells &out)
{
    const QuadConsts<float> qc = make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy);
    const CellConsts<float> zc = make_cell_consts<float>(r.side, r.c, r.s);
    out.W1 = r.dW + 1; out.H1 = r.dH + 1;
    const size_t n = (size_t)out.W1 * out.H1;
    for (int t = 0; t < 4; ++t) { out.a[t].assign(n, 0.f); out.va[t].assign(n, 0.f); }
    out.unc.assign(n, 0);
    out.ok = true;
    switch (zc.win) {
    case 1: emu_cells_win<1>(r, qc, zc, img, stride, out); break;
    case 2: emu_cells_win<2>(r, qc, zc, img, stride, out); break;
    case 3: emu_cells_win<3>(r, qc, zc, img, stride, out); break;
    case 4: emu_cells_win<4>(r, qc, zc, img, stride, out); break;
    case 5: emu_cells_win<5>(r, qc, zc, img, stride, out); break;
    case 6: emu_cells_win<6>(r, qc, zc, img, stride, out); break;
    case 7: emu_cells_win<7>(r, qc, zc, img, stride, out); break;
    case 8: emu_cells_win<8>(r, qc, zc, img, stride, out); break;
    default: out.ok = false; break;
    }
}
// dst pixel (x, y) from its four cells; false: one of them (or the pixel's total area) leaves it to double precision
static bool emu_cell_pixel(const EmuCells &c, const QuadConsts<float> &qc, int x, int y, float &value)
{
    const size_t o = (size_t)y * c.W1 + x, w = o + 1, n = o + c.W1, nw = n + 1;
    if (c.unc[o] || c.unc[w] || c.unc[n] || c.unc[nw]) return false;
    float A, VA;
    cell_combine(c.a[CELL_O][o], c.a[CELL_W][w], c.a[CELL_N][n], c.a[CELL_NW][nw], A);
    cell_combine(c.va[CELL_O][o], c.va[CELL_W][w], c.va[CELL_N][n], c.va[CELL_NW][nw], VA);
    if (A > 0.f && A < qc.minArea) return false;
    value = A > 0.f ? VA / A : 0.f;
    return true;
}

static long g_axisFixups = 0;      // dst pixels of the last axis-aligned request recomputed by the fix-up pass
// onlyAxisDiffering: the fix-up pass behind K1 -- only the dst pixels where the separable model departs from the
// reference's classifier (aai_axis_verify.hpp) are computed, the others keep what K1 wrote
static void emu_rotated(const Geometry &g, const aai_request &rq, const float *img, int64_t srcStride, float *dst, int64_t dstStride,
                        bool onlyAxisDiffering = false)
{
    const RotLaunch r = make_rot_launch(g, rq.mode, rq.policy);
    g_knifePairs = g_knifePixels = g_missedPairs = 0;
    g_quadPixels = g_quadUncertain = 0;
    const bool quad = g_useQuad && (r.quad || r.wide);
    const bool fastQuad = rq.mode == AAI_MODE_FAST;
    const QuadConsts<float> qc = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    EmuCells cells;
    if (g_useCell && r.quad && rq.mode == AAI_MODE_AREA && cell_supported(r.side, r.c, r.s) && !onlyAxisDiffering) emu_cells(r, img, srcStride, cells);
    for (int dy = 0; dy < r.dH; ++dy)
        for (int dx = 0; dx < r.dW; ++dx) {
            double px, py;
            pixel_centre(r, dx, dy, px, py);
            double qx, qy;                       // the centre as the fp32 window kernels compute it (aai_rotated_quad.hip)
            quad_centre(r, dx, dy, qx, qy);
            const double hb = r.h * (r.c + r.s);
            const int x0 = std::max(0, (int)std::floor(px - hb + 0.5 - AAI_KNIFE_GUARD)), x1 = std::min(r.mW - 1, (int)std::ceil(px + hb - 0.5 + AAI_KNIFE_GUARD));
            const int y0 = std::max(0, (int)std::floor(py - hb + 0.5 - AAI_KNIFE_GUARD)), y1 = std::min(r.mH - 1, (int)std::ceil(py + hb - 0.5 + AAI_KNIFE_GUARD));
            if (onlyAxisDiffering) {
                if (!(rq.mode == AAI_MODE_FAST ? axis_pixel_differs_fast(r, dx, dy) : axis_pixel_differs(r, dx, dy))) continue;
                ++g_axisFixups;
            }
            float *out = dst + (int64_t)dy * dstStride + dx;
            SVec sv4[4];
            bool haveVertices = false;
            long knifeHere = 0;
            // the production pass flags whole dst pixels; only flagged ones reach the strict replay
            const bool flagged = pixel_on_knife_edge(r, px, py, rq.mode != AAI_MODE_FAST);
            if (cells.ok && !flagged) {
                float value = 0.f;
                if (emu_cell_pixel(cells, qc, dx, dy, value)) { *out = value; ++g_quadPixels; continue; }
                ++g_quadUncertain;
            } else if (quad && !flagged) {
                // the GPU's production pass for generic pixels: fp32, relative to the nearest virtual pixel
                float value = 0.f;
                bool done;
                if (fastQuad && r.wide) {
                    switch (qc.winFast) {
                    case 5: done = emu_wide_fast_pixel<5>(qc, r, img, srcStride, px, py, value); break;
                    case 6: done = emu_wide_fast_pixel<6>(qc, r, img, srcStride, px, py, value); break;
                    case 7: done = emu_wide_fast_pixel<7>(qc, r, img, srcStride, px, py, value); break;
                    default: done = emu_wide_fast_pixel<8>(qc, r, img, srcStride, px, py, value); break;
                    }
                } else if (fastQuad) {
                    switch (qc.winFast) {
                    case 2: done = emu_quad_fast_pixel<2>(qc, r, img, srcStride, qx, qy, value); break;
                    case 3: done = emu_quad_fast_pixel<3>(qc, r, img, srcStride, qx, qy, value); break;
                    case 4: done = emu_quad_fast_pixel<4>(qc, r, img, srcStride, qx, qy, value); break;
                    case 5: done = emu_quad_fast_pixel<5>(qc, r, img, srcStride, qx, qy, value); break;
                    case 6: done = emu_quad_fast_pixel<6>(qc, r, img, srcStride, qx, qy, value); break;
                    case 7: done = emu_quad_fast_pixel<7>(qc, r, img, srcStride, qx, qy, value); break;
                    default: done = emu_quad_fast_pixel<8>(qc, r, img, srcStride, qx, qy, value); break;
                    }
                } else if (r.wide) {
                    switch (qc.win) {
                    case 5: done = emu_wide_pixel<5>(qc, r, img, srcStride, px, py, value); break;
                    case 6: done = emu_wide_pixel<6>(qc, r, img, srcStride, px, py, value); break;
                    case 7: done = emu_wide_pixel<7>(qc, r, img, srcStride, px, py, value); break;
                    default: done = emu_wide_pixel<8>(qc, r, img, srcStride, px, py, value); break;
                    }
                } else {
                    switch (qc.win) {
                    case 3: done = emu_quad_pixel<3>(qc, r, img, srcStride, qx, qy, value); break;
                    case 4: done = emu_quad_pixel<4>(qc, r, img, srcStride, qx, qy, value); break;
                    case 5: done = emu_quad_pixel<5>(qc, r, img, srcStride, qx, qy, value); break;
                    case 6: done = emu_quad_pixel<6>(qc, r, img, srcStride, qx, qy, value); break;
                    case 7: done = emu_quad_pixel<7>(qc, r, img, srcStride, qx, qy, value); break;
                    default: done = emu_quad_pixel<8>(qc, r, img, srcStride, qx, qy, value); break;
                    }
                }
                if (done) { *out = value; ++g_quadPixels; continue; }
                ++g_quadUncertain;
            }
            if (rq.mode == AAI_MODE_FAST && !flagged) {
                // production pass: one interval of centres per line (see aai_rotated_kernel): lines are source rows when
                // there is no replication (virtual rows in quadrants 0/2, virtual columns in 1/3), else virtual rows
                int count = 0; double acc = 0;
                const bool lines = r.scale == 1 && std::min(r.mW, r.mH) >= 4;
                const bool cols = lines && virt_lines_are_columns(r);
                const int u0 = cols ? x0 : y0, u1 = cols ? x1 : y1;
                const int nIn = cols ? r.mH : r.mW;
                for (int u = u0; u <= u1; ++u) {
                    double lo, hi;
                    centre_interval(r, cols, u - (cols ? px : py), lo, hi);
                    const double pIn = cols ? py : px;
                    const double da = std::fmax(std::ceil(pIn + lo), 0.0), db = std::fmin(std::floor(pIn + hi), (double)(nIn - 1));
                    if (!(da <= db)) continue;
                    const int wa = (int)da, wb = (int)db;
                    bool rev = false;
                    const int64_t line = lines ? virt_line(r, u, srcStride, rev) : 0;
                    for (int w = wa; w <= wb; ++w) {
                        const int X = cols ? u : w, Y = cols ? w : u;
                        const int64_t off = virt_offset(r, X, Y, srcStride);
                        if (lines && off != line + (rev ? nIn - 1 - w : w)) ++g_missedPairs;       // virt_line must agree with virt_offset
                        acc += (double)img[off];
                    }
                    count += wb - wa + 1;
                }
                *out = count > 0 ? (float)(acc / count) : 0.f;
            } else if (rq.mode == AAI_MODE_FAST) {
                const double lim = r.h + DBL_EPSILON * r.side;
                int count = 0; double acc = 0;
                for (int Y = y0; Y <= y1; ++Y)
                    for (int X = x0; X <= x1; ++X) {
                        const double ex = X - px, ey = Y - py;
                        const double a = std::fabs(ex * r.c - ey * r.s), b = std::fabs(ex * r.s + ey * r.c);
                        bool in = a <= lim && b <= lim;
                        const bool edgy = (std::fabs(a - r.h) < AAI_KNIFE_GUARD && b <= r.h + AAI_KNIFE_GUARD) ||
                                          (std::fabs(b - r.h) < AAI_KNIFE_GUARD && a <= r.h + AAI_KNIFE_GUARD);
                        if (edgy && !flagged) {
                            SVec tv[4];
                            strict_vertices(r, dx, dy, tv);
                            SVec pc; pc.x = X; pc.y = Y;
                            if (strict_centre_inside(pc, tv) != in) ++g_missedPairs;
                        }
                        if (edgy && flagged) {
                            ++knifeHere;
                            if (g_strict) {
                                if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                                SVec pc; pc.x = X; pc.y = Y;
                                in = strict_centre_inside(pc, sv4);
                            }
                        }
                        if (in) { ++count; acc += (double)img[virt_offset(r, X, Y, srcStride)]; }
                    }
                *out = count > 0 ? (float)(acc / count) : 0.f;
            } else {
                double sumA = 0, sumVA = 0;
                // one (dst, src) pair: class, area, knife accounting -- shared by the two loop structures below
                auto pair = [&](int X, int Y) {
                    const double ex = X - px, ey = Y - py;
                    const double a = ex * r.c - ey * r.s, b = ex * r.s + ey * r.c;
                    double d = 0;
                    bool edgy = false, edgy2 = false;
                    const int cls = classify_pair<true>(r, a, b, d, edgy);
                    if (cls == PAIR_OUTSIDE) return;
                    double area;
                    if (cls == PAIR_INSIDE) area = 1.0;
                    else if (g_forceGeneral) area = pair_area<true>(r, px - (X - 0.5), py - (Y - 0.5), r.policy, edgy2);   // four-edge form
                    else if (cls == PAIR_GENERAL) area = wedge_pair_area<true>(r, px - (X - 0.5), py - (Y - 0.5), a < 0.0, b < 0.0, r.policy, edgy2);
                    else area = single_cut_area<true>(r, d, cls == PAIR_CUT_LR, r.policy, edgy2);
                    // A pair-level knife flag in a pixel the per-pixel test let through is harmless as long as
                    // the strict replay would not have changed the area (e.g. a pixel corner on the EXTENSION
                    // of an edge line beyond the vertex); anything else is a gap in the per-pixel test.
                    if ((edgy || edgy2) && !flagged) {
                        SVec tv[4];
                        strict_vertices(r, dx, dy, tv);
                        if (std::fabs(strict_pair_area(tv, X, Y, r.policy) - area) > 1e-9) ++g_missedPairs;
                    }
                    if ((edgy || edgy2) && flagged) {
                        ++knifeHere;
                        if (g_strict) {
                            if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                            area = strict_pair_area(sv4, X, Y, r.policy);
                        }
                    }
                    if (area != 0.0) { sumA += area; sumVA += area * (double)img[virt_offset(r, X, Y, srcStride)]; }
                };
                if (r.runs && !(flagged && g_strict)) {
                    // production pass for large footprints (aai_rotated_runs_kernel): boundary | interior | boundary
                    const bool cols = virt_lines_are_columns(r);
                    const int u0 = cols ? x0 : y0, u1 = cols ? x1 : y1, w0 = cols ? y0 : x0, w1 = cols ? y1 : x1;
                    const int nIn = cols ? r.mH : r.mW;
                    for (int u = u0; u <= u1; ++u) {
                        int t0, t1, i0, i1;
                        line_runs(r, cols, cols ? py : px, u - (cols ? px : py), w0, w1, t0, t1, i0, i1);
                        if (t0 > t1) continue;
                        bool rev;
                        const int64_t line = virt_line(r, u, srcStride, rev);
                        double sum = 0;
                        for (int w = t0; w <= t1; ++w) {
                            const int X = cols ? u : w, Y = cols ? w : u;
                            const int64_t off = line + (rev ? nIn - 1 - w : w);
                            if (off != virt_offset(r, X, Y, srcStride)) ++g_missedPairs;   // virt_line must agree with virt_offset
                            if (w >= i0 && w <= i1) sum += (double)img[off];
                            else pair(X, Y);
                        }
                        if (i0 <= i1) { sumVA += sum; sumA += (double)(i1 - i0 + 1); }
                    }
                } else {
                    for (int Y = y0; Y <= y1; ++Y)
                        for (int X = x0; X <= x1; ++X) pair(X, Y);
                }
                *out = DBL_EPSILON < std::fabs(sumA) ? (float)(sumVA / sumA) : 0.f;
            }
            g_knifePairs += knifeHere;
            if (knifeHere) ++g_knifePixels;
        }
}

// debugging aid: the quad sums of one dst pixel in fp32 (as the GPU computes them) and in double precision
template <typename F, int WIN>
static void emu_quad_sums(const RotLaunch &r, const float *img, double px, double py, double *sumA, double *sumVA)
{
    struct Src {
        const RotLaunch *r; const float *img; int64_t stride; float v[WIN * WIN];
        void issue(int xg0, int yg0, unsigned long long valid) { for (int j = 0; j < WIN; ++j) for (int i = 0; i < WIN; ++i) v[j * WIN + i] = ((valid >> (j * WIN + i)) & 1) ? img[virt_offset(*r, xg0 + i, yg0 + j, stride)] : 0.f; }
        void commit() {}
        void at(int slot, F (&vals)[1]) const { vals[0] = (F)v[slot]; }
    } qs{&r, img, r.W, {}};
    const QuadConsts<F> q = make_quad_consts<F>(r.side, r.c, r.s, r.policy, r.scale);
    const double cxr = std::floor(px + 0.5), cyr = std::floor(py + 0.5);
    F a, va[1];
    if (q.hiPrec) quad_pixel<F, WIN, false, true, 1>(q, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, a, va);
    else quad_pixel<F, WIN, false, false, 1>(q, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, a, va);
    *sumA = a; *sumVA = va[0];
}
// per-slot areas of one dst pixel: the value source is 1 at one slot and 0 elsewhere
template <typename F, int WIN>
static void emu_quad_slot_areas(const RotLaunch &r, double px, double py, double *areas)
{
    struct Src { int hot; void issue(int, int, unsigned long long) {} void commit() {} void at(int slot, F (&vals)[1]) const { vals[0] = slot == hot ? F(1) : F(0); } };
    const QuadConsts<F> q = make_quad_consts<F>(r.side, r.c, r.s, r.policy, r.scale);
    const double cxr = std::floor(px + 0.5), cyr = std::floor(py + 0.5);
    for (int k = 0; k < WIN * WIN; ++k) {
        Src qs{k};
        F a, va[1];
        if (q.hiPrec) quad_pixel<F, WIN, false, true, 1>(q, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, a, va);
        else quad_pixel<F, WIN, false, false, 1>(q, (int)cxr, (int)cyr, px - cxr, py - cyr, r.mW, r.mH, qs, a, va);
        areas[k] = va[0];
    }
}
// debugging aid / precision check: the four parts (own, W, N, NW: area sums, then area x value sums) of dst pixel (dx, dy)
// in the cell formulation, evaluated in fp32 as the kernel does (out32[8]) and with the same code in double precision
// (out64[8]); returns the cell window size, or < 0
template <typename F, int WIN>
static void emu_cell_parts(const RotLaunch &r, const float *img, int dx, int dy, double *out)
{
    struct Src {
        const RotLaunch *r; const float *img; int64_t stride; float v[WIN * WIN];
        void issue(int xg0, int yg0, unsigned long long valid, bool = false) { for (int j = 0; j < WIN; ++j) for (int i = 0; i < WIN; ++i) v[j * WIN + i] = ((valid >> (j * WIN + i)) & 1) ? img[virt_offset(*r, xg0 + i, yg0 + j, stride)] : 0.f; }
        void commit() {}
        void at(int slot, F (&vals)[1]) const { vals[0] = (F)v[slot]; }
    };
    const QuadConsts<F> q = make_cell_quad_consts<F>(r.side, r.c, r.s, r.policy);
    const CellConsts<F> z = make_cell_consts<F>(r.side, r.c, r.s);
    const int cxs[4] = {dx, dx + 1, dx, dx + 1}, cys[4] = {dy, dy, dy + 1, dy + 1};
    for (int t = 0; t < 4; ++t) {
        out[t] = out[4 + t] = 0;
        int Zx, Zy; double dfx, dfy;
        if (!cell_anchor(r, cell_column(r, z, cxs[t]), cys[t], Zx, Zy, dfx, dfy)) continue;
        Src qs{&r, img, r.W, {}};
        F sA[4], sVA[4];
        if (q.hiPrec) cell_eval<F, WIN, false, true>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, qs, sA, sVA);
        else cell_eval<F, WIN, false, false>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, qs, sA, sVA);
        out[t] = sA[t]; out[4 + t] = sVA[t];
    }
}
extern "C" {
int aai_emu_quad_slot_debug(const aai_request *rq, int dx, int dy, double *f32areas, double *f64areas)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -1;
    const RotLaunch r = make_rot_launch(g, AAI_MODE_AREA, rq->policy);
    double px, py; quad_centre(r, dx, dy, px, py);
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    switch (q.win) {
    case 3: emu_quad_slot_areas<float, 3>(r, px, py, f32areas); emu_quad_slot_areas<double, 3>(r, px, py, f64areas); break;
    case 4: emu_quad_slot_areas<float, 4>(r, px, py, f32areas); emu_quad_slot_areas<double, 4>(r, px, py, f64areas); break;
    case 5: emu_quad_slot_areas<float, 5>(r, px, py, f32areas); emu_quad_slot_areas<double, 5>(r, px, py, f64areas); break;
    default: return -2;
    }
    return q.win;
}

// Returns an AAI_* status; on success fills dW/dH and writes dW*dH floats to dst (if dst != NULL).
int aai_emu_resample(const aai_request *rq, const float *src, float *dst, int *dW, int *dH, int *usedAxisPath)
{
    Geometry g;
    std::string msg;
    int rc = make_geometry(*rq, g, msg);
    if (rc != AAI_OK) return rc;
    *dW = g.dW; *dH = g.dH;
    const bool axis = g.axisAligned && !g_forceRotated && (rq->mode == AAI_MODE_AREA || rq->mode == AAI_MODE_FAST);
    if (usedAxisPath) *usedAxisPath = axis ? 1 : 0;
    if (!dst || !g.dW || !g.dH) return AAI_OK;
    g_axisFixups = 0;
    if (axis) {
        emu_axis(g, rq->mode, src, g.W, dst, g.dW);
        // (mirrors get_plan / enqueue in aai_engine.cpp)
        if (!g_skipAxisFixup)
            emu_rotated(g, *rq, src, g.W, dst, g.dW, true);
    } else emu_rotated(g, *rq, src, g.W, dst, g.dW);
    return AAI_OK;
}

// 1 when the area mode of this request takes the rows-as-runs production kernel (RotLaunch::runs)
int aai_emu_uses_runs(const aai_request *rq)
{
    Geometry g;
    std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK || g.axisAligned) return 0;
    return make_rot_launch(g, rq->mode, rq->policy).runs;
}

// the live tile span table the plan uploads for a rotated canvas (rotated_live_spans): 2 ints per 16-row tile row; returns the
// number of ints (0: no table), -1 on a bad request / too small a buffer
int aai_emu_live_spans(const aai_request *rq, int *out, int capacity)
{
    Geometry g;
    std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -1;
    std::vector<int> spans;
    rotated_live_spans(make_rot_launch(g, rq->mode, rq->policy), rq->mode == AAI_MODE_BILINEAR || rq->mode == AAI_MODE_BICUBIC, spans);
    if ((int)spans.size() > capacity) return -1;
    for (size_t i = 0; i < spans.size(); ++i) out[i] = spans[i];
    return (int)spans.size();
}

// parts per axis of the fp32 window of a wide footprint (RotLaunch::wide; 0: not one)
int aai_emu_wide_parts(const aai_request *rq)
{
    Geometry g;
    std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK || g.axisAligned) return 0;
    return make_rot_launch(g, rq->mode, rq->policy).wide;
}

// Property behind the rows-as-runs kernel: for every dst pixel and every line of its window, line_runs' interior
// pixels are PAIR_INSIDE for classify_pair and the pixels outside its touched interval are PAIR_OUTSIDE -- along rows
// and along columns.  Returns the number of violations (0 expected), -1 on a bad request.
long aai_emu_check_line_runs(const aai_request *rq)
{
    Geometry g;
    std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK || g.axisAligned) return -1;
    const RotLaunch r = make_rot_launch(g, AAI_MODE_AREA, rq->policy);
    long bad = 0;
    for (int dy = 0; dy < r.dH; ++dy)
        for (int dx = 0; dx < r.dW; ++dx) {
            double px, py;
            pixel_centre(r, dx, dy, px, py);
            const double hb = r.h * (r.c + r.s);
            const int x0 = std::max(0, (int)std::floor(px - hb + 0.5 - AAI_KNIFE_GUARD)), x1 = std::min(r.mW - 1, (int)std::ceil(px + hb - 0.5 + AAI_KNIFE_GUARD));
            const int y0 = std::max(0, (int)std::floor(py - hb + 0.5 - AAI_KNIFE_GUARD)), y1 = std::min(r.mH - 1, (int)std::ceil(py + hb - 0.5 + AAI_KNIFE_GUARD));
            for (int cols = 0; cols < 2; ++cols) {
                const int u0 = cols ? x0 : y0, u1 = cols ? x1 : y1, w0 = cols ? y0 : x0, w1 = cols ? y1 : x1;
                for (int u = u0; u <= u1; ++u) {
                    int t0, t1, i0, i1;
                    line_runs(r, cols != 0, cols ? py : px, u - (cols ? px : py), w0, w1, t0, t1, i0, i1);
                    for (int w = w0; w <= w1; ++w) {
                        const int X = cols ? u : w, Y = cols ? w : u;
                        const double ex = X - px, ey = Y - py;
                        double d = 0;
                        bool edgy = false;
                        const int cls = classify_pair<false>(r, ex * r.c - ey * r.s, ex * r.s + ey * r.c, d, edgy);
                        const bool touched = t0 <= t1 && w >= t0 && w <= t1, interior = touched && i0 <= i1 && w >= i0 && w <= i1;
                        if (interior && cls != PAIR_INSIDE) ++bad;
                        if (!touched && cls != PAIR_OUTSIDE) ++bad;
                    }
                }
            }
        }
    return bad;
}

// Axis-aligned requests with C interleaved channels: src is [H][W][C], dst [dH][dW][C].  0 on success, > 0 = an
// invariant of the channel tables failed, < 0 = bad request / not axis-aligned.
int aai_emu_resample_channels(const aai_request *rq, int C, const float *src, float *dst)
{
    Geometry g;
    std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -1;
    if (!g.axisAligned || (rq->mode != AAI_MODE_AREA && rq->mode != AAI_MODE_FAST)) return -2;
    if (!g.dW || !g.dH) return 0;
    return emu_axis_channels(g, rq->mode, C, src, dst);
}

void aai_emu_force_general(int on) { g_forceGeneral = on; }
void aai_emu_use_quad(int on) { g_useQuad = on; }
void aai_emu_use_cell(int on) { g_useCell = on; }

// Do the source rows aai_band_source_rows reports for dst rows [r0, r1) hold every pixel the cell kernel FETCHES for that band
// (the staged windows of cells [0, dW] x [r0, r1], clamped to the lattice like QuadSrc::issue clamps them)?  Returns the number
// of window pixels outside [srcRow0, srcRow1), or -1 when the cell formulation does not serve the request.
long aai_emu_cell_band_cover(const aai_request *rq, int r0, int r1)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -2;
    const RotLaunch r = make_rot_launch(g, rq->mode, rq->policy);
    if (!r.cell) return -1;
    int a = 0, b = 0;
    rotated_band_source_rows(g, r0, r1, false, a, b);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    long outside = 0;
    for (int cy = r0; cy <= r1; ++cy)
        for (int cx = 0; cx <= r.dW; ++cx) {
            int Zx, Zy; double dfx, dfy;
            if (!cell_anchor(r, cell_column(r, z, cx), cy, Zx, Zy, dfx, dfy)) continue;
            const int xg0 = Zx + (int)std::ceil((float)dfx - z.hbz), yg0 = Zy + (int)std::ceil((float)dfy - z.hbz);
            if (xg0 > r.mW - 1 || xg0 + z.win - 1 < 0 || yg0 > r.mH - 1 || yg0 + z.win - 1 < 0) continue;      // whole window off the lattice: nothing is fetched
            for (int j = 0; j < z.win; ++j)
                for (int i = 0; i < z.win; ++i) {
                    const int X = std::min(std::max(xg0 + i, 0), r.mW - 1), Y = std::min(std::max(yg0 + j, 0), r.mH - 1);
                    const int64_t off = virt_offset(r, X, Y, /*rowStride*/ 1 << 20);
                    const int row = (int)(off >> 20);
                    if (row < a || row >= b) ++outside;
                }
        }
    return outside;
}

// The same for the wide-footprint kernel (aai_wide_kernel): every part's window of every dst pixel of rows [r0, r1), clamped like
// QuadSrc::issue clamps it.  -1: not a wide footprint.
long aai_emu_wide_band_cover(const aai_request *rq, int r0, int r1)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -2;
    const RotLaunch r = make_rot_launch(g, rq->mode, rq->policy);
    if (!r.wide) return -1;
    int a = 0, b = 0;
    rotated_band_source_rows(g, r0, r1, false, a, b);
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    long outside = 0;
    for (int dy = r0; dy < r1; ++dy)
        for (int dx = 0; dx < r.dW; ++dx) {
            double px, py;
            pixel_centre(r, dx, dy, px, py);
            const double cx = std::floor(px + 0.5), cy = std::floor(py + 0.5);
            if (!(cx > -40.0 && cx < (double)r.mW + 40.0 && cy > -40.0 && cy < (double)r.mH + 40.0)) continue;
            const float fpx = (float)(px - cx), fpy = (float)(py - cy);
            for (int part = 0; part < q.parts * q.parts; ++part) {
                const int xg0 = (int)cx + (int)(std::floor(fpx - q.hbm) + (float)((part % q.parts) * q.win));
                const int yg0 = (int)cy + (int)(std::floor(fpy - q.hbm) + (float)((part / q.parts) * q.win));
                if (xg0 > r.mW - 1 || xg0 + q.win - 1 < 0 || yg0 > r.mH - 1 || yg0 + q.win - 1 < 0) continue;      // the part misses the lattice: nothing is fetched
                for (int j = 0; j < q.win; ++j)
                    for (int i = 0; i < q.win; ++i) {
                        const int X = std::min(std::max(xg0 + i, 0), r.mW - 1), Y = std::min(std::max(yg0 + j, 0), r.mH - 1);
                        const int64_t off = virt_offset(r, X, Y, /*rowStride*/ 1 << 20);
                        const int row = (int)(off >> 20);
                        if (row < a || row >= b) ++outside;
                    }
            }
        }
    return outside;
}

// ---- one cover check for every kernel family that takes a band buffer --------------------------------------------------------------
// Replays, for dst rows [r0, r1) of a request, what a family's kernel FETCHES, and counts the fetched elements whose source row lies
// outside the rows aai_band_source_rows reports (rotated_band_source_rows).  The window kernels (QuadSrc: quad, fast -- 16 x 4 and
// row-shaped waves evaluate the same pixels --, wide, wide fast, cell) are replayed through the product's own pixel functions
// (quad_pixel / quad_fast_pixel / cell_eval of the math headers) with a source that records instead of loading: window origins, sizes,
// parts and skip conditions are the kernels' own code, not a restatement; QuadSrc::issue fetches the window's positions clamped to the
// lattice (its vector loads bring exactly one window line each, its 2 x 2 fetch a subset).  The double-precision kernels (production,
// rows-as-runs, the strict fix-up pass) visit rot_window(), their 16-byte segment loads stay inside a source row; the samplers fetch
// N tap rows from floor(sy) + FIRST, clamped to the image.
// family: 0 = double precision, 1 = quad / wide (area), 2 = quad / wide (fast), 3 = cell, 4 = bilinear, 5 = bicubic.
// Returns the number of elements outside, -1: the family does not serve the request, -2: bad request.
extern "C++" {
namespace {
struct CoverCount { const RotLaunch *r; int a, b; long outside, fetched; };
template <int WIN>
struct CoverSrc {
    CoverCount *c;
    void issue(int xg0, int yg0, unsigned long long, bool = false)
    {
        const RotLaunch &r = *c->r;
        for (int j = 0; j < WIN; ++j)
            for (int i = 0; i < WIN; ++i) {
                const int X = std::min(std::max(xg0 + i, 0), r.mW - 1), Y = std::min(std::max(yg0 + j, 0), r.mH - 1);
                const int row = (int)(virt_offset(r, X, Y, /*rowStride*/ 1 << 20) >> 20);
                ++c->fetched;
                if (row < c->a || row >= c->b) ++c->outside;
            }
    }
    void commit() {}
    void at(int, float (&vals)[1]) const { vals[0] = 0.f; }
    float reg(int) const { return 0.f; }
};

template <int WIN, bool HP>
void cover_quad_pixel(const RotLaunch &r, const QuadConsts<float> &q, int dx, int dy, CoverCount &cc)
{
    double px, py;
    if (q.parts > 1) pixel_centre(r, dx, dy, px, py); else quad_centre(r, dx, dy, px, py);      // (as aai_wide_kernel / aai_quad_kernel do)
    const double cx = std::floor(px + 0.5), cy = std::floor(py + 0.5);
    // (the kernels' reach tests: 16 for one window, 40 for a wide footprint's parts)
    const double far = q.parts > 1 ? 40.0 : 16.0;
    if (!(cx > -far && cx < (double)r.mW + far && cy > -far && cy < (double)r.mH + far)) return;
    CoverSrc<WIN> s{&cc};
    float sumA, sumVA[1];
    if (q.parts == 1) quad_pixel<float, WIN, false, HP, 1>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA);
    else
        for (int part = 0; part < q.parts * q.parts; ++part)
            quad_pixel<float, WIN, false, HP, 1, CoverSrc<WIN>, true>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA, part % q.parts, part / q.parts);
}
template <int WIN>
void cover_fast_pixel(const RotLaunch &r, const QuadConsts<float> &q, int dx, int dy, CoverCount &cc)
{
    double px, py;
    if (q.partsFast > 1) pixel_centre(r, dx, dy, px, py); else quad_centre(r, dx, dy, px, py);
    const double cx = std::floor(px + 0.5), cy = std::floor(py + 0.5);
    const double far = q.partsFast > 1 ? 40.0 : 16.0;
    if (!(cx > -far && cx < (double)r.mW + far && cy > -far && cy < (double)r.mH + far)) return;
    CoverSrc<WIN> s{&cc};
    float sum; int count;
    for (int part = 0; part < q.partsFast * q.partsFast; ++part)
        quad_fast_pixel<float, WIN, false>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sum, count, part % q.partsFast, part / q.partsFast);
}
template <int WIN, bool HP>
void cover_cell(const RotLaunch &r, const QuadConsts<float> &q, const CellConsts<float> &z, int cx, int cy, bool upOnly, CoverCount &cc)
{
    // as aai_cell_kernel does: every cell of a strip's live rows is evaluated, without a reach test, whatever its window meets
    int Zx, Zy; double dfx, dfy;
    cell_anchor<true>(r, cell_column(r, z, cx), cy, Zx, Zy, dfx, dfy);
    CoverSrc<WIN> s{&cc};
    float sA[4], sVA[4];
    cell_eval<float, WIN, false, HP>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA, upOnly);
}
}  // namespace
}  // extern "C++"

// replicated_indices against integer division, for every window origin from `lo` to `hi` on a lattice of n * scale virtual pixels:
// the number of window positions whose source index differs from clamp(g, 0, n scale - 1) / scale
long aai_emu_replicated_indices(int n, int scale, int lo, int hi)
{
    long bad = 0;
    const int mN = n * scale;
    for (int g0 = lo; g0 <= hi; ++g0) {
        int q[8];
        replicated_indices<8>(g0, mN, scale, 1.0 / scale, (float)(1.0 / scale), q);
        for (int i = 0; i < 8; ++i) {
            const int g = g0 + i, c = g < 0 ? 0 : (g > mN - 1 ? mN - 1 : g);
            if (q[i] != c / scale) ++bad;
        }
    }
    return bad;
}

long aai_emu_band_cover(const aai_request *rq, int r0, int r1, int family, long *fetched)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -2;
    const bool sampler = family >= 4;
    const int mode = family == 2 ? AAI_MODE_FAST : (family == 4 ? AAI_MODE_BILINEAR : (family == 5 ? AAI_MODE_BICUBIC : AAI_MODE_AREA));
    const RotLaunch r = make_rot_launch(g, mode, rq->policy);
    CoverCount cc{&r, 0, 0, 0, 0};
    rotated_band_source_rows(g, r0, r1, sampler, cc.a, cc.b);
    if (fetched) *fetched = 0;
    if (family == 0) {
        for (int dy = r0; dy < r1; ++dy)
            for (int dx = 0; dx < r.dW; ++dx) {
                double px, py;
                pixel_centre(r, dx, dy, px, py);
                int x0, x1, y0, y1;
                rot_window(r, px, py, x0, x1, y0, y1);
                for (int Y = y0; Y <= y1; ++Y)
                    for (int X = x0; X <= x1; ++X) {
                        const int row = (int)(virt_offset(r, X, Y, 1 << 20) >> 20);
                        ++cc.fetched;
                        if (row < cc.a || row >= cc.b) ++cc.outside;
                    }
            }
    } else if (family == 1 || family == 2) {
        if (!(r.c > 0.0 && r.s > 0.0)) return -1;
        const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
        const bool fast = family == 2;
        if (fast ? !(r.quad || r.wide) : !(r.quad || r.wide)) return -1;
        const int win = fast ? q.winFast : q.win;
        for (int dy = r0; dy < r1; ++dy)
            for (int dx = 0; dx < r.dW; ++dx) {
#define AAI_COVER_CASE(W)                                                                                                  \
    case W:                                                                                                                \
        if (fast) cover_fast_pixel<W>(r, q, dx, dy, cc);                                                                   \
        else if (q.hiPrec) cover_quad_pixel<W, true>(r, q, dx, dy, cc);                                                    \
        else cover_quad_pixel<W, false>(r, q, dx, dy, cc);                                                                 \
        break;
                switch (win) { AAI_COVER_CASE(2) AAI_COVER_CASE(3) AAI_COVER_CASE(4) AAI_COVER_CASE(5) AAI_COVER_CASE(6) AAI_COVER_CASE(7) AAI_COVER_CASE(8) default: return -1; }
#undef AAI_COVER_CASE
            }
    } else if (family == 3) {
        if (!r.cell) return -1;
        const QuadConsts<float> q = make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy);
        const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
        const CellLive live = make_cell_live(r, z);
        for (int cy = r0; cy <= r1; ++cy)
            for (int cx = 0; cx <= r.dW; ++cx) {
                // (the kernel's strips of 63 columns: rows outside a strip's live range are not evaluated at all)
                // (a strip holds the wave's cell columns: its dst columns and the first one of the next strip; with two cell rows per
                // step -- cell_wave_rows -- a row is evaluated when it or its partner of the step is live: one row of margin covers either
                // pairing)
                int liveLo, liveHi;
                const int waveRows = cell_wave_rows(r.side, r.scale, r.c, r.s), cols = cell_wave_cols(waveRows), pairing = waveRows - 1;
                const int x0 = std::min(cx, r.dW - 1) / cols * cols;
                cell_live_rows(live, x0, x0 + cols, liveLo, liveHi);
                bool rowLive = cy >= liveLo - pairing && cy <= liveHi + pairing;
                if (cx % cols == 0 && cx >= cols) {
                    cell_live_rows(live, cx - cols, cx, liveLo, liveHi);
                    rowLive = rowLive || (cy >= liveLo - pairing && cy <= liveHi + pairing);
                }
                if (!rowLive) continue;
#define AAI_COVER_CASE(W)                                                                                                  \
    case W:                                                                                                                \
        if (q.hiPrec) cover_cell<W, true>(r, q, z, cx, cy, cy == r1, cc);                                                  \
        else cover_cell<W, false>(r, q, z, cx, cy, cy == r1, cc);                                                          \
        break;
                switch (z.win) { AAI_COVER_CASE(2) AAI_COVER_CASE(3) AAI_COVER_CASE(4) AAI_COVER_CASE(5) AAI_COVER_CASE(6) AAI_COVER_CASE(7) AAI_COVER_CASE(8) default: return -1; }
#undef AAI_COVER_CASE
            }
    } else {
        // aai_sample_kernel: sample point from the host-composed coefficients, N tap rows from floor(sy) + FIRST clamped to the image
        const int N = family == 4 ? 2 : 4, FIRST = family == 4 ? 0 : -1;
        for (int dy = r0; dy < r1; ++dy)
            for (int dx = 0; dx < r.dW; ++dx) {
                const double sx = std::fma((double)dx, r.sAx, std::fma((double)dy, r.sBx, r.sCx)), sy = std::fma((double)dx, r.sAy, std::fma((double)dy, r.sBy, r.sCy));
                const double guard = 1e-9;
                if (sx < -0.5 - guard || sx > r.W - 0.5 + guard || sy < -0.5 - guard || sy > r.H - 0.5 + guard) continue;      // outside the extent: no taps
                const int iy = (int)std::floor(sy);
                for (int k = 0; k < N; ++k) {
                    const int row = std::min(std::max(iy + FIRST + k, 0), r.H - 1);
                    cc.fetched += N;
                    if (row < cc.a || row >= cc.b) cc.outside += N;
                }
            }
    }
    if (fetched) *fetched = cc.fetched;
    return cc.outside;
}

// aai_quad_fast_lds_kernel stages the box of every 16 x 16 dst tile (fast_tile_box) in LDS and its lanes take their windows from it:
// counts the lattice positions of windows (quad_fast_pixel's own, through a recording source) that lie OUTSIDE their tile's box.
// -1: the staged kernel does not serve the request (replication, a box beyond the LDS budget, a wide footprint).
extern "C++" {
namespace {
struct BoxCount { int X0, X1, Y0, Y1, mW, mH; long outside, seen; };
template <int WIN>
struct BoxSrc {
    BoxCount *c;
    void issue(int xg0, int yg0, unsigned long long, bool = false)
    {
        for (int j = 0; j < WIN; ++j)
            for (int i = 0; i < WIN; ++i) {
                const int X = xg0 + i, Y = yg0 + j;
                if (X < 0 || X >= c->mW || Y < 0 || Y >= c->mH) continue;      // off the lattice: never used
                ++c->seen;
                if (X < c->X0 || X > c->X1 || Y < c->Y0 || Y > c->Y1) ++c->outside;
            }
    }
    float reg(int) const { return 0.f; }
};
template <int WIN>
void box_tile(const RotLaunch &r, const QuadConsts<float> &q, const FastTile &ft, int tx, int ty, BoxCount &bc)
{
    const bool any = fast_tile_box(r, ft, tx * 16, ty * 16, bc.X0, bc.X1, bc.Y0, bc.Y1);
    if (!any) { bc.X0 = bc.Y0 = 1; bc.X1 = bc.Y1 = 0; }                        // an empty box: every lattice position of a window counts
    for (int dy = ty * 16; dy < std::min(ty * 16 + 16, r.dH); ++dy)
        for (int dx = tx * 16; dx < std::min(tx * 16 + 16, r.dW); ++dx) {
            double px, py;
            quad_centre(r, dx, dy, px, py);
            const double cx = std::floor(px + 0.5), cy = std::floor(py + 0.5);
            if (!(cx > -16.0 && cx < (double)r.mW + 16.0 && cy > -16.0 && cy < (double)r.mH + 16.0)) continue;
            BoxSrc<WIN> s{&bc};
            float sum; int count;
            quad_fast_pixel<float, WIN, false>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sum, count);
        }
}
}  // namespace
}  // extern "C++"

long aai_emu_fast_tile_cover(const aai_request *rq, long *seen)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -2;
    const RotLaunch r = make_rot_launch(g, AAI_MODE_FAST, rq->policy);
    if (!r.quad || r.scale != 1 || !(r.c > 0.0 && r.s > 0.0)) return -1;
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    if (q.partsFast != 1) return -1;
    FastTile ft;
    if (!make_fast_tile(r, q, ft)) return -1;
    BoxCount bc{0, 0, 0, 0, r.mW, r.mH, 0, 0};
    for (int ty = 0; ty < (r.dH + 15) / 16; ++ty)
        for (int tx = 0; tx < (r.dW + 15) / 16; ++tx) {
            switch (q.winFast) {
            case 2: box_tile<2>(r, q, ft, tx, ty, bc); break;
            case 3: box_tile<3>(r, q, ft, tx, ty, bc); break;
            case 4: box_tile<4>(r, q, ft, tx, ty, bc); break;
            case 5: box_tile<5>(r, q, ft, tx, ty, bc); break;
            case 6: box_tile<6>(r, q, ft, tx, ty, bc); break;
            case 7: box_tile<7>(r, q, ft, tx, ty, bc); break;
            case 8: box_tile<8>(r, q, ft, tx, ty, bc); break;
            default: return -1;
            }
            const int side = std::max(bc.X1 - bc.X0, bc.Y1 - bc.Y0) + 1;
            if (side > ft.maxSide) ++bc.outside;                               // the LDS pitch's bound must hold too
        }
    if (seen) *seen = bc.seen;
    return bc.outside;
}

// cell_live_rows must be a superset: counts the cells that contribute something (non-zero area sums) although their row lies
// outside the interval reported for the 64-column strip they belong to.  -1: the cell formulation does not serve the request.
long aai_emu_cell_live_rows_check(const aai_request *rq)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -2;
    const RotLaunch r = make_rot_launch(g, rq->mode, rq->policy);
    if (!r.cell) return -1;
    EmuCells cells;
    std::vector<float> img((size_t)r.W * r.H, 1.0f);
    emu_cells(r, img.data(), r.W, cells);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    long bad = 0;
    for (int x0 = 0; x0 <= r.dW; x0 += 63) {
        int lo, hi;
        cell_live_rows(make_cell_live(r, z), x0, x0 + 63, lo, hi);
        for (int y = 0; y <= r.dH; ++y)
            for (int x = x0; x <= std::min(x0 + 63, r.dW); ++x) {
                const size_t at = (size_t)y * cells.W1 + x;
                const bool contributes = cells.a[0][at] != 0.f || cells.a[1][at] != 0.f || cells.a[2][at] != 0.f || cells.a[3][at] != 0.f || cells.unc[at];
                if (contributes && (y < lo || y > hi)) ++bad;
            }
    }
    return bad;
}

// The host-side class verification of an axis-aligned plan (aai_plan.cpp: axis_verify_by_class) against the per-pixel scan
// it replaces (aai_axis_verify_kernel = axis_pixel_differs for every dst pixel).  Returns -1 when the geometry does not
// qualify (inexact arithmetic), else the number of dst pixels on which the two disagree; counts of flagged pixels out.
long aai_emu_axis_class_verify(const aai_request *rq, long *byClass, long *byScan)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK || !g.axisAligned) return -2;
    const RotLaunch r = make_rot_launch(g, rq->mode, rq->policy);
    std::vector<std::pair<int, int>> flagged;
    bool dense = false;
    if (!axis_verify_by_class(r, flagged, dense, 1u << 30)) return -1;
    std::vector<char> a((size_t)r.dW * r.dH, 0);
    for (auto &f : flagged) a[(size_t)f.second * r.dW + f.first] = 1;
    long bad = 0, n = 0;
    for (int y = 0; y < r.dH; ++y)
        for (int x = 0; x < r.dW; ++x) {
            const bool d = rq->mode == AAI_MODE_FAST ? axis_pixel_differs_fast(r, x, y) : axis_pixel_differs(r, x, y);
            n += d ? 1 : 0;
            if (d != (a[(size_t)y * r.dW + x] != 0)) ++bad;
        }
    *byClass = (long)flagged.size(); *byScan = n;
    return bad;
}

int aai_emu_cell_parts(const aai_request *rq, int dx, int dy, const float *img, double *out32, double *out64)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -1;
    const RotLaunch r = make_rot_launch(g, AAI_MODE_AREA, rq->policy);
    if (!r.cell) return -2;
    const int win = make_cell_consts<float>(r.side, r.c, r.s).win;
    switch (win) {
    case 2: emu_cell_parts<float, 2>(r, img, dx, dy, out32); emu_cell_parts<double, 2>(r, img, dx, dy, out64); break;
    case 3: emu_cell_parts<float, 3>(r, img, dx, dy, out32); emu_cell_parts<double, 3>(r, img, dx, dy, out64); break;
    case 4: emu_cell_parts<float, 4>(r, img, dx, dy, out32); emu_cell_parts<double, 4>(r, img, dx, dy, out64); break;
    case 5: emu_cell_parts<float, 5>(r, img, dx, dy, out32); emu_cell_parts<double, 5>(r, img, dx, dy, out64); break;
    case 6: emu_cell_parts<float, 6>(r, img, dx, dy, out32); emu_cell_parts<double, 6>(r, img, dx, dy, out64); break;
    default: return -3;
    }
    return win;
}
void aai_emu_force_rotated(int on) { g_forceRotated = on; }
void aai_emu_skip_axis_fixup(int on) { g_skipAxisFixup = on; }
long aai_emu_axis_fixups() { return g_axisFixups; }

int aai_emu_quad_pixel_debug(const aai_request *rq, int dx, int dy, const float *img, double *out4)
{
    Geometry g; std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -1;
    const RotLaunch r = make_rot_launch(g, AAI_MODE_AREA, rq->policy);
    double px, py; quad_centre(r, dx, dy, px, py);
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    switch (q.win) {
    case 3: emu_quad_sums<float, 3>(r, img, px, py, out4 + 0, out4 + 1); emu_quad_sums<double, 3>(r, img, px, py, out4 + 2, out4 + 3); break;
    case 4: emu_quad_sums<float, 4>(r, img, px, py, out4 + 0, out4 + 1); emu_quad_sums<double, 4>(r, img, px, py, out4 + 2, out4 + 3); break;
    case 5: emu_quad_sums<float, 5>(r, img, px, py, out4 + 0, out4 + 1); emu_quad_sums<double, 5>(r, img, px, py, out4 + 2, out4 + 3); break;
    default: return -2;
    }
    return 0;
}
void aai_emu_quad_stats(long *pixels, long *uncertain) { *pixels = g_quadPixels; *uncertain = g_quadUncertain; }

// Pair-by-pair comparison of the quad formulation (evaluated in DOUBLE) with the older fast path and with the
// strict replay of the reference, over every window position of every dst pixel of an area-mode request:
//   maxOld     largest |quad - old fast path| over all pairs
//   maxStrict  largest |quad - strict replay| over the pairs of dst pixels without a knife edge
// Returns the number of pairs compared, -1 when the geometry is not served by the quad formulation.
long aai_emu_quad_pair_check(const aai_request *rq, double *maxOld, double *maxStrict)
{
    Geometry g;
    std::string msg;
    *maxOld = *maxStrict = 0;
    if (make_geometry(*rq, g, msg) != AAI_OK || g.axisAligned) return -1;
    const RotLaunch r = make_rot_launch(g, AAI_MODE_AREA, rq->policy);
    if (!quad_supported(r.side, r.c, r.s)) return -1;
    const QuadConsts<double> q = make_quad_consts<double>(r.side, r.c, r.s, r.policy, r.scale);
    long n = 0;
    for (int dy = 0; dy < r.dH; ++dy)
        for (int dx = 0; dx < r.dW; ++dx) {
            double px, py;
            pixel_centre(r, dx, dy, px, py);
            const bool flagged = pixel_on_knife_edge(r, px, py, true);
            const double hb = r.h * (r.c + r.s);
            const int x0 = (int)std::floor(px - hb + 0.5 - AAI_KNIFE_GUARD), x1 = (int)std::ceil(px + hb - 0.5 + AAI_KNIFE_GUARD);
            const int y0 = (int)std::floor(py - hb + 0.5 - AAI_KNIFE_GUARD), y1 = (int)std::ceil(py + hb - 0.5 + AAI_KNIFE_GUARD);
            SVec sv4[4];
            strict_vertices(r, dx, dy, sv4);
            int vX[4], vY[4]; double vfx[4], vfy[4];
            for (int v = 0; v < 4; ++v) {
                const double wx = px + q.ox[v], wy = py + q.oy[v];
                vX[v] = (int)std::floor(wx + 0.5); vY[v] = (int)std::floor(wy + 0.5);
                vfx[v] = wx - vX[v]; vfy[v] = wy - vY[v];
            }
            for (int Y = y0; Y <= y1; ++Y)
                for (int X = x0; X <= x1; ++X) {
                    const double ex = X - px, ey = Y - py;
                    const double a = ex * r.c - ey * r.s, b = ex * r.s + ey * r.c;
                    // older fast path
                    double d = 0; bool e1 = false, e2 = false;
                    const int cls = classify_pair<false>(r, a, b, d, e1);
                    double old = 0;
                    if (cls == PAIR_INSIDE) old = 1;
                    else if (cls == PAIR_GENERAL) old = wedge_pair_area<false>(r, px - (X - 0.5), py - (Y - 0.5), a < 0.0, b < 0.0, r.policy, e2);
                    else if (cls != PAIR_OUTSIDE) old = single_cut_area<false>(r, d, cls == PAIR_CUT_LR, r.policy, e2);
                    // quad formulation
                    const double A = q.h - std::fabs(a), B = q.h - std::fabs(b);
                    const double mn = std::min(A, B), mx = std::max(A, B);
                    double area = 0;
                    int vtx = -1;
                    for (int v = 0; v < 4; ++v) if (vX[v] == X && vY[v] == Y) vtx = v;
                    if (vtx >= 0) area = quad_vertex_area(q, vfx[vtx], vfy[vtx], vtx);
                    else if (mn <= -q.k) area = 0;
                    else if (mn >= q.k) area = 1;
                    else if (mx >= q.k) area = quad_cut(q, std::min(std::max(mn + q.k, 0.0), q.k2), A < B && q.ref);
                    else {
                        double nearS;
                        const double tA = std::min(std::max(A + q.k, 0.0), q.k2), tB = std::min(std::max(B + q.k, 0.0), q.k2);
                        area = quad_double<double, false>(q, A, B, std::min(tA, q.k2 - tA), tA > q.k, std::min(tB, q.k2 - tB), tB > q.k, (a < 0) == (b < 0), nearS);
                    }
                    *maxOld = std::max(*maxOld, std::fabs(area - old));
                    if (!flagged) *maxStrict = std::max(*maxStrict, std::fabs(area - strict_pair_area(sv4, X, Y, r.policy)));
                    ++n;
                }
        }
    return n;
}
void aai_emu_set_strict(int on) { g_strict = on; }
void aai_emu_knife_stats(long *pairs, long *pixels) { *pairs = g_knifePairs; *pixels = g_knifePixels; }
long aai_emu_missed_knife_pairs(void) { return g_missedPairs; }

// Per-pair dump for one dst pixel of the rotated path: class, knife flag, fast area, strict area.
int aai_emu_pixel_pairs(const aai_request *rq, int dx, int dy, int cap, int *xs, int *ys, int *cls, int *knife, double *fast, double *strict)
{
    Geometry g;
    std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -1;
    const RotLaunch r = make_rot_launch(g, rq->mode, rq->policy);
    double px, py;
    pixel_centre(r, dx, dy, px, py);
    const double hb = r.h * (r.c + r.s);
    const int x0 = std::max(0, (int)std::floor(px - hb + 0.5 - AAI_KNIFE_GUARD)), x1 = std::min(r.mW - 1, (int)std::ceil(px + hb - 0.5 + AAI_KNIFE_GUARD));
    const int y0 = std::max(0, (int)std::floor(py - hb + 0.5 - AAI_KNIFE_GUARD)), y1 = std::min(r.mH - 1, (int)std::ceil(py + hb - 0.5 + AAI_KNIFE_GUARD));
    SVec sv4[4];
    strict_vertices(r, dx, dy, sv4);
    int n = 0;
    for (int Y = y0; Y <= y1; ++Y)
        for (int X = x0; X <= x1; ++X) {
            const double ex = X - px, ey = Y - py;
            const double a = ex * r.c - ey * r.s, b = ex * r.s + ey * r.c;
            double d = 0;
            bool edgy = false, edgy2 = false;
            const int c = classify_pair<true>(r, a, b, d, edgy);
            double area = 0;
            if (c == PAIR_INSIDE) area = 1;
            else if (c == PAIR_GENERAL) area = wedge_pair_area<true>(r, px - (X - 0.5), py - (Y - 0.5), a < 0.0, b < 0.0, r.policy, edgy2);
            else if (c != PAIR_OUTSIDE) area = single_cut_area<true>(r, d, c == PAIR_CUT_LR, r.policy, edgy2);
            if (n < cap) { xs[n] = X; ys[n] = Y; cls[n] = c; knife[n] = (edgy || edgy2) ? 1 : 0; fast[n] = area; strict[n] = strict_pair_area(sv4, X, Y, r.policy); ++n; }
        }
    return n;
}

// Planner invariants for a fuzzed geometry (axis-aligned requests only): returns 0 if all hold, else a code.
int aai_emu_axis_invariants(const aai_request *rq)
{
    Geometry g;
    std::string msg;
    if (make_geometry(*rq, g, msg) != AAI_OK) return -1;
    if (!g.axisAligned) return -2;
    AxisTables t;
    build_axis_tables(g, rq->mode, t);
    if (t.nA != (t.transposed ? g.dH : g.dW) || t.nB != (t.transposed ? g.dW : g.dH)) return 1;
    auto entry_ok = [](const AxisEntry &e, int extent) {
        if (e.s0 < 0 || e.s1 < e.s0 || e.s1 >= extent) return false;
        if (e.wFirst < 0.f || e.wMid < 0.f || e.wLast < 0.f) return false;
        const double total = (double)e.wFirst + (e.s1 > e.s0 ? (double)e.wLast + (double)e.wMid * (e.s1 - e.s0 - 1) : 0.0);
        const bool empty = e.wFirst == 0.f && e.wMid == 0.f && e.wLast == 0.f;
        return empty || std::fabs(total - 1.0) < 1e-5;
    };
    for (const auto &en : t.lane) if (!entry_ok(en, g.W)) return 2;
    for (const auto &en : t.row) if (!entry_ok(en, g.H)) return 3;
    // strips partition the lane axis in order, hold <= 256 outputs, and contain their windows
    int next = 0;
    for (const auto &s : t.strips) {
        if (s.k0 != next || s.k1 <= s.k0 || s.k1 - s.k0 > 256) return 4;
        for (int k = s.k0; k < s.k1; ++k)
            if (!t.wide && (t.lane[k].s0 < s.x0 || t.lane[k].s1 >= s.x0 + STRIP_COLS)) return 5;
        next = s.k1;
    }
    if (next != t.nA) return 6;
    // interleaved channels: entry k = pixel * C + channel over the ELEMENTS of the source row; strips still partition the
    // lane axis, break on pixel boundaries only, hold <= 256 outputs and contain every window they own (windows of
    // neighbouring pixels' channels interleave, and parked empties sit below their successors: the binding extent is
    // not the last entry's)
    for (int C = 2; C <= 4; ++C) {
        AxisTables tc;
        build_axis_tables(g, rq->mode, tc, C);
        if (tc.nA != t.nA * C || tc.channels != C) return 11;
        int nx = 0;
        for (const auto &st : tc.strips) {
            if (st.k0 != nx || st.k1 <= st.k0 || st.k1 - st.k0 > 256 || st.k0 % C != 0 || st.k1 % C != 0) return 12;
            for (int k = st.k0; k < st.k1; ++k) {
                const AxisEntry &e = tc.lane[k];
                if (e.s0 % C != k % C && !(e.wFirst == 0.f && e.wMid == 0.f && e.wLast == 0.f)) return 13;       // taps stay on their channel
                if (!tc.wide && (e.s0 < st.x0 || e.s1 >= st.x0 + STRIP_COLS)) return 14;
                if (e.s1 >= g.W * C) return 15;
            }
            nx = st.k1;
        }
        if (nx != tc.nA) return 16;
    }
    // band slicing: three bands re-create the table of dst rows, re-based to their own first source row
    const int n = g.dH;
    for (int part = 0; part < 3 && n >= 3; ++part) {
        const int r0 = part * n / 3, r1 = (part + 1) * n / 3;
        AxisTables b;
        build_axis_tables(g, rq->mode, b);
        int a = 0, z = 0;
        restrict_axis_tables_to_band(g, b, r0, r1, a, z);
        if (a < 0 || z > g.H || a >= z) return 7;
        const std::vector<AxisEntry> &full = t.transposed ? t.lane : t.row, &band = b.transposed ? b.lane : b.row;
        const bool flip = t.transposed ? t.flipA : t.flipB;
        if ((int)band.size() != r1 - r0) return 8;
        for (int i = 0; i < r1 - r0; ++i) {
            const AxisEntry &f = full[(flip ? n - r1 : r0) + i], &q = band[i];
            const bool empty = f.wFirst == 0.f && f.wMid == 0.f && f.wLast == 0.f;
            if (q.wFirst != f.wFirst || q.wMid != f.wMid || q.wLast != f.wLast) return 9;
            if (!empty && !t.transposed && (q.s0 + a != f.s0 || q.s1 + a != f.s1)) return 10;
        }
    }
    return 0;
}

// Strip table introspection for the planner tests.
int aai_emu_strip_stats(const aai_request *rq, int *nStrips, int *maxOutputsPerStrip, int *wide, int *maxRowSpan)
{
    Geometry g;
    std::string msg;
    int rc = make_geometry(*rq, g, msg);
    if (rc != AAI_OK) return rc;
    if (!g.axisAligned) return AAI_ERR_BAD_ARGUMENT;
    AxisTables t;
    build_axis_tables(g, rq->mode, t);
    *nStrips = (int)t.strips.size();
    int m = 0;
    for (const auto &s : t.strips) {
        m = std::max(m, s.k1 - s.k0);
        for (int k = s.k0; k < s.k1; ++k)
            if (!t.wide && (t.lane[k].s0 < s.x0 || t.lane[k].s1 >= s.x0 + STRIP_COLS)) return -100;   // window escapes its strip
    }
    *maxOutputsPerStrip = m; *wide = t.wide ? 1 : 0; *maxRowSpan = t.maxRowSpan;
    return AAI_OK;
}

}  // extern "C"
