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

using namespace aai;

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

static void emu_rotated(const Geometry &g, const aai_request &rq, const float *img, int64_t srcStride, float *dst, int64_t dstStride)
{
    RotLaunch r{};
    r.fracX = g.fracX; r.fracY = g.fracY; r.side = g.side; r.isoX = g.isoX; r.isoY = g.isoY;
    r.offX = g.offX; r.offY = g.offY; r.sn = g.sn; r.cs = g.cs;
    r.reach = g.side * std::sqrt(2.0) / 2 + 1;
    r.dW = g.dW; r.dH = g.dH; r.mW = g.mW; r.mH = g.mH; r.W = g.W; r.H = g.H;
    r.scale = g.scale; r.quadrant = g.quadrant; r.mode = rq.mode; r.policy = rq.policy;
    for (int dy = 0; dy < r.dH; ++dy)
        for (int dx = 0; dx < r.dW; ++dx) {
            Frame f;
            pixel_centre(r, dx, dy, f.px, f.py);
            frame_init(f, r);
            const double hb = f.h * (f.c + f.s);
            const int x0 = std::max(0, (int)std::floor(f.px - hb + 0.5)), x1 = std::min(r.mW - 1, (int)std::ceil(f.px + hb - 0.5));
            const int y0 = std::max(0, (int)std::floor(f.py - hb + 0.5)), y1 = std::min(r.mH - 1, (int)std::ceil(f.py + hb - 0.5));
            float *out = dst + (int64_t)dy * dstStride + dx;
            if (rq.mode == AAI_MODE_FAST) {
                const double lim = f.h + DBL_EPSILON * r.side;
                int count = 0; double acc = 0;
                for (int Y = y0; Y <= y1; ++Y)
                    for (int X = x0; X <= x1; ++X) {
                        const double ex = X - f.px, ey = Y - f.py;
                        const double a = ex * f.c - ey * f.s, b = ex * f.s + ey * f.c;
                        if (std::fabs(a) <= lim && std::fabs(b) <= lim) { ++count; acc += (double)img[virt_offset(r, X, Y, srcStride)]; }
                    }
                *out = count > 0 ? (float)(acc / count) : 0.f;
                continue;
            }
            const double k = 0.5 * (f.c + f.s);
            const double inner = f.h - k - 1e-9, outer = f.h + k + 1e-9;
            double sumA = 0, sumVA = 0;
            for (int Y = y0; Y <= y1; ++Y)
                for (int X = x0; X <= x1; ++X) {
                    const double ex = X - f.px, ey = Y - f.py;
                    const double a = std::fabs(ex * f.c - ey * f.s), b = std::fabs(ex * f.s + ey * f.c);
                    const double m = std::fmax(a, b);
                    if (m >= outer) continue;
                    double area = 1.0;
                    if (!(m <= inner)) area = pair_area(f, f.px - (X - 0.5), f.py - (Y - 0.5), r.policy);
                    if (area > 0.0) { sumA += area; sumVA += area * (double)img[virt_offset(r, X, Y, srcStride)]; }
                }
            *out = DBL_EPSILON < std::fabs(sumA) ? (float)(sumVA / sumA) : 0.f;
        }
}

extern "C" {

// Returns an AAI_* status; on success fills dW/dH and writes dW*dH floats to dst (if dst != NULL).
int aai_emu_resample(const aai_request *rq, const float *src, float *dst, int *dW, int *dH, int *usedAxisPath)
{
    Geometry g;
    std::string msg;
    int rc = make_geometry(*rq, g, msg);
    if (rc != AAI_OK) return rc;
    *dW = g.dW; *dH = g.dH;
    const bool axis = g.axisAligned && (rq->mode == AAI_MODE_AREA || rq->mode == AAI_MODE_FAST);
    if (usedAxisPath) *usedAxisPath = axis ? 1 : 0;
    if (!dst || !g.dW || !g.dH) return AAI_OK;
    if (axis) emu_axis(g, rq->mode, src, g.W, dst, g.dW);
    else emu_rotated(g, *rq, src, g.W, dst, g.dW);
    return AAI_OK;
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
