// aai_plan.cpp -- host-side validation, affine set-up and the separable tables for the axis-aligned
// kernel.  Double precision throughout; mirrors Source.cpp:112-305 of the reference.
#include "aai_plan.hpp"
#include "aai_rot_quad.hpp"
#include "aai_rot_cell.hpp"
#include "aai_axis_verify.hpp"
#include "aai_axis_verify.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>

namespace aai {

namespace {
constexpr double kEps = DBL_EPSILON;
constexpr double kPi = 3.14159265358979323846;   // M_PI
constexpr double kMaxExtent = 1073741824.0;       // 2^30: keeps every index below 31 bits

bool finite_all(std::initializer_list<double> v)
{
    for (double x : v) if (!std::isfinite(x)) return false;
    return true;
}
}  // namespace

int make_geometry(const aai_request &rq, Geometry &g, std::string &msg)
{
    // The reference's four checks, in its order and with its texts (Source.cpp:112-132).
    if (kEps < std::fabs(rq.src_res_x - rq.src_res_y) || kEps < std::fabs(rq.dst_res_x - rq.dst_res_y)) {
        msg = "Assumed X & Y resolution are same.";
        return AAI_ERR_RESOLUTION_MISMATCH;
    }
    if (rq.src_res_x <= kEps || rq.dst_res_x <= kEps) {
        msg = "0 or negative resolution is not acceptable.";
        return AAI_ERR_RESOLUTION_NONPOSITIVE;
    }
    if (rq.src_height <= 0) { msg = "There is no data in src array."; return AAI_ERR_NO_ROWS; }
    if (rq.src_width <= 0) { msg = "There is no data in the second dimension of src array."; return AAI_ERR_NO_COLUMNS; }
    // The reference lets NaN/Inf through all comparisons above and then hits undefined behaviour at
    // Source.cpp:139 (double -> unsigned of a non-finite value); reject instead.
    if (!finite_all({rq.src_res_x, rq.src_res_y, rq.dst_res_x, rq.dst_res_y, rq.src_iso_x, rq.src_iso_y, rq.rotation_deg})) {
        msg = "Non-finite argument.";
        return AAI_ERR_NONFINITE;
    }

    Geometry o;
    o.W = rq.src_width; o.H = rq.src_height;

    // Source.cpp:139 -- integer pre-expansion so that the dst pixel side exceeds sqrt(2) virtual pixels
    const double scaleReal = rq.dst_res_x / rq.src_res_x * std::sqrt(2.0) + 1 + kEps;
    if (!(scaleReal < 65536.0)) { msg = "Expansion ratio too large."; return AAI_ERR_TOO_LARGE; }
    o.scale = (int)(unsigned)scaleReal;

    // Source.cpp:141-148 -- wrap to [0,360) (closed form: the reference loops +-360), split off the quadrant
    double ang = rq.rotation_deg;
    if (std::fabs(ang) > 1e9) { msg = "Rotation angle magnitude too large."; return AAI_ERR_BAD_ARGUMENT; }
    while (ang < 0) ang += 360;
    while (360 <= ang) ang -= 360;
    if (ang < 90) o.quadrant = 0;
    else if (ang < 180) { o.quadrant = 1; ang -= 90; }
    else if (ang < 270) { o.quadrant = 2; ang -= 180; }
    else { o.quadrant = 3; ang -= 270; }
    o.angle = ang;
    o.sn = std::sin(ang / 180.0 * kPi);
    o.cs = std::cos(ang / 180.0 * kPi);

    // Source.cpp:150-156
    const double vW = (double)((o.quadrant & 1) ? o.H : o.W) * o.scale;
    const double vH = (double)((o.quadrant & 1) ? o.W : o.H) * o.scale;
    if (vW >= kMaxExtent || vH >= kMaxExtent) { msg = "Virtual source too large."; return AAI_ERR_TOO_LARGE; }
    o.mW = (int)vW; o.mH = (int)vH;

    // Source.cpp:173-186 (the isocenter is deliberately NOT pre-rotated: quirk A.5)
    o.isoX = rq.src_iso_x * o.scale + (o.scale - 1) / 2.0;
    o.isoY = rq.src_iso_y * o.scale + (o.scale - 1) / 2.0;
    const double sres = rq.src_res_x * o.scale;
    o.ratio = rq.dst_res_x / sres;
    o.side = sres / rq.dst_res_x;
    const double dWr = std::round((o.mW * std::fabs(o.cs) + o.mH * std::fabs(o.sn)) * o.ratio);
    const double dHr = std::round((o.mW * std::fabs(o.sn) + o.mH * std::fabs(o.cs)) * o.ratio);
    if (!(dWr < kMaxExtent) || !(dHr < kMaxExtent) || dWr * dHr >= 1099511627776.0 /* 2^40 */) {
        msg = "Output image too large."; return AAI_ERR_TOO_LARGE;
    }
    o.dW = (int)dWr; o.dH = (int)dHr;
    // The reference runs into undefined behaviour (a crash in practice) when either extent rounds to zero: its
    // edge-line tables index row / column dstSize - 1 (Source.cpp:243-305).  Report it instead.
    if (o.dW <= 0 || o.dH <= 0) { msg = "Output image would be empty."; return AAI_ERR_EMPTY_OUTPUT; }
    const double dix = (o.isoX * o.cs + (o.mH - o.isoY) * o.sn) * o.ratio;
    const double diy = (o.isoX * o.sn + o.isoY * o.cs) * o.ratio;
    if (std::fabs(dix) >= 2147483647.0 || std::fabs(diy) >= 2147483647.0) {
        msg = "Isocenter out of range."; return AAI_ERR_TOO_LARGE;
    }
    o.fracX = dix - (int)dix;
    o.fracY = diy - (int)diy;
    o.dIsoX = (int)dix;
    o.dIsoY = (int)diy;

    // Source.cpp:187-200
    const double wm1 = (double)(o.mW - 1), hm1 = (double)(o.mH - 1);
    double ox = 0, oy = 0;
    ox = std::min(ox, -o.isoX * o.cs + o.isoY * o.sn + o.isoX);
    oy = std::min(oy, -o.isoX * o.sn - o.isoY * o.cs + o.isoY);
    ox = std::min(ox, (wm1 - o.isoX) * o.cs + o.isoY * o.sn + o.isoX);
    oy = std::min(oy, (wm1 - o.isoX) * o.sn - o.isoY * o.cs + o.isoY);
    ox = std::min(ox, -o.isoX * o.cs - (hm1 - o.isoY) * o.sn + o.isoX);
    oy = std::min(oy, -o.isoX * o.sn + (hm1 - o.isoY) * o.cs + o.isoY);
    ox = std::min(ox, (wm1 - o.isoX) * o.cs - (hm1 - o.isoY) * o.sn + o.isoX);
    oy = std::min(oy, (wm1 - o.isoX) * o.sn + (hm1 - o.isoY) * o.cs + o.isoY);
    o.offX = ox; o.offY = oy;

    // Source.cpp:229-240
    o.lt45 = ang < 45;
    if (o.lt45) { o.tsn = o.sn; o.tcs = o.cs; o.ttn = std::tan(ang / 180.0 * kPi); }
    else {
        o.tsn = std::sin((ang - 90) / 180.0 * kPi);
        o.tcs = std::cos((ang - 90) / 180.0 * kPi);
        o.ttn = std::tan((ang - 90) / 180.0 * kPi);
    }
    if (std::fabs(o.ttn) < kEps) o.ttn = 0;
    o.axisAligned = o.lt45 && o.ttn == 0;
    g = o;
    return AAI_OK;
}

RotLaunch make_rot_launch(const Geometry &g, int mode, int policy)
{
    RotLaunch r{};
    r.fracX = g.fracX; r.fracY = g.fracY; r.side = g.side; r.isoX = g.isoX; r.isoY = g.isoY;
    r.offX = g.offX; r.offY = g.offY; r.sn = g.sn; r.cs = g.cs;
    r.reach = g.side * std::sqrt(2.0) / 2 + 1;
    r.dW = g.dW; r.dH = g.dH; r.mW = g.mW; r.mH = g.mH; r.W = g.W; r.H = g.H;
    r.scale = g.scale; r.quadrant = g.quadrant; r.mode = mode; r.policy = policy & AAI_POLICY_RULE_MASK;
    r.preferCell = (policy & AAI_POLICY_PREFER_CELL) ? 1 : 0; r.noFixup = (policy & AAI_POLICY_DIAG_NO_FIXUP) ? 1 : 0;
    r.dyBase = 0; r.dyEnd = g.dH; r.srcRow0 = 0; r.srcRow1 = g.H; r.chan = 1;
    r.invScale = 1.0 / g.scale;
    const double c = g.cs, s = g.sn, h = 0.5 * g.side;
    r.c = c; r.s = s; r.h = h;
    r.o0x = -h * (c + s); r.o0y = h * (s - c);
    r.o1x = h * (c - s);  r.o1y = -h * (s + c);
    // the area kernels only run with s > 0 and c > 0 (reduced angle 0 goes to the axis-aligned kernel);
    // the samplers ignore these fields
    r.m1 = s / c;  r.im1 = c / s;
    r.m2 = -c / s; r.im2 = -s / c;
    r.Lc = g.side * c; r.Ls = g.side * s;
    r.rLc = 1.0 / r.Lc; r.rLs = 1.0 / r.Ls;
    r.k = 0.5 * (c + s);
    r.lo = std::min(c, s); r.hi = std::max(c, s);
    r.rc = 1.0 / c; r.rs = 1.0 / s; r.rhi = 1.0 / r.hi; r.r2cs = 1.0 / (2.0 * c * s);
    r.lt45 = g.lt45 ? 1 : 0; r.tsn = g.tsn; r.tcs = g.tcs; r.ttn = g.ttn;
    // Footprints with an interior at least kRunsMinInterior pixels wide walk rows as runs (aai_rotated_runs_kernel);
    // the threshold is from profiles/r01_rotated_runs.txt.  Only without replication (scale 1: rows are straight lines).
    r.runs = (mode == AAI_MODE_AREA && g.scale == 1 && 2.0 * (h - r.k) >= kRunsMinInterior) ? 1 : 0;
    // Footprints whose window of source pixels fits 8 x 8 take the fp32 quad formulation (aai_rot_quad.hpp), unless the
    // reduced angle is so close to an axis that the reference's own corner-triangle rule amplifies fp32 coordinates
    // beyond the parity bar (quad_supported).
    r.quad = ((mode == AAI_MODE_AREA || mode == AAI_MODE_FAST) && !(policy & AAI_POLICY_DOUBLE_PRECISION) && c > 0.0 && s > 0.0 &&
              quad_supported(g.side, c, s)) ? 1 : 0;
    // (fast mode's window holds pixel centres only and is two positions narrower: one 8 x 8 window reaches a little further)
    if (mode == AAI_MODE_FAST && !(policy & AAI_POLICY_DOUBLE_PRECISION) && c > 0.0 && s > 0.0 && quad_fast_parts(g.side, c, s) == 1) r.quad = 1;
    // (the window kernels map a replicated lattice back to source pixels with fp32 quotients, exact below 2^22 pixels a side:
    // QuadSrc::issue; a wider replicated lattice -- a source of 700,000 pixels a side and more -- stays in double precision)
    if (g.scale > 1 && (g.mW >= (1 << 22) || g.mH >= (1 << 22))) r.quad = 0;
    // (Fast mode with replication used to stay on the fp64 line-walking kernel: the 4 x 4 ... 5 x 5 area-mode window fetched
    // too much.  With the centre-only window of round 3 -- 3 x 3 at x4 up-sampling -- the window kernel wins everywhere
    // measured: x4 at 45 degrees 2.75 -> 2.67 ms, x2 at 30 0.68 -> 0.59, 1:1 at 61 0.212 -> 0.204, x3 at 17.5 0.37 -> 0.29:
    // tools/fast_scaled_ab.sh.)
    // Area mode: the cell formulation (aai_rot_cell.hpp) evaluates every (dst, src) pair once instead of once per dst pixel
    r.cell = (r.quad && mode == AAI_MODE_AREA && cell_supported(g.side, c, s)) ? 1 : 0;
    // Area mode, footprints too wide for one 8 x 8 window (ratios above ~5.5 : 1): the same fp32 arithmetic over a window split
    // into 2 x 2 or 4 x 4 parts, a lane per part (aai_rotated_wide.hip) -- the double-precision runs kernel was bound by its
    // boundary pairs (8 : 1 at 17.5 degrees: 0.34 ms where the source is read in 0.05)
    r.wide = (mode == AAI_MODE_AREA && !(policy & AAI_POLICY_DOUBLE_PRECISION) && c > 0.0 && s > 0.0 && g.scale == 1) ? quad_wide_parts(g.side, c, s) : 0;
    // ... and fast mode likewise (aai_wide_fast_kernel: memberships instead of areas, the window in registers): the double-precision
    // line-walking kernel took 0.26 ms for 8:1 at 17.5 degrees, more than the area mode's wide kernel
    if (mode == AAI_MODE_FAST && !(policy & AAI_POLICY_DOUBLE_PRECISION) && c > 0.0 && s > 0.0 && g.scale == 1 && quad_fast_parts(g.side, c, s) > 1)
        r.wide = quad_fast_parts(g.side, c, s);
    {
        // virtual centre: X = dx (side cs) + dy (side sn) + X0, Y = -dx (side sn) + dy (side cs) + Y0   (pixel_centre)
        const double u0 = g.fracX * g.side - g.isoX + g.offX, v0 = g.fracY * g.side - g.isoY + g.offY;
        const double Xa = g.side * g.cs, Xb = g.side * g.sn, X0 = u0 * g.cs + v0 * g.sn + g.isoX;
        const double Ya = -g.side * g.sn, Yb = g.side * g.cs, Y0 = -u0 * g.sn + v0 * g.cs + g.isoY;
        r.cXa = Xa; r.cXb = Xb; r.cX0 = X0; r.cYa = Ya; r.cYb = Yb; r.cY0 = Y0;
        const double is = 1.0 / g.scale;
        // continuous virtual -> continuous original coordinates per quadrant, then (v + 1/2) / scale - 1/2
        double xa, xb, x0, ya, yb, y0;       // virtual-lattice combination that feeds source x / source y
        switch (g.quadrant) {
        default:
        case 0: xa = Xa; xb = Xb; x0 = X0;               ya = Ya; yb = Yb; y0 = Y0;               break;
        case 1: xa = Ya; xb = Yb; x0 = Y0;               ya = -Xa; yb = -Xb; y0 = g.mW - 1 - X0;  break;
        case 2: xa = -Xa; xb = -Xb; x0 = g.mW - 1 - X0;  ya = -Ya; yb = -Yb; y0 = g.mH - 1 - Y0;  break;
        case 3: xa = -Ya; xb = -Yb; x0 = g.mH - 1 - Y0;  ya = Xa; yb = Xb; y0 = X0;               break;
        }
        r.sAx = xa * is; r.sBx = xb * is; r.sCx = (x0 + 0.5) * is - 0.5;
        r.sAy = ya * is; r.sBy = yb * is; r.sCy = (y0 + 0.5) * is - 0.5;
    }
    return r;
}

QuadMap make_quad_map(const Geometry &g, int64_t rowStride, int srcRow0, int channels, int elementBytes)
{
    QuadMap m{};
    m.nX = g.mW / g.scale; m.nY = g.mH / g.scale;
    m.base = -(int64_t)srcRow0 * rowStride;
    switch (g.quadrant) {
    default:
    case 0: m.strideX = channels;  m.flipX = 0; m.strideY = rowStride; m.flipY = 0; break;   // (X, Y)          -> (x, y) = (X, Y)
    case 1: m.strideX = rowStride; m.flipX = 1; m.strideY = channels;  m.flipY = 0; break;   // (Y, mW-1-X)
    case 2: m.strideX = channels;  m.flipX = 1; m.strideY = rowStride; m.flipY = 1; break;   // (mW-1-X, mH-1-Y)
    case 3: m.strideX = rowStride; m.flipX = 0; m.strideY = channels;  m.flipY = 1; break;   // (mH-1-Y, X)
    }
    m.scale = g.scale;
    m.invScale = (float)(1.0 / g.scale);
    m.invScaleD = 1.0 / g.scale;
    const int64_t bytes = ((int64_t)(g.H - 1) * rowStride + (int64_t)g.W * channels) * elementBytes;      // of the whole image
    m.lastLoad4 = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(bytes - 4, 0xffffffffll));
    m.lastLoad8 = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(bytes - 8, 0xffffffffll));
    m.mul24Ok = (m.nX < (1 << 23) && m.nY < (1 << 23) && rowStride * elementBytes < (1 << 23) && (int64_t)channels * elementBytes < (1 << 23)) ? 1 : 0;
    if (g.scale == 1 && channels == 1 && elementBytes == 4 && g.mW < (1 << 23) && g.mH < (1 << 23) && m.mul24Ok) {
        const uint32_t sxb = (uint32_t)m.strideX * 4u, syb = (uint32_t)m.strideY * 4u;
        m.fastOk = 1;
        m.fastAlongX = m.strideX == 1 ? 1 : 0;
        m.fastSX = m.flipX ? 0u - sxb : sxb;
        m.fastSY = m.flipY ? 0u - syb : syb;
        m.fastC0 = (m.flipX ? (uint32_t)(m.nX - 1) * sxb : 0u) + (m.flipY ? (uint32_t)(m.nY - 1) * syb : 0u);
        m.fastLine = m.fastAlongX ? m.fastSY : m.fastSX;
        m.fastRev4 = (m.fastAlongX ? m.flipX : m.flipY) ? 4u : 0u;
    }
    return m;
}

void rotated_band_source_rows(const Geometry &g, int row0, int row1, bool sampler, int &srcRow0, int &srcRow1)
{
    // virtual-lattice bounding box of the band: centres are affine in (dx,dy), so its four corner pixels bound it, and every kernel
    // family that may serve the request reaches a known distance beyond a pixel's centre -- the constants come from the headers
    // that hold the windows' formulas (rot_window_reach, quad_window_reach, quad_fast_window_reach, cell_window_reach), not from
    // prose; tests/emulation replays the windows of every family against the rows reported here (aai_emu_band_cover).
    // The cell kernel also evaluates the cells of dst row `row1` and dst column dW (their N / W parts belong to the band's last row /
    // column): the box spans cells [0, dW] x [row0, row1].  The samplers' taps reach 2 ORIGINAL pixels from the sample point.
    const double c = std::fabs(g.cs), s = std::fabs(g.sn);
    double hb = rot_window_reach(0.5 * g.side, c, s);                      // the double-precision kernels and the fix-up pass: always possible
    if (sampler) hb = 3.0 * g.scale;
    else if (c > 0.0 && s > 0.0) {
        if (quad_supported(g.side, c, s) || quad_wide_parts(g.side, c, s) || quad_fast_parts(g.side, c, s)) {
            const QuadConsts<float> q = make_quad_consts<float>(g.side, c, s, AAI_POLICY_REFERENCE, g.scale);
            if (quad_supported(g.side, c, s) || quad_wide_parts(g.side, c, s)) hb = std::max(hb, quad_window_reach(q));
            if (quad_fast_parts(g.side, c, s)) hb = std::max(hb, quad_fast_window_reach(q));
        }
        if (cell_supported(g.side, c, s)) hb = std::max(hb, cell_window_reach(make_cell_consts<float>(g.side, c, s)));
    }
    hb += 0.5;                                                             // (slack for the centres' own rounding)
    double minX = 1e300, maxX = -1e300, minY = 1e300, maxY = -1e300;
    for (int c = 0; c < 4; ++c) {
        double px, py;
        dst_centre(g, (c & 1) ? (sampler ? g.dW - 1 : g.dW) : 0, (c & 2) ? (sampler ? row1 - 1 : row1) : row0, px, py);
        minX = std::min(minX, px); maxX = std::max(maxX, px); minY = std::min(minY, py); maxY = std::max(maxY, py);
    }
    auto clampi = [](double v, int lo, int hi) { return (int)std::min((double)hi, std::max((double)lo, v)); };
    const int X0 = clampi(std::floor(minX - hb), 0, g.mW - 1), X1 = clampi(std::ceil(maxX + hb), 0, g.mW - 1);
    const int Y0 = clampi(std::floor(minY - hb), 0, g.mH - 1), Y1 = clampi(std::ceil(maxY + hb), 0, g.mH - 1);
    int a, b;      // source rows of the virtual box (Source.cpp:164-167)
    switch (g.quadrant) {
    default:
    case 0: a = Y0 / g.scale; b = Y1 / g.scale; break;
    case 1: a = (g.mW - 1 - X1) / g.scale; b = (g.mW - 1 - X0) / g.scale; break;
    case 2: a = (g.mH - 1 - Y1) / g.scale; b = (g.mH - 1 - Y0) / g.scale; break;
    case 3: a = X0 / g.scale; b = X1 / g.scale; break;
    }
    srcRow0 = std::max(0, std::min(a, b)); srcRow1 = std::min(g.H, std::max(a, b) + 1);
}

void rotated_live_spans(const RotLaunch &r, bool sampler, std::vector<int> &spans)
{
    spans.clear();
    if (!(r.side * r.cs > 1e-9 && r.side * r.sn > 1e-9) || r.dW <= 0 || r.dH <= 0) return;
    // area / fast: the footprint's bounding box reaches h (c + s) from the centre, windows are anchored at the nearest lattice
    // point; samplers: a sample point outside the image's extent is 0 (the taps' reach does not matter)
    const double rho = sampler ? 2.0 : r.h * (r.c + r.s) + 2.5;
    const int tileRows = (r.dH + 15) / 16, tilesX = (r.dW + 15) / 16;
    spans.resize((size_t)tileRows * 2);
    long dead = 0;
    for (int t = 0; t < tileRows; ++t) {
        int lo, hi;
        rot_live_cols(r, t * 16, std::min(t * 16 + 15, r.dH - 1), rho, lo, hi);
        if (lo > hi) { spans[2 * t] = 1; spans[2 * t + 1] = 0; dead += tilesX; continue; }
        spans[2 * t] = lo >> 4; spans[2 * t + 1] = hi >> 4;
        dead += (lo >> 4) + (tilesX - 1 - (hi >> 4));
    }
    if (dead * 16 < (long)tileRows * tilesX) spans.clear();      // fewer than 1/16 of the tiles: not worth a table
}

// ------------------------------------------------------------------------------------------------
// K1 tables.  With the reduced angle at zero every dst pixel is an axis-parallel box on the virtual
// lattice, so overlap areas factor into (x overlap) * (y overlap) and the reference's
// sum(v*a)/sum(a) (Source.cpp:573-577) becomes two 1-D normalised box filters.
// ------------------------------------------------------------------------------------------------
namespace {

struct VirtRange {        // inclusive virtual-pixel range with the weights of its two end pixels
    int a = 0, b = -1;    // empty when a > b
    double wa = 0, wb = 0;
};

// Edge k (k = 0..n) of the dst pixels along one axis, exactly as the reference's line tables place it
// (Source.cpp:243-305 with tmpTan == 0, then getIntersectionPoint, Source.cpp:962-985).
double edge_along_x(const Geometry &g, int k)
{
    double px, py;
    const double h = g.side / 2;
    if (k < g.dW) { dst_centre(g, k, 0, px, py); return px - h * (g.tcs + g.tsn); }
    dst_centre(g, g.dW - 1, 0, px, py);
    return px + h * (g.tcs - g.tsn);
}
double edge_along_y(const Geometry &g, int k)
{
    double px, py;
    const double h = g.side / 2;
    if (k < g.dH) { dst_centre(g, 0, k, px, py); return py - h * (g.tcs - g.tsn); }
    dst_centre(g, 0, g.dH - 1, px, py);
    return py + h * (g.tcs + g.tsn);
}

// AREA: overlap length of [lo,hi] with each virtual pixel [X-0.5, X+0.5], X in [0,m-1].
VirtRange area_range(double lo, double hi, int m)
{
    VirtRange r;
    const double clo = std::max(lo, -0.5), chi = std::min(hi, m - 0.5);
    if (!(clo < chi)) return r;
    int a = (int)std::floor(clo + 0.5), b = (int)std::floor(chi + 0.5);
    a = std::max(a, 0); b = std::min(b, m - 1);
    auto ov = [&](int X) { return std::max(0.0, std::min(chi, X + 0.5) - std::max(clo, X - 0.5)); };
    while (a <= b && !(ov(a) > 0)) ++a;
    while (b >= a && !(ov(b) > 0)) --b;
    if (a > b) return r;
    r.a = a; r.b = b; r.wa = ov(a); r.wb = ov(b);
    return r;
}

// FAST: virtual pixels whose CENTRE lies in the closed interval [lo,hi].  The reference
// decides this with ray/edge parameters compared against +-DBL_EPSILON (Source.cpp:857): the parameter
// along a dst edge is s = (X-lo)/(hi-lo) and must satisfy -eps < s < 1+eps; the ray parameter
// r = distance/100 must satisfy r > -eps.
VirtRange fast_range(double lo, double hi, int m)
{
    VirtRange r;
    if (!(hi > lo)) return r;
    auto inside = [&](int X) { return axis_centre_inside(lo, hi, X); };
    int a = (int)std::ceil(lo) - 1, b = (int)std::floor(hi) + 1;
    a = std::max(a, 0); b = std::min(b, m - 1);
    while (a <= b && !inside(a)) ++a;
    while (b >= a && !inside(b)) --b;
    if (a > b) return r;
    r.a = a; r.b = b; r.wa = 1; r.wb = 1;
    return r;
}

// Fold a virtual range onto original-image indices (each covers `scale` consecutive virtual pixels),
// optionally mirrored (quadrants 1-3 read an axis backwards, Source.cpp:165-167).
AxisEntry fold(const VirtRange &vr, int m, int scale, bool reversed, double otherAxisSpan)
{
    AxisEntry e{};
    e.s0 = 0; e.s1 = 0; e.wFirst = e.wMid = e.wLast = 0.f;
    if (vr.a > vr.b) return e;
    int a = vr.a, b = vr.b;
    double wa = vr.wa, wb = vr.wb;
    if (reversed) { const int na = m - 1 - b, nb = m - 1 - a; a = na; b = nb; std::swap(wa, wb); }
    const double total = (a == b) ? wa : wa + wb + (double)(b - a - 1);
    // Source.cpp:577: DBL_EPSILON < |sumArea| or the pixel is 0.  sumArea is the 2-D product; a row/column
    // of full pixels contributes about `otherAxisSpan`.
    if (!(total * otherAxisSpan > kEps)) return e;
    const int t0 = a / scale, t1 = b / scale;
    auto weight = [&](int t) {
        const int va = std::max(a, t * scale), vb = std::min(b, t * scale + scale - 1);
        double w = (double)(vb - va + 1);
        if (va == a) w += wa - 1;
        if (vb == b && !(a == b)) w += wb - 1;
        return w;
    };
    e.s0 = t0; e.s1 = t1;
    e.wFirst = (float)(weight(t0) / total);
    e.wLast = (t1 > t0) ? (float)(weight(t1) / total) : 0.f;
    e.wMid = (float)((double)scale / total);
    return e;
}

}  // namespace

// Greedy strips: consecutive lane-axis outputs whose windows fit in STRIP_COLS source columns (elements of a source
// row: with interleaved channels a pixel takes `channels` of them, and entry k = pixel * channels + channel).
// Windows mostly ascend, but not strictly: a parked empty entry sits on its predecessor's window, and the channel
// entries of neighbouring pixels interleave (pixel p+1's channel 0 starts below pixel p's channel 2 when both
// pixels share source columns), so the strip tracks the extent [lo, hi] of EVERYTHING it has accepted.  Strips break
// on pixel boundaries only (and hold at most 256 - 256 % channels outputs), so that the channels of one pixel never
// land in two strips.
static void build_axis_strips(AxisTables &t, int srcRowElements)
{
    t.strips.clear();
    t.maxOutputsPerStrip = 0;
    t.wide = srcRowElements < 4;      // the strip kernel loads whole 4-column vectors
    const int n = (int)t.lane.size();
    const int ch = t.channels > 1 ? t.channels : 1;
    // at most 256 outputs per strip: the kernel then keeps four window descriptions per lane in registers and stores
    // 16 bytes per lane (up-sampling would otherwise put >1000 outputs in a strip)
    const int maxOutputs = 256 - 256 % ch;
    int k = 0;
    while (k < n) {
        int lo = t.lane[k].s0, hi = t.lane[k].s1;
        int kk = k;
        while (kk < n && kk - k + ch <= maxOutputs) {
            // the next pixel: all its channel entries, or nothing
            int plo = lo, phi = hi;
            for (int c = 0; c < ch && kk + c < n; ++c) { plo = std::min(plo, t.lane[kk + c].s0); phi = std::max(phi, t.lane[kk + c].s1); }
            if (phi - plo + 1 > STRIP_COLS) break;
            lo = plo; hi = phi;
            kk += ch;
        }
        if (kk > n) kk = n;
        if (kk == k) { t.wide = true; kk = std::min(n, k + ch); lo = t.lane[k].s0; }   // a single pixel wider than a strip
        AxisStrip s{};
        s.k0 = k; s.k1 = kk; s.x0 = lo;
        t.maxOutputsPerStrip = std::max(t.maxOutputsPerStrip, s.k1 - s.k0);
        t.strips.push_back(s);
        k = kk;
    }
}

// Derived data of a (possibly band-restricted) pair of tables: parked empties, row statistics, strips.
static void finalize_axis_tables(const Geometry &g, AxisTables &t)
{
    t.nA = (int)t.lane.size(); t.nB = (int)t.row.size();
    // Empty entries (dst pixels off the image) carry s0 = s1 = 0; park them on a neighbour's window so
    // that strips stay compact.
    int lastS = 0;
    for (auto &e : t.lane) { if (e.wFirst == 0.f && e.wMid == 0.f && e.wLast == 0.f) { e.s0 = e.s1 = lastS; } else lastS = e.s0; }
    lastS = 0;
    for (auto &e : t.row) { if (e.wFirst == 0.f && e.wMid == 0.f && e.wLast == 0.f) { e.s0 = e.s1 = lastS; } else lastS = e.s0; }

    t.maxRowSpan = 0;
    for (const auto &e : t.row) t.maxRowSpan = std::max(t.maxRowSpan, e.s1 - e.s0 + 1);
    t.rowsShared = false;
    for (size_t i = 0; i + 1 < t.row.size(); ++i)
        if (t.row[i].wMid + t.row[i].wFirst > 0.f && t.row[i + 1].wMid + t.row[i + 1].wFirst > 0.f && t.row[i].s1 >= t.row[i + 1].s0) { t.rowsShared = true; break; }

    build_axis_strips(t, g.W * (t.channels > 1 ? t.channels : 1));
}

void build_axis_tables(const Geometry &g, int mode, AxisTables &t, int channels)
{
    // Which virtual axis reads which source axis (Source.cpp:164-167):
    //   q0: X -> src x (+), Y -> src y (+)      q1: X -> src y (-), Y -> src x (+)
    //   q2: X -> src x (-), Y -> src y (-)      q3: X -> src y (+), Y -> src x (-)
    const bool transposed = (g.quadrant & 1) != 0;
    const bool revX = (g.quadrant == 1 || g.quadrant == 2);
    const bool revY = (g.quadrant == 2 || g.quadrant == 3);

    std::vector<AxisEntry> alongX(g.dW), alongY(g.dH);
    double lo = edge_along_x(g, 0);
    for (int k = 0; k < g.dW; ++k) {
        const double hi = edge_along_x(g, k + 1);
        const VirtRange vr = (mode == AAI_MODE_FAST) ? fast_range(lo, hi, g.mW) : area_range(lo, hi, g.mW);
        alongX[k] = fold(vr, g.mW, g.scale, revX, g.side);
        lo = hi;
    }
    lo = edge_along_y(g, 0);
    for (int k = 0; k < g.dH; ++k) {
        const double hi = edge_along_y(g, k + 1);
        const VirtRange vr = (mode == AAI_MODE_FAST) ? fast_range(lo, hi, g.mH) : area_range(lo, hi, g.mH);
        alongY[k] = fold(vr, g.mH, g.scale, revY, g.side);
        lo = hi;
    }

    // Lane axis = the virtual axis that reads source x; put both tables in ascending source order.
    std::vector<AxisEntry> &laneSrc = transposed ? alongY : alongX;
    std::vector<AxisEntry> &rowSrc = transposed ? alongX : alongY;
    t.transposed = transposed;
    t.flipA = transposed ? revY : revX;     // lane-axis output index runs against source x
    t.flipB = transposed ? revX : revY;
    t.lane = laneSrc; t.row = rowSrc;
    if (t.flipA) std::reverse(t.lane.begin(), t.lane.end());
    if (t.flipB) std::reverse(t.row.begin(), t.row.end());
    t.nA = (int)t.lane.size(); t.nB = (int)t.row.size();

    finalize_axis_tables(g, t);
    t.channels = 1;
    if (channels > 1) {
        // Interleaved channels: the source row is W * channels elements wide and the lane
        // axis runs over its elements.  Entry (p, ch) = pixel entry p moved to the elements of channel ch: taps
        // `channels` elements apart (AxisLaunch::tapStep), same weights.  Windows stay ascending, so strips work as before.
        std::vector<AxisEntry> wideLane;
        wideLane.reserve(t.lane.size() * channels);
        for (const AxisEntry &e : t.lane)
            for (int ch = 0; ch < channels; ++ch) {
                AxisEntry x = e;
                x.s0 = e.s0 * channels + ch; x.s1 = e.s1 * channels + ch;
                wideLane.push_back(x);
            }
        t.lane.swap(wideLane);
        t.nA = (int)t.lane.size();
        t.channels = channels;
        build_axis_strips(t, g.W * channels);      // (t.channels is set: strips break on pixel boundaries)
    }
}

// Keep only the dst rows [row0,row1) (row bands of one image on different GPUs, or an
// image larger than device memory).  dst rows run along the row table (quadrants 0, 2) or along the lane table
// (quadrants 1, 3); the other table is kept whole.  Source rows are re-based to srcRow0 = the first source row any
// remaining window touches, which is returned together with the row after the last one.
void restrict_axis_tables_to_band(const Geometry &g, AxisTables &t, int row0, int row1, int &srcRow0, int &srcRow1, int extraRows)
{
    std::vector<AxisEntry> &rowsOfDst = t.transposed ? t.lane : t.row;     // table indexed by (possibly flipped) dst row
    const bool flip = t.transposed ? t.flipA : t.flipB;
    const int n = (int)rowsOfDst.size();
    const int k0 = flip ? n - row1 : row0, k1 = flip ? n - row0 : row1;
    rowsOfDst = std::vector<AxisEntry>(rowsOfDst.begin() + k0, rowsOfDst.begin() + k1);
    // source rows touched: the row table indexes source y in both layouts
    int lo = g.H, hi = -1;
    for (const auto &e : t.row)
        if (e.wFirst != 0.f || e.wMid != 0.f || e.wLast != 0.f) { lo = std::min(lo, e.s0); hi = std::max(hi, e.s1); }
    if (hi < lo) { lo = 0; hi = 0; }
    // (extraRows: the fix-up pass behind K1 may read a source row that only TOUCHES the band's footprint)
    lo = std::max(0, lo - extraRows); hi = std::min(g.H - 1, hi + extraRows);
    for (auto &e : t.row) { e.s0 = std::max(e.s0, lo) - lo; e.s1 = std::max(e.s1, lo) - lo; }
    srcRow0 = lo; srcRow1 = hi + 1;
    finalize_axis_tables(g, t);
}

// K1's separable model against the reference's classifier, on the HOST, for geometries whose arithmetic is exact.
// aai_axis_verify_kernel asks, for every dst pixel with a knife edge, whether the strict replay of the reference's
// classifier and the product of the two clipped extents agree pair by pair -- 6.6 ms at config 2, where EVERY pixel has
// a knife edge (every edge on a pixel boundary).  But at reduced angle 0 that answer depends on dst pixel (dx, dy) only
// through the position of column dx and of row dy RELATIVE to the source lattice and through whether their windows
// are clipped by the image: columns whose centre has the same fractional part and the same clipping are
// indistinguishable, as long as every coordinate is computed without rounding.  That holds when the side is a multiple
// of 1/256 and the offsets are multiples of 2^-20 of moderate size (integer and simple ratios with isocenters on
// half / quarter pixels: configs 1, 2, 4): then one representative per (column class, row class) is checked here -- a
// handful of evaluations instead of dW x dH -- and the flagged pixels are the products of the classes that differ.
// Returns false when the geometry does not qualify (the device scan runs instead).
static bool dyadic(double v, double scale, double limit)
{
    const double w = v * scale;
    return std::fabs(v) < limit && w == std::floor(w);
}
bool axis_verify_by_class(const RotLaunch &r, std::vector<std::pair<int, int>> &flagged, bool &dense, unsigned maxListed)
{
    const double two20 = 1048576.0;
    if (!(dyadic(r.side, 256.0, 256.0) && dyadic(r.fracX, two20, 2.0) && dyadic(r.fracY, two20, 2.0) && dyadic(r.isoX, two20, two20) &&
          dyadic(r.isoY, two20, two20) && dyadic(r.offX, two20, two20) && dyadic(r.offY, two20, two20)))
        return false;
    if (r.dW > 65536 || r.dH > 65536) return false;           // (index + offset) x side stays within 53 bits
    const double hb = r.h * (r.c + r.s);
    struct Cls { double frac; int clipLo, clipHi, last; int rep, count; };
    auto classes = [&](int n, int m, bool alongX, std::vector<Cls> &out, std::vector<int> &of) {
        of.resize(n);
        for (int i = 0; i < n; ++i) {
            double px, py;
            pixel_centre(r, alongX ? i : 0, alongX ? 0 : i, px, py);
            const double pc = alongX ? px : py;
            const double f = pc - std::floor(pc);
            // the window [pc - hb, pc + hb] against the lattice [0, m - 1]: how far it is clipped (0 = not at all)
            const int lo = (int)std::floor(pc - hb + 0.5 - AAI_KNIFE_GUARD), hi = (int)std::ceil(pc + hb - 0.5 + AAI_KNIFE_GUARD);
            const int clipLo = lo < 0 ? -lo : 0, clipHi = hi > m - 1 ? hi - (m - 1) : 0;
            int k = 0;
            for (; k < (int)out.size(); ++k)
                if (out[k].frac == f && out[k].clipLo == clipLo && out[k].clipHi == clipHi && out[k].last == (i == n - 1)) break;
            if (k == (int)out.size()) {
                if (out.size() >= 64) return false;
                out.push_back(Cls{f, clipLo, clipHi, i == n - 1 ? 1 : 0, i, 0});      // (the last column / row has its own edge formula)
            }
            ++out[k].count;
            of[i] = k;
        }
        return true;
    };
    std::vector<Cls> cx, cy;
    std::vector<int> ofx, ofy;
    if (!classes(r.dW, r.mW, true, cx, ofx) || !classes(r.dH, r.mH, false, cy, ofy)) return false;
    std::vector<char> differs(cx.size() * cy.size(), 0);
    uint64_t total = 0;
    for (size_t j = 0; j < cy.size(); ++j)
        for (size_t i = 0; i < cx.size(); ++i) {
            const bool d = r.mode == AAI_MODE_FAST ? axis_pixel_differs_fast(r, cx[i].rep, cy[j].rep) : axis_pixel_differs(r, cx[i].rep, cy[j].rep);
            differs[j * cx.size() + i] = d ? 1 : 0;
            if (d) total += (uint64_t)cx[i].count * cy[j].count;
        }
    flagged.clear();
    dense = total > maxListed;
    if (dense || !total) return true;
    flagged.reserve((size_t)total);
    for (int y = 0; y < r.dH; ++y)
        for (int x = 0; x < r.dW; ++x)
            if (differs[(size_t)ofy[y] * cx.size() + ofx[x]]) flagged.emplace_back(x, y);
    return true;
}


}  // namespace aai
