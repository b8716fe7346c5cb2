// aai_rot_cell.hpp -- third formulation of the rotated-lattice area average (K2): every (dst, src) pair's geometry is
// evaluated ONCE, by the cell of the dst grid that owns the source pixel, and shared between the dst pixels it feeds.
//
// Replaces Source.cpp:413-579 + 986-1431 of the reference, like aai_rot_quad.hpp, for every dst pixel that is not on one
// of the reference's DBL_EPSILON knife edges.  Same per-pair area formulas as the quad formulation (quad_cut_tp,
// quad_vertex_area); what changes is WHO evaluates a pair.
//
// The quad formulation gives each dst pixel a lane that visits every virtual source pixel its square can touch:
// (L + c + s)^2 of them, half of which it shares with its neighbours -- at config 5 (L = 1.5 at 45 degrees) 8.5 touched
// pixels per dst pixel of which 8 lie on an edge or a vertex of the square; at config 3 (L = 3) 18.  But in the dst frame
// the squares form a regular grid: a source pixel cut by the grid line between two dst pixels contributes g to one and
// 1 - g to the other (under the reference's corner rule too: types 2 and 4 of Source.cpp:1055-1062, 1083-1086 are
// complementary on the same legs), and a source pixel near a grid vertex G feeds the four dst pixels around G from ONE
// classification.  So the plane of pixel CENTRES is tiled by zones, one per cell (= dst pixel + its top-left grid vertex
// G): with (a', b') the centre's dst-frame coordinates relative to G and k = (c + s) / 2 the pixel's half extent along
// either dst axis, the zone is [-k, L - k) x [-k, L - k) and splits into
//   * a' < k and b' < k   "vertex zone": both grid lines through G cut the pixel.  Either it holds G -- its four wedge
//                         areas go to the four dst pixels around G (reference types 7-9) -- or exactly one vertical and
//                         one horizontal RAY from G cross it: the dst pixel between the two rays gets gA + gB - 1, the
//                         one across the vertical ray the left/right cut ALONE (the reference's corner rule applies),
//                         the one across the horizontal ray the top/bottom cut alone, the fourth nothing (types 2-6);
//   * a' < k <= b'        "left-edge zone": only the vertical grid line cuts it: g to this cell's dst pixel, 1 - g to its
//                         left neighbour, both with the reference's corner rule (types 2, 3, 4);
//   * b' < k <= a'        "top-edge zone": g to this cell's dst pixel, 1 - g to the one above (exact under both policies);
//   * k <= a', k <= b'    interior: area 1 to this cell's dst pixel.
// Every virtual pixel lies in exactly one zone, so a dst pixel costs L^2 pair evaluations (2.25 at config 5, 9 at
// config 3) and a window of floor(L (c + s)) + 1 lattice positions per axis (3 and 4) instead of 4 and 5.
// dst pixel (x, y) = own part of cell (x, y) + W part of cell (x + 1, y) + N part of cell (x, y + 1) + NW part of cell
// (x + 1, y + 1): the kernel (aai_rotated_cell.hip) keeps a wave on 64 consecutive cells of a row, walks down the rows and
// passes the parts between lanes / iterations in registers.
//
// Precision and decisions: as in the quad formulation everything is relative to a lattice point next to the zone's
// centre and runs in fp32 (hiPrec: the edges' t from double precision).  A zone boundary is a DECISION here -- a pixel
// claimed by two cells or by none is an error of its whole area -- so SCAN mode reports a cell when any of its
// decisions (zone and sub-zone membership, the reference's triangle / trapezoid switch, G inside a pixel, which rays
// cross) is closer than QuadConsts::margin to its threshold, and the plan leaves all four dst pixels that cell feeds to the
// double-precision fix-up pass.
#pragma once

#include "aai_rot_quad.hpp"

namespace aai {

// The shape of the cell kernel's wave.  1: 64 consecutive cell columns of ONE cell row per step (63 dst columns a strip).  2: 32 cell columns
// of TWO consecutive cell rows (31 dst columns a strip, the upper half-wave a row below the lower one): the two rows' windows share half
// of their source lines, which one load instruction then fetches once -- L1 -> L2 requests -44 %, fabric reads -54 % at config 3 -- at the
// price of row pieces of 124 bytes per store instead of 252 and ~8 % more instructions.  Measured (profiles/r04_cell_kernel.txt, 6.): shape 2
// wins from about 2.4:1 up (config 3 one image 158 -> 147-150 us, 8 images per launch 163 -> 130 us per image; 3:1 at 30 degrees 330 -> 281,
// 4:1 at 45 degrees 253 -> 184), is level or behind between 2:1 and 2.3:1 (2.2:1 at 75 degrees 213 -> 238), behind within a few degrees
// of an axis (2.39:1 at 2 degrees 211 -> 247: a row of 64 cells already runs along the source rows there) and wherever the stores weigh as
// much as the loads (1:1 at 30 degrees 824 -> 934 us, config 5's replicated source 2.65 -> 3.04 ms).
// (c, s = cos, sin of the reduced angle.)  The kernel's launcher and the CPU replay of its fetches (tests/emulation) both ask this function.
AAI_HD int cell_wave_rows(double side, int scale, double c, double s) { return scale <= 1 && side >= 2.35 && (c < s ? c : s) >= 0.1 ? 2 : 1; }
constexpr int cell_wave_lanes(int waveRows) { return 64 / waveRows; }              // cell columns a wave evaluates per cell row
constexpr int cell_wave_cols(int waveRows) { return 64 / waveRows - 1; }           // dst columns it completes (the last cell column only feeds its left neighbour)

enum CellTarget { CELL_O = 0, CELL_W = 1, CELL_N = 2, CELL_NW = 3 };     // own dst pixel (x, y); (x-1, y); (x, y-1); (x-1, y-1)

template <typename F>
struct CellConsts {
    F thr;                 // 2k - h: a zone coordinate below this puts the pixel centre within k of the grid line through G
    F hbz;                 // h (c + s) + guard: half extent of the lattice window that holds a zone
    int win;               // lattice positions per axis of that window, <= kQuadMaxWin
    // classification by intervals (cell_eval, pass 1): along a window row the zone coordinates grow with the column, so each threshold
    // is crossed at ONE column: az < X from column X / c + (row term) on.  X / c + 1/2 for X = -h, +h, thr and the same with 1 / s:
    F eALo, eAHi, eAV, eBLo, eBHi, eBH;
    double zx, zy;         // zone centre - dst pixel centre, virtual frame
    double gx, gy;         // G - zone centre, virtual frame
    double kD, hmkD;
};

template <typename F>
AAI_HD CellConsts<F> make_cell_consts(double side, double c, double s)
{
    CellConsts<F> z;
    const double h = 0.5 * side, k = 0.5 * (c + s), hmk = h - k;
    z.thr = (F)(2.0 * k - h);
    z.eALo = (F)(-h / c + 0.5); z.eAHi = (F)(h / c + 0.5); z.eAV = (F)((2.0 * k - h) / c + 0.5);
    z.eBLo = (F)(-h / s + 0.5); z.eBHi = (F)(h / s + 0.5); z.eBH = (F)((2.0 * k - h) / s + 0.5);
    const double hbz = h * (c + s) + 1e-5;
    z.hbz = (F)hbz;
    z.win = (int)floor(2.0 * hbz) + 1;
    // the zone is the dst square shifted by (-k, -k) in the dst frame; a dst-frame offset (a, b) is (a c + b s, -a s + b c)
    z.zx = -k * (c + s); z.zy = k * (s - c);
    z.gx = -hmk * (c + s); z.gy = hmk * (s - c);
    z.kD = k; z.hmkD = hmk;
    return z;
}

// How far from a dst pixel's centre the cell kernel fetches at most, along either lattice axis: the window starts at Z + ceil(f - hbz)
// around the ZONE's centre, which sits (zx, zy) beside the pixel's, and holds `win` positions (rotated_band_source_rows; the cells
// of a band are those of its pixels plus one more column and row: the caller spans cells [0, dW] x [row0, row1]).
template <typename F>
AAI_HD double cell_window_reach(const CellConsts<F> &z)
{
    const double off = fabs(z.zx) > fabs(z.zy) ? fabs(z.zx) : fabs(z.zy);
    const double lo = (double)z.hbz, hi = (double)z.win - (double)z.hbz;
    return off + (lo > hi ? lo : hi) + 1e-3;
}

// The quad constants as the cell formulation uses them.  hiPrec (edge parameters and vertex positions from double precision) stays
// for rotations close to an axis (QuadConsts::steep), but NOT for replicated source pixels: the quad formulation needed it there
// (8-bit noise, a 1 beside a 255: 7-8.5e-6 in plain fp32), the cell formulation does not -- every side of a cut takes its small
// part directly and a dst pixel sums L^2 pairs instead of (L + c + s)^2 -- 2.2e-6 worst over 150 random up-sampling geometries
// on 8-bit noise against 7e-7 with hiPrec, for 10 % of config 5's time (4.54 -> 4.12 ms).
template <typename F>
AAI_HD QuadConsts<F> make_cell_quad_consts(double side, double c, double s, int policy)
{
    return make_quad_consts<F>(side, c, s, policy, /*scale: no hiPrec on its account*/ 1);
}

// the cell formulation serves what the quad formulation serves (same formulas)
AAI_HD bool cell_supported(double side, double c, double s)
{
    if (!quad_supported(side, c, s)) return false;
    return (int)floor(2.0 * (0.5 * side * (c + s) + 1e-5)) + 1 <= kQuadMaxWin;
}

// rows J ... WIN - 1: (a*, b*) = this row's six crossing columns (+ 1/2); stepA / stepB = what a row adds to them
template <int WIN, int J>
struct CellRows {
    template <typename F>
    static AAI_HD void run(F aLo, F aHi, F aV, F bLo, F bHi, F bH, F stepA, F stepB, RowPlane &pLo, RowPlane &pHi, RowPlane &pV, RowPlane &pH)
    {
        // inside the zone: above BOTH lower crossings and below both upper ones
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(qmax(aLo, bLo)), pLo);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(qmin(aHi, bHi)), pHi);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(aV), pV);
        quad_row_bits<WIN, J>(quad_columns_below<WIN>(bH), pH);
        CellRows<WIN, J + 1>::run(aLo + stepA, aHi + stepA, aV + stepA, bLo + stepB, bHi + stepB, bH + stepB, stepA, stepB, pLo, pHi, pV, pH);
    }
};
template <int WIN>
struct CellRows<WIN, WIN> {
    template <typename F>
    static AAI_HD void run(F, F, F, F, F, F, F, F, RowPlane &, RowPlane &, RowPlane &, RowPlane &) {}
};

// A grid line cuts a unit pixel into a smaller and a larger part; tp = the line's distance from the pixel's nearer extreme
// corner (quad_cut_tp's mirrored parameter).  Area of the SMALLER part, exact and under the reference's corner rule for a
// left/right line; the larger part is one minus it.  Each side of the line takes its own part directly -- never
// 1 - (1 - small): a dst value far below its neighbours gets its relative accuracy from the small parts.
template <typename F>
AAI_HD void cell_cut_small(const QuadConsts<F> &q, F tp, F &exact, F &ref)
{
    const F trap = qfma(tp, q.rhi, -q.trapOff);
    const F triExact = (tp * tp) * q.r2cs;
    const F triRef = qfma(-q.hrc, tp, F(0.5)) * qfma(-q.rs, tp, F(1));
    const bool tri = tp <= q.lo;
    exact = tri ? triExact : trap;
    ref = tri ? (q.ref != 0 ? triRef : triExact) : trap;
}

// The pixel that holds G: the four rays from G (along +a = (c, -s), +b = (s, c), -a, -b in lattice axes, y down) cut it into the
// parts of the four dst pixels around G.  Area of a part = half the sum over the pixel's sides of (distance of G from the side) x
// (length of the side inside the part) -- quad_vertex_area's formula, but every side is split ONCE: side x = 1/2 is met by the
// rays +a and +b and belongs to N above the first crossing, W below the second, O in between; side y = -1/2 (rays -b, +a): NW | N |
// O; side x = -1/2 (rays -b, -a): N | NW | W; side y = 1/2 (rays -a, +b): NW | W | O.  Eight crossings, eight clamps at 0 and twelve
// multiply-adds instead of four independent evaluations (types 7-9 of Source.cpp:1276-1401 for all four dst pixels at once).
// (fx, fy) = G relative to the pixel centre, both in [-1/2, 1/2]; area[CellTarget].
template <typename F>
AAI_HD void cell_vertex_areas(F m1, F im1, F fx, F fy, F (&area)[4])
{
    auto sat = [](F x) -> F { return qmax(x, F(0)); };      // (every crossing lies on the far side of G: the lengths never exceed 1)
    const F dR = F(0.5) - fx, dL = F(0.5) + fx, dT = F(0.5) + fy, dB = F(0.5) - fy;
    // side x = 1/2, from y = -1/2 down: ray +a crosses it at fy - dR m1, ray +b at fy + dR im1
    const F rN = sat(qfma(-dR, m1, fy) + F(0.5)), rW = sat(F(0.5) - qfma(dR, im1, fy)), rO = (F(1) - rN) - rW;
    // side y = -1/2, from x = -1/2 rightwards: ray -b crosses it at fx - dT m1, ray +a at fx + dT im1
    const F tNW = sat(qfma(-dT, m1, fx) + F(0.5)), tO = sat(F(0.5) - qfma(dT, im1, fx)), tN = (F(1) - tNW) - tO;
    // side x = -1/2, from y = -1/2 down: ray -b crosses it at fy - dL im1, ray -a at fy + dL m1
    const F lN = sat(qfma(-dL, im1, fy) + F(0.5)), lW = sat(F(0.5) - qfma(dL, m1, fy)), lNW = (F(1) - lN) - lW;
    // side y = 1/2, from x = -1/2 rightwards: ray -a crosses it at fx - dB im1, ray +b at fx + dB m1
    const F bNW = sat(qfma(-dB, im1, fx) + F(0.5)), bO = sat(F(0.5) - qfma(dB, m1, fx)), bW = (F(1) - bNW) - bO;
    area[CELL_O] = F(0.5) * qfma(dR, rO, qfma(dT, tO, dB * bO));
    area[CELL_W] = F(0.5) * qfma(dR, rW, qfma(dL, lW, dB * bW));
    area[CELL_N] = F(0.5) * qfma(dR, rN, qfma(dT, tN, dL * lN));
    area[CELL_NW] = F(0.5) * qfma(dT, tNW, qfma(dL, lNW, dB * bNW));
}

// One cell.  (Zx, Zy) = the lattice point nearest the zone's centre, (dfx, dfy) = zone centre - (Zx, Zy), both in
// [-1/2, 1/2].  Source protocol as in quad_pixel (issue / commit / at).  sA[t], sVA[t] = sums of areas and of area x value
// this cell contributes to target t (CellTarget).  SCAN: src is never touched, every value counts as 1, and the return
// value says whether a decision of this cell is too close to its threshold for fp32.
// upOnly: only the parts for the dst pixels ABOVE the cell's row are wanted (the extra cell row below a strip): the interior and
// left-edge zones, which feed this row only, are skipped (the own / W sums are then incomplete and must not be used).
// NC: interleaved channels share every area: sVA[t * NC + c] = target t, channel c (NC = 1: a plain image; src.at() hands NC values)
template <typename F, int WIN, bool SCAN, bool HP, int NC = 1, typename Src>
AAI_HD bool cell_eval(const QuadConsts<F> &q, const CellConsts<F> &z, int Zx, int Zy, double dfx, double dfy, int mW, int mH, Src &src,
                      F (&sA)[4], F (&sVA)[4 * NC], bool upOnly = false)
{
    typedef typename QuadMask<WIN>::type u64;
    static_assert(WIN >= 1 && WIN <= kQuadMaxWin, "window size");
    const F fpx = (F)dfx, fpy = (F)dfy;
    // (the sums start from the parts of the pixel that holds G, below: no zeroing, no first addition)
    struct Vals { F v[NC]; };
    auto value = [&](int slot) -> Vals {
        Vals r;
        if (SCAN) {
#pragma unroll
            for (int c = 0; c < NC; ++c) r.v[c] = F(1);
        } else src.at(slot, r.v);
        return r;
    };
    // A pixel that does not reach a target contributes area 0 there -- and must then contribute nothing, whatever its value: 0 x NaN
    // would put a source pixel's NaN into a dst pixel it does not overlap.  Only the vertex zone's two-ray pixels have such targets
    // (`add`, behind a vote on the value).  Everywhere else the areas are positive: an interior pixel has area 1; a pixel cut by one
    // grid line gives both sides a positive part and the pixel that holds G four positive wedges unless a zone decision or G sits
    // within the scan's margin of its threshold -- and then the scan has left every dst pixel this cell feeds to the fix-up pass.
    auto add = [&](int target, F area, const Vals &v) {
        sA[target] += area;
#pragma unroll
        for (int c = 0; c < NC; ++c) sVA[target * NC + c] = qfma(area, area != F(0) ? v.v[c] : F(0), sVA[target * NC + c]);
    };
    auto addf = [&](int target, F area, const Vals &v) {
        sA[target] += area;
#pragma unroll
        for (int c = 0; c < NC; ++c) sVA[target * NC + c] = qfma(area, v.v[c], sVA[target * NC + c]);
    };
    auto finite = [&](const Vals &v) -> bool {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < NC; ++c) ok = ok && qabs(v.v[c]) <= F(3.0e38);
        return AAI_WAVE_ALL(ok);
    };

    // window origin: the first lattice point the zone's bounding box can hold
    const F fi0 = ceil(fpx - z.hbz), fj0 = ceil(fpy - z.hbz);
    const int i0 = (int)fi0, j0 = (int)fj0;
    const int xg0 = Zx + i0, yg0 = Zy + j0;
    // Away from the image border the whole wave's windows lie inside the lattice: one vote replaces the validity masks here and
    // the clamps in the loads
    const bool interior = AAI_WAVE_ALL(xg0 >= 0 && xg0 + WIN <= mW && yg0 >= 0 && yg0 + WIN <= mH);
    u64 valid = WIN * WIN >= 64 ? ~(u64)0 : (((u64)1 << (WIN * WIN >= 64 ? 0 : WIN * WIN)) - 1);
    if (!interior) {
        const int ia = xg0 < 0 ? -xg0 : 0, ib = (mW - 1 - xg0 < WIN - 1) ? mW - 1 - xg0 : WIN - 1;
        const int ja = yg0 < 0 ? -yg0 : 0, jb = (mH - 1 - yg0 < WIN - 1) ? mH - 1 - yg0 : WIN - 1;
        // (a window that misses the image altogether has no valid position: every sum below comes out zero -- no early exit, the other
        // lanes of the wave walk on anyway)
        const unsigned cols = ia > ib ? 0u : (2u << ib) - (1u << ia);
        valid = 0;
#pragma unroll
        for (int j = 0; j < WIN; ++j)
            if (j >= ja && j <= jb) valid |= (u64)cols << (j * WIN);
    }
    if (!SCAN) src.issue(xg0, yg0, valid, interior);

    // zone coordinates (dst frame, relative to the zone's centre) of lattice point (Zx, Zy)
    const F ac = qfma(fpy, q.s, -(fpx * q.c)), bc = -qfma(fpx, q.s, fpy * q.c);
    // hiPrec: the same relative to G, in double precision (a' = az + (h - k))
    const double a1cD = qfma(dfy, q.sD, -(dfx * q.cD)) + z.hmkD, b1cD = -qfma(dfx, q.sD, dfy * q.cD) + z.hmkD;
    // a grid line's mirrored parameter for lattice point (fi, fj): t = k + m on the E / S side, k - m on the W / N side with
    // m the centre's signed distance from the line through G; flip: the E / S side holds more than half of the pixel
    auto precise_tp = [&](F fi, F fj, bool lr, bool &flip) -> F {
        const double m = lr ? qfma((double)fi, q.cD, qfma(-(double)fj, q.sD, a1cD)) : qfma((double)fi, q.sD, qfma((double)fj, q.cD, b1cD));
        const F t1 = (F)(z.kD + m), t2 = (F)(z.kD - m);
        flip = t1 > t2;
        return qmax(qmin(t1, t2), F(0));
    };
    auto plain_tp = [&](F m1, bool &flip) -> F {        // m1 = the centre's distance from the line, fp32
        const F t = qmin(qmax(m1 + q.k, F(0)), q.k2);
        flip = t > q.k;
        return qmin(t, q.k2 - t);
    };
    bool uncertain = false;

    // ---- pass 1: which zone does every lattice point of the window belong to ------------------------------------------
    // Zone coordinates of window position (i, j): az = ac + (fi0 + i) c - (fj0 + j) s, bz = bc + (fi0 + i) s + (fj0 + j) c -- both grow
    // with i, so along a window row each threshold X is crossed at ONE column:
    //   az < X  <=>  i < X / c + iA + j s / c,   iA = -ac / c + fj0 s / c - fi0;     bz < X  <=>  i < X / s + iB - j c / s.
    // Per row: six running crossing columns, the zone's two bounds (max / min), four column counts and four bit runs (CellRows) -- 20
    // instructions where testing the row's positions one by one took ~12 each.  The crossing columns carry fp32 rounding of their own:
    // the scan (below) also classifies position by position and reports every cell where the two disagree.
    u64 mVtx, mLeft, mTop, mIn;
    {
        const F iA = qfma(-ac, q.rc, qfma(fj0, q.m1, -fi0)), iB = qfma(-bc, q.rs, qfma(-fj0, q.im1, -fi0));
        RowPlane pLo, pHi, pV, pH;
        CellRows<WIN, 0>::run(iA + z.eALo, iA + z.eAHi, iA + z.eAV, iB + z.eBLo, iB + z.eBHi, iB + z.eBH, q.m1, -q.im1, pLo, pHi, pV, pH);
        auto whole = [](const RowPlane &p) -> u64 { return WIN * WIN <= 32 ? (u64)p.lo : (u64)(((unsigned long long)p.hi << 32) | p.lo); };
        const u64 zone = whole(pHi) & ~whole(pLo) & valid, wV = whole(pV), wH = whole(pH);
        const u64 zv = zone & wV, zn = zone ^ zv;
        mVtx = zv & wH; mLeft = zv ^ mVtx; mTop = zn & wH; mIn = zn ^ mTop;
    }
    if (SCAN) {
        QuadPlane<WIN> plZ, plV, plH;         // in this cell's zone; within k of the vertical / horizontal grid line through G
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
                F az, bz;
                qfma2(fi, fi, q.c, q.s, rowA, rowB, az, bz);
                const u64 bit = (u64)1 << (j * WIN + i);
                plZ.push(j * WIN + i, qabs(az) < q.h && qabs(bz) < q.h);
                plV.push(j * WIN + i, az < z.thr);
                plH.push(j * WIN + i, bz < z.thr);
                const bool live = (valid & bit) != 0;
                const F na = qabs(qabs(az) - q.h), nb = qabs(qabs(bz) - q.h);
                const bool nearZone = qabs(az) < q.h + q.margin && qabs(bz) < q.h + q.margin;
                if (live && nearZone && (na < q.margin || nb < q.margin || qabs(az - z.thr) < q.margin || qabs(bz - z.thr) < q.margin)) uncertain = true;
            }
        }
        const u64 pZ = plZ.mask() & valid, pV = plV.mask(), pH = plH.mask();
        if ((pZ & pV & pH) != mVtx || (pZ & pV & ~pH) != mLeft || (pZ & ~pV & pH) != mTop || (pZ & ~pV & ~pH) != mIn) uncertain = true;
    }
    if (!SCAN) src.commit();

    // ---- the pixel that holds G: four wedge areas, one per dst pixel around G ----------------------------------------------
    const F gzx = HP ? (F)(dfx + z.gx) : fpx + (F)z.gx, gzy = HP ? (F)(dfy + z.gy) : fpy + (F)z.gy;     // G relative to (Zx, Zy)
    {
        const F rx = floor(gzx + F(0.5)), ry = floor(gzy + F(0.5));
        const F fx = gzx - rx, fy = gzy - ry;
        const int i = (int)rx - i0, j = (int)ry - j0;
        // (the window always holds G's pixel: its centre lies within hbz - 1e-5 of the zone's; the scan checks it all the same)
        const bool held = !SCAN || (i >= 0 && i < WIN && j >= 0 && j < WIN);
        if (!held) uncertain = true;
        const int slot = held ? j * WIN + i : 0;
        const u64 bit = (u64)1 << slot;
        if (SCAN && (qabs(fx) > F(0.5) - q.margin || qabs(fy) > F(0.5) - q.margin)) uncertain = true;
        // G's pixel lies in the vertex zone (its centre is within 1/2 of G along both lattice axes, i.e. within k along the dst axes),
        // unless G sits on its rim -- which the scan has just reported
        mVtx &= ~bit;
        if (SCAN) { mLeft &= ~bit; mTop &= ~bit; mIn &= ~bit; }
        F area[4];
        if (HP) {
            const double fxD = (dfx + z.gx) - (double)rx, fyD = (dfy + z.gy) - (double)ry;
            if (q.steep) {
                double areaD[4];
                cell_vertex_areas<double>(q.m1D, q.im1D, fxD, fyD, areaD);
#pragma unroll
                for (int t = 0; t < 4; ++t) area[t] = (F)areaD[t];
            } else cell_vertex_areas<F>(q.m1, q.im1, (F)fxD, (F)fyD, area);
        } else cell_vertex_areas<F>(q.m1, q.im1, fx, fy, area);
        // G is vertex 0 (left/top) of the cell's own dst pixel, vertex 1 (right/top) of its left neighbour,
        // vertex 2 (left/bottom) of the one above, vertex 3 of the one above left
        const bool inImage = held && (interior || (valid & bit) != 0);
        const Vals v = value(slot);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sA[t] = inImage ? area[t] : F(0);
#pragma unroll
            for (int c = 0; c < NC; ++c) sVA[t * NC + c] = inImage ? area[t] * v.v[c] : F(0);
        }
    }

    if (upOnly) { mIn = 0; mLeft = 0; }
    // coordinates of window position (i, j) relative to G: those of position (0, 0) plus i (c, s) + j (-s, c)
    const F a00 = qfma(fi0, q.c, qfma(-fj0, q.s, ac)) + q.hmk, b00 = qfma(fi0, q.s, qfma(fj0, q.c, bc)) + q.hmk;
    // ---- interior: area 1 to the cell's own dst pixel --------------------------------------------------------------------
    while (mIn) {
        const int slot = quad_ctz(mIn);
        mIn &= mIn - 1;
        addf(CELL_O, F(1), value(slot));
    }
    // ---- left-edge zone: the vertical grid line alone (the reference's corner rule applies on both sides) -----------------
    while (mLeft) {
        const int slot = quad_ctz(mLeft);
        mLeft &= mLeft - 1;
        const int j = slot / WIN, i = slot - j * WIN;
        const F a1 = qfma((F)i, q.c, qfma(-(F)j, q.s, a00));
        bool flip;
        F tp = plain_tp(a1, flip);
        if (HP) tp = precise_tp(fi0 + (F)i, fj0 + (F)j, true, flip);
        if (SCAN && q.ref != 0 && qabs(tp - q.lo) < (HP ? q.marginT : q.margin)) uncertain = true;
        F sE, sR;
        cell_cut_small(q, tp, sE, sR);
        const Vals v = value(slot);
        const F big = F(1) - sR;
        addf(CELL_O, flip ? big : sR, v);                  // flip: the E side holds the larger part
        addf(CELL_W, flip ? sR : big, v);
    }
    // ---- top-edge zone: the horizontal grid line alone (exact under both policies) ------------------------------------------
    while (mTop) {
        const int slot = quad_ctz(mTop);
        mTop &= mTop - 1;
        const int j = slot / WIN, i = slot - j * WIN;
        const F b1 = qfma((F)i, q.s, qfma((F)j, q.c, b00));
        bool flip;
        F tp = plain_tp(b1, flip);
        if (HP) tp = precise_tp(fi0 + (F)i, fj0 + (F)j, false, flip);
        F sS, unused;
        cell_cut_small(q, tp, sS, unused);
        const Vals v = value(slot);
        const F big = F(1) - sS;
        addf(CELL_O, flip ? big : sS, v);
        addf(CELL_N, flip ? sS : big, v);
    }
    // ---- vertex zone, G outside the pixel: one vertical and one horizontal ray from G cross it --------------------------------
    while (mVtx) {
        const int slot = quad_ctz(mVtx);
        mVtx &= mVtx - 1;
        const int j = slot / WIN, i = slot - j * WIN;
        const F fj = fj0 + (F)j, fi = fi0 + (F)i;
        const F a1 = qfma((F)i, q.c, qfma(-(F)j, q.s, a00)), b1 = qfma((F)i, q.s, qfma((F)j, q.c, b00));
        bool flipA, flipB;
        F tpA = plain_tp(a1, flipA), tpB = plain_tp(b1, flipB);
        if (HP) { tpA = precise_tp(fi, fj, true, flipA); tpB = precise_tp(fi, fj, false, flipB); }
        // G relative to the pixel centre along the pixel's own axes; the vertical grid line runs along (s, c), the
        // horizontal one along (c, -s): the chord of either line lies on the ray that leads back towards the pixel
        const F gx = gzx - fi, gy = gzy - fj;
        const bool xout = qabs(gx) > F(0.5);
        const bool down = xout ? gx < F(0) : gy < F(0);
        const bool right = xout ? gx < F(0) : gy > F(0);
        if (SCAN) {
            if (qmin(qabs(qabs(gx) - F(0.5)), qabs(qabs(gy) - F(0.5))) < q.margin) uncertain = true;
            if (q.ref != 0 && qabs(tpA - q.lo) < (HP ? q.marginT : q.margin)) uncertain = true;
        }
        F sA_, sAr, sB_, unused;
        cell_cut_small(q, tpA, sA_, sAr);
        cell_cut_small(q, tpB, sB_, unused);
        // E / W of the vertical line: the side `flipA` names holds the larger part; `right`: the dst pixel BETWEEN the two rays lies E of
        // it (exact part there: both edges cross the pixel), the one across the ray W of it (that cut ALONE: the reference's corner rule)
        const F bigA = F(1) - sA_, bigAr = F(1) - sAr, bigB = F(1) - sB_;
        const bool nearA = right == flipA, nearB = down == flipB;
        const F inA = nearA ? bigA : sA_, loneA = nearA ? sAr : bigAr;
        const F inB = nearB ? bigB : sB_, loneB = nearB ? sB_ : bigB;
        const F both = qmax((inA + inB) - F(1), F(0));
        // `both` goes to the dst pixel on the (right, down) side of G, loneA across the vertical ray, loneB across the horizontal one
        const F p0 = right ? both : loneA, p1 = right ? loneA : both, q0 = right ? loneB : F(0), q1 = right ? F(0) : loneB;
        const F aO = down ? p0 : q0, aW = down ? p1 : q1, aN = down ? q0 : p0, aNW = down ? q1 : p1;
        const Vals v = value(slot);
        if (finite(v)) {
            addf(CELL_O, aO, v); addf(CELL_W, aW, v); addf(CELL_N, aN, v); addf(CELL_NW, aNW, v);
        } else {
            add(CELL_O, aO, v); add(CELL_W, aW, v); add(CELL_N, aN, v); add(CELL_NW, aNW, v);
        }
    }
    return uncertain;
}

// The zone centre of cell (dx, dy) on the virtual lattice: nearest lattice point and fraction.  Split into the part that
// depends on the column only (once per lane and strip) and the step down the rows (two fused multiply-adds): the kernel, its
// scan and the CPU replay all come through here, so they agree to the last bit.  (Source.cpp:212-219 up to rounding ~1e-12.)
struct CellColumn { double bx, by; };
// The zone centre of cell (dx, dy) is affine in (dx, dy) (RotLaunch::cXa ... cY0, the coefficients of quad_centre, plus the zone's
// offset): the column's part once per wave, two fused multiply-adds per row
template <typename F>
AAI_HD CellColumn cell_column(const RotLaunch &r, const CellConsts<F> &z, int dx)
{
    CellColumn col;
    col.bx = qfma((double)dx, r.cXa, r.cX0 + z.zx);
    col.by = qfma((double)dx, r.cYa, r.cY0 + z.zy);
    return col;
}
// false: so far from the lattice that the cell touches nothing (and the integers below would leave int range)
// NEAR: the caller knows the cell to be near the lattice (the kernel: cell rows inside cell_live_rows of a strip are within 63 cell
// columns of a cell in reach) -- the test is skipped, a cell beyond reach finds its whole window outside the lattice in cell_eval
template <bool NEAR = false>
AAI_HD bool cell_anchor(const RotLaunch &r, const CellColumn &col, int dy, int &Zx, int &Zy, double &dfx, double &dfy)
{
    const double v = (double)dy;
    const double zx = qfma(v, r.cXb, col.bx), zy = qfma(v, r.cYb, col.by);
    const double cx = floor(zx + 0.5), cy = floor(zy + 0.5);
    // (one test, not four nested ones: short-circuit branches cost more than the compares)
    const bool inReach = NEAR || ((cx > -16.0) & (cx < (double)r.mW + 16.0) & (cy > -16.0) & (cy < (double)r.mH + 16.0));
    if (!inReach) return false;
    Zx = (int)cx; Zy = (int)cy; dfx = zx - cx; dfy = zy - cy;
    return true;
}

// Rows of cells that can touch the lattice at all, for the cell columns [xa, xb]: the zone centre is affine in (dx, dy) with
// positive coefficients L cos, L sin (reduced angle), so each of the four lattice bounds limits dy from one side at one end of
// the column range.  Conservative (a superset, with a margin of one row): rows outside [lo, hi] contribute nothing to any dst
// pixel and the kernel does not even compute their anchors -- the corners of a rotated canvas are 36 % of config 3's cells and
// 50 % of config 5's.  lo > hi: none.  The four bounds are lines in the column index whose coefficients the host composes once
// (make_cell_live: the divisions); a wave evaluates four multiply-adds.
struct CellLive {
    double loX0, loXk, hiX0, hiXk;     // from the lattice's X extent: dy >= loX0 + xb loXk, dy <= hiX0 + xa hiXk (useX)
    double loY0, loYk, hiY0, hiYk;     // from its Y extent: dy >= loY0 + xa loYk, dy <= hiY0 + xb hiYk (useY)
    int useX, useY;
};
template <typename F>
AAI_HD CellLive make_cell_live(const RotLaunch &r, const CellConsts<F> &z)
{
    CellLive cl;
    const double reach = (double)z.hbz + 17.0;                     // the window's half extent, the anchor's slack of 16, rounding
    const double Lc = r.side * r.cs, Ls = r.side * r.sn;
    // zone centre of cell (dx, dy): (A0 + dx Lc + dy Ls, B0 - dx Ls + dy Lc)
    const double u0 = r.fracX * r.side - r.isoX + r.offX, v0 = r.fracY * r.side - r.isoY + r.offY;
    const double A0 = u0 * r.cs + v0 * r.sn + r.isoX + z.zx, B0 = -u0 * r.sn + v0 * r.cs + r.isoY + z.zy;
    cl.useX = Ls > 0.0 ? 1 : 0; cl.useY = Lc > 0.0 ? 1 : 0;
    cl.loX0 = cl.useX ? (-reach - A0) / Ls : 0.0; cl.loXk = cl.useX ? -Lc / Ls : 0.0;
    cl.hiX0 = cl.useX ? ((double)r.mW - 1.0 + reach - A0) / Ls : 0.0; cl.hiXk = cl.loXk;
    cl.loY0 = cl.useY ? (-reach - B0) / Lc : 0.0; cl.loYk = cl.useY ? Ls / Lc : 0.0;
    cl.hiY0 = cl.useY ? ((double)r.mH - 1.0 + reach - B0) / Lc : 0.0; cl.hiYk = cl.loYk;
    return cl;
}
AAI_HD void cell_live_rows(const CellLive &cl, int xa, int xb, int &lo, int &hi)
{
    double a = -1e300, b = 1e300;
    if (cl.useX) { a = fmax(a, qfma((double)xb, cl.loXk, cl.loX0)); b = fmin(b, qfma((double)xa, cl.hiXk, cl.hiX0)); }
    if (cl.useY) { a = fmax(a, qfma((double)xa, cl.loYk, cl.loY0)); b = fmin(b, qfma((double)xb, cl.hiYk, cl.hiY0)); }
    a = floor(a) - 2.0; b = ceil(b) + 2.0;                         // (one row of margin, one for the composed coefficients' rounding)
    lo = a < -2147483000.0 ? -2147483000 : (a > 2147483000.0 ? 2147483000 : (int)a);
    hi = b < -2147483000.0 ? -2147483000 : (b > 2147483000.0 ? 2147483000 : (int)b);
}

// dst pixel (x, y) from the parts of its four cells, in the order the kernel adds them
template <typename F>
AAI_HD void cell_combine(F own, F w, F n, F nw, F &total) { total = (own + w) + (n + nw); }

}  // namespace aai
