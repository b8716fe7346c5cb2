// aai_rotated_cell.hip -- K2, the area average at a general rotation, in its "cell" formulation for gfx950 (the arithmetic
// lives in aai_rot_cell.hpp, shared with the CPU replay of the test-suite).
//
// Replaces Source.cpp:413-579 + 986-1431.  One lane per CELL of the dst grid (a dst pixel plus its top-left grid vertex):
// the lane evaluates the virtual source pixels whose centres lie in the cell's zone -- L^2 of them, every source pixel of
// the image exactly once -- and splits each one's area between the (up to four) dst pixels around the vertex.  A wave owns
// a tile of TW x (64 / TW) cells per iteration and walks DOWN the rows of its strip:
//   dst (x, y) = own(x, y) + W(x + 1, y) + N(x, y + 1) + NW(x + 1, y + 1)
// so the W / NW parts come from the next lane (one DPP shift each), the N / NW parts from the tile's next row (lane + TW) or,
// for the tile's last row, from the next iteration (the own + W sum waits in two registers); the last lane of every tile row
// only feeds its left neighbour, i.e. a wave yields TW - 1 columns x `rows` rows from TW x (rows + 1) cell evaluations.
// No barrier, no atomics; the window of pixel values is staged per lane exactly as in the quad kernel (aai_quad_src.hpp).
// Rows whose cells all miss the image (the corners of a rotated canvas) cost one wave-uniform test.
// Tile shape: the walk is written for TW x (64 / TW) tiles; 64 x 1 ships.  With 64 x 1 a wave's footprint at config 3 is a
// slanted line 190 source columns long and every 128-byte line of it is touched again in each of the next ~5 iterations, by
// which time the ~900 waves of an XCD have pushed it out of the 4 MiB L2: 8.4 M 128-byte requests on the memory side = 4.0 x
// the source.  16 x 4 tiles cut that to 3.0 M requests (1.46 x, below the quad kernel's 1.8 x) -- and run no faster (222 vs 216
// us; 5.6 vs 4.4 ms at config 5): the kernel is bound by vector-instruction issue (1.3e8 instructions x ~4 cycles / 1024
// SIMDs = its duration), the Infinity Cache absorbs the re-reads, and 16-wide rows lose one column in 16 to the halo instead
// of one in 64.  AAI_CELL_TW=16 at build time (-DAAI_CELL_TILE16) brings the variant back; profiles/r03_cell_kernel.txt.
//
// Decisions are left to double precision as in the quad kernel: aai_cell_scan_kernel runs the same code without pixel loads
// once per geometry and flags every dst pixel fed by a cell with a decision too close to its threshold (or with too little
// total area for fp32 weights); the production kernel skips flagged pixels and the fix-up pass (aai_rotated_kernel<STRICT>)
// computes them beside it.
#include "aai_quad_src.hpp"
#include "aai_rot_cell.hpp"

#include <cstdlib>
#include <cstring>

namespace aai {

namespace {

// lane i <- lane i + 1 within a tile row of TW lanes (the row's last lane gets 0: it never stores): one DPP move
// (wave_shl:1 across the wave, a gfx9 control; row_shl:1 within 16 lanes) instead of a round trip through the LDS crossbar
template <int TW>
__device__ __forceinline__ int from_next_lane_bits(int v)
{
    if (TW == 64) return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);      // wave_shl:1
    return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, false);                     // row_shl:1 (rows of 16 lanes)
}
template <int TW> __device__ __forceinline__ float from_next_lane(float v) { return __int_as_float(from_next_lane_bits<TW>(__float_as_int(v))); }
template <int TW> __device__ __forceinline__ int from_next_lane(int v) { return from_next_lane_bits<TW>(v); }

// waves per SIMD the staged windows leave room for: WIN * WIN KiB of LDS per 256-lane block, 160 KiB per CU
constexpr int cell_waves_per_simd(int win) { return 160 / (win * win) >= 8 ? 8 : 160 / (win * win); }
// ... and the register budget that goes with it: 80 registers (6 waves) up to 4 x 4 windows; from 5 x 5 (25 staged values per
// lane) 128 registers (4 waves): 96 still spill there.  (8 waves / 64 registers for the small windows: 12-44 bytes of scratch per
// lane and 14 % SLOWER at config 3, 11 % at config 5.)
constexpr int cell_min_waves(int win) { return win <= 4 ? 6 : (cell_waves_per_simd(win) > 4 ? 4 : cell_waves_per_simd(win)); }

// flag word / bit of dst pixel (dx, dy) in the 16 x 16 tiling of the plan's scans (one 64-bit word per 16 x 4 pixels)
__device__ __forceinline__ size_t flag_word(int dx, int dy, int tilesX) { return ((size_t)(dy >> 4) * tilesX + (dx >> 4)) * 4 + ((dy & 15) >> 2); }
__device__ __forceinline__ int flag_bit(int dx, int dy) { return ((dy & 3) << 4) | (dx & 15); }

// The walk both kernels share.  Per iteration every lane evaluates one cell (eval(cx, cy, sA, sVA) -> this cell is
// uncertain, SCAN only) and finishes at most one dst pixel: emit(px, py, A, VA, uncertain).  rowsPerStrip is a multiple of
// TR = 64 / TW.
// Cell rows outside [liveLo, liveHi] cannot touch the image (cell_live_rows): their iterations only finish the pixels waiting above.
// look(px, py) runs at the top of the iteration that will finish pixel (px, py) and its result is handed to emit: whatever emit
// needs from memory (the flag word of the pixel) is requested before the cell is evaluated, not waited for after it.
template <int TW, typename Look, typename Eval, typename Emit>
__device__ __forceinline__ void cell_walk(int dW, int x0, int y0, int y1, int liveLo, int liveHi, int lane, Look look, Eval eval, Emit emit)
{
    constexpr int TR = 64 / TW;
    const int lx = lane & (TW - 1), ly = lane / TW;
    const int cx = x0 + lx;
    const bool column = lx < TW - 1 && cx < dW;                 // this lane's column is one the wave completes
    float carryA = 0.f, carryVA = 0.f;                           // own + W of the tile's last row, waiting for the next iteration
    int carryU = 0;
    for (int yb = y0; yb <= y1; yb += TR) {
        const int cy = yb + ly;
        // the pixel this lane finishes in this iteration (if any): the carried row for the tile's last row of lanes
        const int py = ly == TR - 1 ? yb - 1 : cy;
        const bool finishes = column && (ly == TR - 1 ? yb > y0 : cy < y1);
        const auto seen = look(cx, finishes ? py : y0, finishes);
        if (yb > liveHi || yb + TR - 1 < liveLo) {               // wave-uniform: every cell of this iteration misses the image
            if (ly == TR - 1) {
                if (finishes) emit(cx, py, carryA, carryVA, carryU, seen);
                carryA = 0.f; carryVA = 0.f; carryU = 0;
            } else if (finishes) emit(cx, py, 0.f, 0.f, 0, seen);
            continue;
        }
        float sA[4] = {0.f, 0.f, 0.f, 0.f}, sVA[4] = {0.f, 0.f, 0.f, 0.f};
        int unc = 0;
        if (cx <= dW && cy <= y1) unc = eval(cx, cy, sA, sVA) ? 1 : 0;
        const float ownA = sA[CELL_O] + from_next_lane<TW>(sA[CELL_W]), ownVA = sVA[CELL_O] + from_next_lane<TW>(sVA[CELL_W]);
        const float belowA = sA[CELL_N] + from_next_lane<TW>(sA[CELL_NW]), belowVA = sVA[CELL_N] + from_next_lane<TW>(sVA[CELL_NW]);
        const int rowU = unc | from_next_lane<TW>(unc);          // the two cells of this tile row that feed column cx
        if (TR == 1) {
            // the row above is finished by this iteration's N / NW parts
            if (finishes) emit(cx, py, carryA + belowA, carryVA + belowVA, carryU | rowU, seen);
            carryA = ownA; carryVA = ownVA; carryU = rowU;
        } else {
            // the tile's last row of the PREVIOUS iteration (held by the lanes of row TR - 1) is finished by this iteration's
            // first row; every other row by the row below it in this tile
            const int up = (lane + 64 - (TR - 1) * TW) & 63, down = (lane + TW) & 63;
            const float topA = __shfl(belowA, up), topVA = __shfl(belowVA, up);
            const int topU = __shfl(rowU, up);
            const float nextA = __shfl(belowA, down), nextVA = __shfl(belowVA, down);
            const int nextU = __shfl(rowU, down);
            if (ly == TR - 1) {
                if (finishes) emit(cx, py, carryA + topA, carryVA + topVA, carryU | topU, seen);
                carryA = ownA; carryVA = ownVA; carryU = rowU;
            } else if (finishes) emit(cx, py, ownA + nextA, ownVA + nextVA, rowU | nextU, seen);
        }
    }
}

template <typename T, int WIN, bool SCALED, bool HP, int TW>
__global__ __launch_bounds__(kQuadBlock, cell_min_waves(WIN)) void aai_cell_kernel(
    RotLaunch r, QuadConsts<float> q, CellConsts<float> z, QuadMap m, const T *__restrict__ src, ImageView sv, float *__restrict__ dst, ImageView dv,
    const unsigned long long *__restrict__ skipMasks, int tilesX, int rowsPerStrip, int bigStrips, int tailRows)
{
    __shared__ float window[WIN * WIN][kQuadBlock];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = (blockIdx.x * (kQuadBlock / 64) + wave) * (TW - 1);
    if (x0 >= r.dW) return;                                   // wave-uniform; no barrier below
    // (row bands in launch order: dealing them from the middle outwards, so that the launch's tail is made of the cheap
    // corner bands, measured 5 % SLOWER at config 3 -- profiles/r03_cell_kernel.txt)
    // the last strips of a launch are shorter (tailRows rows instead of rowsPerStrip): the waves that finish it live a fraction
    // as long, and the chip drains in a fraction of the time
    const int by = blockIdx.y;
    const int y0 = r.dyBase + (by < bigStrips ? by * rowsPerStrip : bigStrips * rowsPerStrip + (by - bigStrips) * tailRows);
    const int y1 = min(y0 + (by < bigStrips ? rowsPerStrip : tailRows), r.dyEnd);           // dst rows [y0, y1); cells rows y0 .. y1
    float *image = dst + (int64_t)blockIdx.z * dv.imageStride;
    const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
    const CellColumn col = cell_column(r, z, x0 + (lane & (TW - 1)));
    int liveLo, liveHi;
    cell_live_rows(r, z, x0, x0 + TW - 1, liveLo, liveHi);     // wave-uniform
    // per-pixel masks only where a 16 x 16 tile this strip touches holds a flagged pixel (QuadMap::tileFlags; wave-uniform, scalar loads)
    const unsigned long long *masks = skipMasks;
    if (masks && m.tileFlags && TW == 64) {
        bool any = false;
        for (int tr = y0 >> 4; tr <= (y1 - 1) >> 4; ++tr) any = any || tiles_flagged(m.tileFlags, m.tileFlagWords, tr, x0 >> 4, 5);
        if (!any) masks = nullptr;
    }
    cell_walk<TW>(r.dW, x0, y0, y1, liveLo, liveHi, lane,
        [&](int px, int py, bool wanted) -> bool {
            // is the pixel one the plan's scans left to the fix-up pass?  (requested here, used after the cell is evaluated)
            return masks && wanted && ((masks[flag_word(px, py, tilesX)] >> flag_bit(px, py)) & 1ull);
        },
        [&](int cx, int cy, float (&sA)[4], float (&sVA)[4]) -> bool {
            int Zx, Zy;
            double dfx, dfy;
            if (!cell_anchor(r, col, cy, Zx, Zy, dfx, dfy)) return false;
            QuadSrc<T, WIN, SCALED, true> s;
            s.img = img; s.m = &m; s.mW = r.mW; s.mH = r.mH; s.lds = window; s.tid = tid;
            // (the extra cell row below the strip only feeds the strip's last pixel row: its interior / left-edge zones are skipped)
            cell_eval<float, WIN, false, HP>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA, TW == 64 && cy == y1);
            return false;
        },
        [&](int px, int py, float A, float VA, int, bool skip) {
            if (skip) return;                                  // a pixel the plan's scans left to the fix-up pass is not written here
            // (written once, never read back: around the caches -- 1 % at configs 3 and 5)
            __builtin_nontemporal_store(A > 0.f ? VA / A : 0.f, image + ((int64_t)(py - r.dyBase) * dv.rowStride + px));         // Source.cpp:577
        });
}

// Once per geometry: the same walk without pixel loads.  Sets the bit of every dst pixel one of whose four cells has a
// decision too close to its threshold, or whose total area is too small for fp32 weights (QuadConsts::minArea), in the
// lane masks of the 16 x 16 tiling (on top of the knife-edge scan's bits) and counts the newly set bits in counter[0].
template <int WIN, bool HP>
__global__ __launch_bounds__(kQuadBlock) void aai_cell_scan_kernel(RotLaunch r, QuadConsts<float> q, CellConsts<float> z, unsigned long long *__restrict__ laneMasks,
                                                                  unsigned *__restrict__ counter, int tilesX, int rowsPerStrip, int band0)
{
    constexpr int TW = 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = (blockIdx.x * (kQuadBlock / 64) + wave) * (TW - 1);
    if (x0 >= r.dW) return;
    const int y0 = (band0 + blockIdx.y) * rowsPerStrip;
    const int y1 = min(y0 + rowsPerStrip, r.dH);
    const CellColumn col = cell_column(r, z, x0 + lane);
    int liveLo, liveHi;
    cell_live_rows(r, z, x0, x0 + TW - 1, liveLo, liveHi);
    cell_walk<TW>(r.dW, x0, y0, y1, liveLo, liveHi, lane,
        [&](int, int, bool) -> int { return 0; },
        [&](int cx, int cy, float (&sA)[4], float (&sVA)[4]) -> bool {
            int Zx, Zy;
            double dfx, dfy;
            if (!cell_anchor(r, col, cy, Zx, Zy, dfx, dfy)) return false;
            NoSrc s;
            return cell_eval<float, WIN, true, HP>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA);
        },
        [&](int px, int py, float A, float, int uncertain, int) {
            if (uncertain | ((A > 0.f && A < q.minArea) ? 1 : 0)) {
                const unsigned long long bit = 1ull << flag_bit(px, py);
                const unsigned long long old = atomicOr(laneMasks + flag_word(px, py, tilesX), bit);
                if (!(old & bit)) atomicAdd(counter, 1u);
            }
        });
}

// tile width per wave (see the header): 64; a build with -DAAI_CELL_TILE16 also holds the 16 x 4 variant, chosen by AAI_CELL_TW=16
static int cell_tile_width(const QuadMap &)
{
#if defined(AAI_CELL_TILE16)
    static const int forced = [] { const char *e = getenv("AAI_CELL_TW"); return e ? atoi(e) : 0; }();
    if (forced == 16) return 16;
#endif
    return 64;
}

template <typename T, int WIN, int TW>
hipError_t launch_cell_tile(const RotLaunch &r, const QuadConsts<float> &q, const CellConsts<float> &z, const QuadMap &m, const T *src, ImageView sv,
                            float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    const int rowsPerStrip = cell_rows_per_strip(r.dW, r.dyEnd - r.dyBase, batch, TW);
    const int strips = (r.dW + TW - 2) / (TW - 1);
    const int rows = r.dyEnd - r.dyBase;
    // A launch with few waves (rowsPerStrip already at its minimum of 8: fewer than four rounds of the chip's wave slots) spends a
    // fifth of its time draining: its last tenth of the rows goes in strips of 4, whose waves live half as long (config 3, one
    // image: 211 -> 201 us; config 5, 32-row strips and 190 k waves, loses 0.4 ... 7 % to any tail: tools/cell_tail_ab.sh).
    // Experiments: AAI_CELL_TAIL="<percent of the rows>,<rows per tail strip>".
    static const int forcedPct = [] { const char *e = getenv("AAI_CELL_TAIL"); return e ? atoi(e) : -1; }();
    static const int forcedR = [] { const char *e = getenv("AAI_CELL_TAIL"); const char *c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 2; }();
    const int tailPct = forcedPct >= 0 ? forcedPct : (rowsPerStrip == 8 ? 10 : 0), tailR = forcedPct >= 0 ? forcedR : 4;
    int bigStrips = (rows + rowsPerStrip - 1) / rowsPerStrip, tailRows = rowsPerStrip, tailStrips = 0;
    if (tailPct > 0 && TW == 64 && tailR > 0 && tailR < rowsPerStrip) {
        bigStrips = (int)((int64_t)rows * (100 - tailPct) / 100 / rowsPerStrip);
        tailRows = tailR;
        tailStrips = (rows - bigStrips * rowsPerStrip + tailRows - 1) / tailRows;
        if (bigStrips + tailStrips > 65535) { bigStrips = (rows + rowsPerStrip - 1) / rowsPerStrip; tailRows = rowsPerStrip; tailStrips = 0; }      // grid.y
    }
    const dim3 grid((strips + 3) / 4, bigStrips + tailStrips, batch);
    const int tilesX = (r.dW + 15) / 16;
#define AAI_CELL_LAUNCH(SCALED, HP) \
    hipLaunchKernelGGL((aai_cell_kernel<T, WIN, SCALED, HP, TW>), grid, dim3(kQuadBlock), 0, stream, r, q, z, m, src, sv, dst, dv, skipMasks, tilesX, rowsPerStrip, bigStrips, tailRows)
    if (m.scale > 1) {
        if (q.hiPrec) AAI_CELL_LAUNCH(true, true); else AAI_CELL_LAUNCH(true, false);
    } else {
        if (q.hiPrec) AAI_CELL_LAUNCH(false, true); else AAI_CELL_LAUNCH(false, false);
    }
#undef AAI_CELL_LAUNCH
    return hipGetLastError();
}

template <typename T, int WIN>
hipError_t launch_cell_win(const RotLaunch &r, const QuadConsts<float> &q, const CellConsts<float> &z, const QuadMap &m, const T *src, ImageView sv,
                           float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
#if defined(AAI_CELL_TILE16)
    if (cell_tile_width(m) == 16) return launch_cell_tile<T, WIN, 16>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
#endif
    (void)cell_tile_width;
    return launch_cell_tile<T, WIN, 64>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
}

template <typename T>
hipError_t launch_cell_typed(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch,
                             const unsigned long long *skipMasks, hipStream_t stream)
{
    const QuadConsts<float> q = make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    switch (z.win) {
    case 2: return launch_cell_win<T, 2>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 3: return launch_cell_win<T, 3>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 4: return launch_cell_win<T, 4>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 5: return launch_cell_win<T, 5>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 6: return launch_cell_win<T, 6>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 7: return launch_cell_win<T, 7>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 8: return launch_cell_win<T, 8>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

// dst rows a wave walks: a strip of R rows costs R + 1 cell rows (R + 64 / tileWidth for tiles of several rows), so taller is
// cheaper -- as long as the launch still has several waves for every SIMD of the chip (1024 SIMDs x ~6 wave slots)
int cell_rows_per_strip(int dW, int rows, int batch, int tileWidth)
{
    static const int forced = [] { const char *e = getenv("AAI_CELL_ROWS"); return e ? atoi(e) : 0; }();
    const int tr = 64 / tileWidth;
    if (forced > 0) return (forced + tr - 1) / tr * tr;
    const int64_t strips = ((int64_t)dW + tileWidth - 2) / (tileWidth - 1) * batch;
    int R = 32;
    while (R > 8 && strips * ((rows + R - 1) / R) < 24576) R >>= 1;
    while ((rows + R - 1) / R > 65535) R <<= 1;               // grid.y
    return R;
}

// (tests: aai_debug_cell_min_waves(0) sends small images to the cell kernel too)
static int g_cellMinWaves = 1024;
void set_cell_min_waves(int waves) { g_cellMinWaves = waves < 0 ? 1024 : waves; }

bool cell_can_serve(const RotLaunch &r, int srcType, ImageView sv)
{
    // plain images below 4 GiB (lanes address their pixels with unsigned 32-bit byte offsets from the image's first element)
    static const bool enabled = [] { const char *e = getenv("AAI_CELL"); return !(e && atoi(e) == 0); }();      // experiments: AAI_CELL=0 keeps the quad kernel
    if (!enabled || !r.cell || r.chan > 1 || r.mode != AAI_MODE_AREA) return false;
    // Small outputs stay on the quad kernel: a cell wave lives for rows + 1 cell rows, and an image of fewer than ~1000 such
    // waves (about 720 x 720 dst pixels) cannot fill the chip with them -- the reference's own example call (158 x 158 dst
    // pixels at 5.9 : 1) takes 71 us on 60 cell waves and 36 us on 390 one-shot quad waves.
    if ((int64_t)((r.dW + 62) / 63) * ((r.dH + 7) / 8) < g_cellMinWaves) return false;
    const int64_t esz = srcType == SRC_U8 ? 1 : srcType == SRC_U16 ? 2 : 4;
    return (int64_t)r.H * sv.rowStride * esz < ((int64_t)1 << 32);
}

hipError_t launch_cell(const RotLaunch &r, const QuadMap &map, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    if (r.dW <= 0 || r.dyEnd <= r.dyBase || batch <= 0) return hipSuccess;
    QuadMap m = map;
    m.anchorRows = 0;
    switch (srcType) {
    case SRC_U8: return launch_cell_typed(r, m, static_cast<const unsigned char *>(src), sv, dst, dv, batch, skipMasks, stream);
    case SRC_U16: return launch_cell_typed(r, m, static_cast<const unsigned short *>(src), sv, dst, dv, batch, skipMasks, stream);
    default: return launch_cell_typed(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream);
    }
}

hipError_t launch_cell_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream)
{
    if (r.dW <= 0 || r.dH <= 0) return hipSuccess;
    const QuadConsts<float> q = make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    const int rows = 16;
    const int strips = (r.dW + 62) / 63;                       // the scan walks 64 x 1 tiles
    const int tilesX = (r.dW + 15) / 16;
    const int bands = (r.dH + rows - 1) / rows;
    for (int b0 = 0; b0 < bands; b0 += 65535) {                // grid.y carries at most 65535 bands
        const dim3 grid((strips + 3) / 4, bands - b0 < 65535 ? bands - b0 : 65535, 1);
#define AAI_CELL_SCAN(W)                                                                                                                               \
    case W:                                                                                                                                            \
        if (q.hiPrec) hipLaunchKernelGGL((aai_cell_scan_kernel<W, true>), grid, dim3(kQuadBlock), 0, stream, r, q, z, laneMasks, counter, tilesX, rows, b0); \
        else hipLaunchKernelGGL((aai_cell_scan_kernel<W, false>), grid, dim3(kQuadBlock), 0, stream, r, q, z, laneMasks, counter, tilesX, rows, b0);    \
        break;
        switch (z.win) {
            AAI_CELL_SCAN(2) AAI_CELL_SCAN(3) AAI_CELL_SCAN(4) AAI_CELL_SCAN(5) AAI_CELL_SCAN(6) AAI_CELL_SCAN(7) AAI_CELL_SCAN(8)
        default: return hipErrorInvalidValue;
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
#undef AAI_CELL_SCAN
    return hipGetLastError();
}

}  // namespace aai
