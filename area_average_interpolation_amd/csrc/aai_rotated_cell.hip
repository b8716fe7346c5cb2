// aai_rotated_cell.hip -- K2, the area average at a general rotation, in its "cell" formulation for gfx950 (the arithmetic
// lives in aai_rot_cell.hpp, shared with the CPU replay of the test-suite).
//
// Replaces Source.cpp:413-579 + 986-1431.  One lane per CELL of the dst grid (a dst pixel plus its top-left grid vertex):
// the lane evaluates the virtual source pixels whose centres lie in the cell's zone -- L^2 of them, every source pixel of
// the image exactly once -- and splits each one's area between the (up to four) dst pixels around the vertex.  A wave owns
// 64 consecutive cells of a dst row and walks DOWN the rows of its strip:
//   dst (x, y) = own(x, y) + W(x + 1, y) + N(x, y + 1) + NW(x + 1, y + 1)
// so the W / NW parts come from the next lane (one cross-lane shift each) and the N / NW parts from the next iteration (the
// own + W sum waits in two registers); lanes 0..62 store, i.e. a wave yields 63 columns x `rows` rows from 64 x (rows + 1)
// cell evaluations.  No barrier, no atomics; the window of pixel values is staged per lane exactly as in the quad kernel
// (aai_quad_src.hpp).  Rows of a strip whose cells all miss the image (the corners of a rotated canvas) cost one
// wave-uniform test.
//
// Decisions are left to double precision as in the quad kernel: aai_cell_scan_kernel runs the same code without pixel loads
// once per geometry and flags every dst pixel fed by a cell with a decision too close to its threshold (or with too little
// total area for fp32 weights); the production kernel skips flagged pixels and the fix-up pass (aai_rotated_kernel<STRICT>)
// computes them beside it.
#include "aai_quad_src.hpp"
#include "aai_rot_cell.hpp"

#include <cstdlib>

namespace aai {

namespace {

constexpr int kCellCols = 63;        // dst columns a wave completes (its 64th cell only feeds column 62)

// lane i <- lane i + 1 (lane 63 gets 0: it never stores): one DPP move across the whole wave (wave_shl:1, a gfx9 control)
// instead of a round trip through the LDS crossbar (ds_bpermute) at the end of every row
__device__ __forceinline__ float from_next_lane(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ int from_next_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }

// waves per SIMD the staged windows leave room for: WIN * WIN KiB of LDS per 256-lane block, 160 KiB per CU
constexpr int cell_waves_per_simd(int win) { return 160 / (win * win) >= 8 ? 8 : 160 / (win * win); }
// ... and the register budget that goes with it: 80 registers (6 waves) up to 4 x 4 windows; from 5 x 5 (25 staged values per
// lane) 128 registers (4 waves): 96 still spill there
constexpr int cell_min_waves(int win) { return win <= 4 ? 6 : (cell_waves_per_simd(win) > 4 ? 4 : cell_waves_per_simd(win)); }

// flag word / bit of dst pixel (dx, dy) in the 16 x 16 tiling of the plan's scans (one 64-bit word per 16 x 4 pixels)
__device__ __forceinline__ size_t flag_word(int dx, int dy, int tilesX) { return ((size_t)(dy >> 4) * tilesX + (dx >> 4)) * 4 + ((dy & 15) >> 2); }
__device__ __forceinline__ int flag_bit(int dx, int dy) { return ((dy & 3) << 4) | (dx & 15); }

template <typename T, int WIN, bool SCALED, bool HP>
__global__ __launch_bounds__(kQuadBlock, cell_min_waves(WIN)) void aai_cell_kernel(
    RotLaunch r, QuadConsts<float> q, CellConsts<float> z, QuadMap m, const T *__restrict__ src, ImageView sv, float *__restrict__ dst, ImageView dv,
    const unsigned long long *__restrict__ skipMasks, int tilesX, int rowsPerStrip)
{
    __shared__ float window[WIN * WIN][kQuadBlock];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = (blockIdx.x * (kQuadBlock / 64) + wave) * kCellCols;
    if (x0 >= r.dW) return;                                   // wave-uniform; no barrier below
    const int cx = x0 + lane;
    // (row bands in launch order: dealing them from the middle outwards, so that the launch's tail is made of the cheap
    // corner bands, measured 5 % SLOWER at config 3 -- profiles/r03_cell_kernel.txt)
    const int y0 = r.dyBase + blockIdx.y * rowsPerStrip;
    const int y1 = min(y0 + rowsPerStrip, r.dyEnd);           // dst rows [y0, y1); cells rows y0 .. y1
    const bool stores = lane < kCellCols && cx < r.dW;
    float *outCol = dst + (int64_t)blockIdx.z * dv.imageStride + cx;
    const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);

    const CellColumn col = cell_column(r, z, cx);
    float pendA = 0.f, pendVA = 0.f;
    for (int cy = y0; cy <= y1; ++cy) {
        // the row above is finished in this iteration: is its pixel one the plan's scans left to the fix-up pass?
        bool skip = false;
        if (skipMasks && cy > y0 && stores) skip = (skipMasks[flag_word(cx, cy - 1, tilesX)] >> flag_bit(cx, cy - 1)) & 1ull;

        float sA[4] = {0.f, 0.f, 0.f, 0.f}, sVA[4] = {0.f, 0.f, 0.f, 0.f};
        int Zx = 0, Zy = 0;
        double dfx = 0.0, dfy = 0.0;
        const bool live = cx <= r.dW && cell_anchor(r, col, cy, Zx, Zy, dfx, dfy);
        if (live) {
            QuadSrc<T, WIN, SCALED, true> s;
            s.img = img; s.m = &m; s.mW = r.mW; s.mH = r.mH; s.lds = window; s.tid = tid;
            cell_eval<float, WIN, false, HP>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA);
        }
        const float wA = from_next_lane(sA[CELL_W]), wVA = from_next_lane(sVA[CELL_W]);
        const float nwA = from_next_lane(sA[CELL_NW]), nwVA = from_next_lane(sVA[CELL_NW]);
        if (cy > y0 && stores && !skip) {
            const float A = pendA + (sA[CELL_N] + nwA), VA = pendVA + (sVA[CELL_N] + nwVA);
            outCol[(int64_t)(cy - 1 - r.dyBase) * dv.rowStride] = A > 0.f ? VA / A : 0.f;         // Source.cpp:577
        }
        pendA = sA[CELL_O] + wA;
        pendVA = sVA[CELL_O] + wVA;
    }
}

// Once per geometry: the same walk without pixel loads.  Sets the bit of every dst pixel one of whose four cells has a
// decision too close to its threshold, or whose total area is too small for fp32 weights (QuadConsts::minArea), in the
// lane masks of the 16 x 16 tiling (on top of the knife-edge scan's bits) and counts the newly set bits in counter[0].
template <int WIN, bool HP>
__global__ __launch_bounds__(kQuadBlock) void aai_cell_scan_kernel(RotLaunch r, QuadConsts<float> q, CellConsts<float> z, unsigned long long *__restrict__ laneMasks,
                                                                  unsigned *__restrict__ counter, int tilesX, int rowsPerStrip, int band0)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = (blockIdx.x * (kQuadBlock / 64) + wave) * kCellCols;
    if (x0 >= r.dW) return;
    const int cx = x0 + lane;
    const int y0 = (band0 + blockIdx.y) * rowsPerStrip;
    const int y1 = min(y0 + rowsPerStrip, r.dH);
    const bool stores = lane < kCellCols && cx < r.dW;
    const CellColumn col = cell_column(r, z, cx);
    float pendA = 0.f;
    int pendU = 0;
    for (int cy = y0; cy <= y1; ++cy) {
        float sA[4] = {0.f, 0.f, 0.f, 0.f}, sVA[4] = {0.f, 0.f, 0.f, 0.f};
        int Zx = 0, Zy = 0;
        double dfx = 0.0, dfy = 0.0;
        int unc = 0;
        if (cx <= r.dW && cell_anchor(r, col, cy, Zx, Zy, dfx, dfy)) {
            NoSrc s;
            unc = cell_eval<float, WIN, true, HP>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA) ? 1 : 0;
        }
        const float wA = from_next_lane(sA[CELL_W]), nwA = from_next_lane(sA[CELL_NW]);
        const int uNext = from_next_lane(unc);
        if (cy > y0 && stores) {
            const float A = pendA + (sA[CELL_N] + nwA);
            if (pendU | unc | uNext | ((A > 0.f && A < q.minArea) ? 1 : 0)) {
                const unsigned long long bit = 1ull << flag_bit(cx, cy - 1);
                const unsigned long long old = atomicOr(laneMasks + flag_word(cx, cy - 1, tilesX), bit);
                if (!(old & bit)) atomicAdd(counter, 1u);
            }
        }
        pendA = sA[CELL_O] + wA;
        pendU = unc | uNext;
    }
}

template <typename T, int WIN>
hipError_t launch_cell_win(const RotLaunch &r, const QuadConsts<float> &q, const CellConsts<float> &z, const QuadMap &m, const T *src, ImageView sv,
                           float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, int rowsPerStrip, hipStream_t stream)
{
    const int strips = (r.dW + kCellCols - 1) / kCellCols;
    const dim3 grid((strips + 3) / 4, (r.dyEnd - r.dyBase + rowsPerStrip - 1) / rowsPerStrip, batch);
    const int tilesX = (r.dW + 15) / 16;
#define AAI_CELL_LAUNCH(SCALED, HP) \
    hipLaunchKernelGGL((aai_cell_kernel<T, WIN, SCALED, HP>), grid, dim3(kQuadBlock), 0, stream, r, q, z, m, src, sv, dst, dv, skipMasks, tilesX, rowsPerStrip)
    if (m.scale > 1) {
        if (q.hiPrec) AAI_CELL_LAUNCH(true, true); else AAI_CELL_LAUNCH(true, false);
    } else {
        if (q.hiPrec) AAI_CELL_LAUNCH(false, true); else AAI_CELL_LAUNCH(false, false);
    }
#undef AAI_CELL_LAUNCH
    return hipGetLastError();
}

template <typename T>
hipError_t launch_cell_typed(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch,
                             const unsigned long long *skipMasks, hipStream_t stream)
{
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    const int rows = cell_rows_per_strip(r.dW, r.dyEnd - r.dyBase, batch);
    switch (z.win) {
    case 2: return launch_cell_win<T, 2>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, rows, stream);
    case 3: return launch_cell_win<T, 3>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, rows, stream);
    case 4: return launch_cell_win<T, 4>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, rows, stream);
    case 5: return launch_cell_win<T, 5>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, rows, stream);
    case 6: return launch_cell_win<T, 6>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, rows, stream);
    case 7: return launch_cell_win<T, 7>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, rows, stream);
    case 8: return launch_cell_win<T, 8>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, rows, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

// dst rows a wave walks: a strip of R rows costs R + 1 cell rows, so taller is cheaper -- as long as the launch still has
// several waves for every SIMD of the chip (1024 SIMDs x ~6 wave slots)
int cell_rows_per_strip(int dW, int rows, int batch)
{
    static const int forced = [] { const char *e = getenv("AAI_CELL_ROWS"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    const int64_t strips = ((int64_t)dW + kCellCols - 1) / kCellCols * batch;
    int R = 32;
    while (R > 8 && strips * ((rows + R - 1) / R) < 24576) R >>= 1;
    while ((rows + R - 1) / R > 65535) R <<= 1;               // grid.y
    return R;
}

bool cell_can_serve(const RotLaunch &r, int srcType, ImageView sv)
{
    // plain images below 4 GiB (lanes address their pixels with unsigned 32-bit byte offsets from the image's first element)
    static const bool enabled = [] { const char *e = getenv("AAI_CELL"); return !(e && atoi(e) == 0); }();      // experiments: AAI_CELL=0 keeps the quad kernel
    if (!enabled || !r.cell || r.chan > 1 || r.mode != AAI_MODE_AREA) return false;
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
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    const int rows = 16;
    const int strips = (r.dW + kCellCols - 1) / kCellCols;
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
