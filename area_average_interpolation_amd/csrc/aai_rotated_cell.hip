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

// (a cap on the kernel's scalar registers -- 96 would admit a seventh wave per SIMD -- was measured and LOSES: config 3 156 -> 164 us, the
// spills cost more than the wave brings; -DAAI_CELL_SGPRS=n brings it back; profiles/r04_cell_kernel.txt)
#ifndef AAI_CELL_SGPRS
#define AAI_CELL_SGPRS 0
#endif
namespace aai {

// dst rows a wave walks.  A workgroup of 4 waves pays 4 R + 1 cell rows for 4 R dst rows, so taller is cheaper in instructions --
// but the four segments of a workgroup share a CU, and what they fetch.  Measured (8192^2 sources, one image; profiles/
// r04_cell_kernel.txt), best R of 4 / 8 / 16 / 32 by a margin of 9 ... 35 %: 3:1 and 4:1 -> 4; 2:1 and 1.5:1 -> 8; 1:1 -> 16; x2 and
// x4 up-sampling -> 16 ... 32: the workgroup's footprint, 4 R dst rows of `srcRowsPerDstRow` source rows each, wants to be about 50
// source rows (a CU's cache holds them while its waves walk).  Never so tall that the launch has fewer than several waves for every
// SIMD of the chip (1024 SIMDs x ~6 wave slots).
static int cell_rows_per_wave(int dW, int rows, int batch, double srcRowsPerDstRow, int waveRows = 1)
{
    const char *e = experiment_env("AAI_CELL_ROWS");
    if (e && atoi(e) > 0) return atoi(e);
    const int cols = cell_wave_cols(waveRows);
    const int64_t strips = ((int64_t)dW + cols - 1) / cols * batch;
    int R;
    int64_t floorWaves = 24576;
    if (waveRows == 2) {
        // the 32 x 2 wave (ratios from 2:1 up; us per image, tools/cell_wave_ab.sh): 16 rows up to config 3's ratio (one image 174 / 154 /
        // 147 at 4 / 8 / 16 rows; 8 images per launch 150 / 133 / 130; 2:1 at 30 degrees 290 / 275 / 273 at 8 / 16 / 32), 8 rows from
        // ~2.7:1 up (3:1 at 30 degrees 281 / 323 / 342 at 8 / 16 / 32; 4:1 at 45: 184 / 210 / 251; 5:1 at 17.5: 189 / 214 / 240)
        R = srcRowsPerDstRow < 2.7 ? 16 : 8;
        floorWaves = 16384;                                     // (config 3, one image, 16 rows: 24 k waves -- and the best of 4 / 8 / 16)
    } else {
        const double want = 14.0 / (srcRowsPerDstRow > 0.05 ? srcRowsPerDstRow : 0.05);
        // (the nearest power of two, except that ratios from ~1.9:1 up keep 4: since the classification by intervals the kernel waits for
        // memory sooner -- 2:1 at 45 degrees, 4 images per launch: 229 / 254 / 372 us per image at 4 / 8 / 16 rows; config 3 at 8 images per
        // launch: 163 / 194 us at 4 / 8 -- profiles/r04_cell_kernel.txt)
        R = want < 7.4 ? 4 : (want < 11.4 ? 8 : (want < 22.7 ? 16 : 32));
    }
    while (R > 4 && strips * ((rows + R - 1) / R) < floorWaves) R >>= 1;
    while ((rows + 4 * R - 1) / (4 * R) > 65535) R <<= 1;       // grid.y
    return R;
}

namespace {

// lane i <- lane i + 1 (the wave's last lane gets 0: it never stores): one DPP move (wave_shl:1, a gfx9 control) instead of a
// round trip through the LDS crossbar
__device__ __forceinline__ int from_next_lane_bits(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float from_next_lane(float v) { return __int_as_float(from_next_lane_bits(__float_as_int(v))); }
__device__ __forceinline__ int from_next_lane(int v) { return from_next_lane_bits(v); }

// waves per SIMD the staged windows leave room for: WIN * WIN KiB of LDS per 256-lane block, 160 KiB per CU
constexpr int cell_waves_per_simd(int win) { return 160 / (win * win) >= 8 ? 8 : 160 / (win * win); }
// ... and the register budget that goes with it: 80 registers (6 waves) up to 4 x 4 windows; from 5 x 5 (25 staged values per
// lane) 128 registers (4 waves): 96 still spill there.  (8 waves / 64 registers for the small windows: 12-44 bytes of scratch per
// lane and 14 % SLOWER at config 3, 11 % at config 5.)
constexpr int cell_min_waves(int win) { return win <= 4 ? 6 : (cell_waves_per_simd(win) > 4 ? 4 : cell_waves_per_simd(win)); }

// flag word / bit of dst pixel (dx, dy) in the 16 x 16 tiling of the plan's scans (one 64-bit word per 16 x 4 pixels)
__device__ __forceinline__ size_t flag_word(int dx, int dy, int tilesX) { return ((size_t)(dy >> 4) * tilesX + (dx >> 4)) * 4 + ((dy & 15) >> 2); }
__device__ __forceinline__ int flag_bit(int dx, int dy) { return ((dy & 3) << 4) | (dx & 15); }

constexpr int kCellMultiMaxKiB = 64;        // the interleaved kernel's window: WIN * WIN * WORDS KiB of LDS at most
constexpr int kCellXcdRowsDefault = 2;      // row blocks per XCD band without replication (profiles/r04_fast_xcd.txt)
constexpr int kCellWaves = kQuadBlock / 64;          // waves of a workgroup: consecutive row segments of ONE strip of 63 dst columns

// What the waves of a workgroup hand to the wave above them: the N / NW parts of their first cell row (which finish the last
// dst row of the segment above), [wave][A, VA, uncertain][lane]
template <int NC>
struct CellHandoffN { float a[kCellWaves][64], va[kCellWaves][NC][64]; int u[kCellWaves][64]; };
typedef CellHandoffN<1> CellHandoff;

// The walk both kernels share.  A workgroup owns 63 dst columns x (kCellWaves x rowsPerWave) dst rows; wave w walks DOWN the
// dst rows [y0, y1) of its segment: per iteration every lane evaluates one cell (eval(cx, cy, sA, sVA, upOnly) -> this cell is
// uncertain, SCAN only) and finishes one dst pixel of the row above: emit(px, py, A, VA, uncertain, seen).  A segment of R dst rows
// needs R + 1 cell rows; the extra one IS the first cell row of the segment below, so every wave parks the N / NW parts of its
// first row in LDS (one barrier per workgroup, right after the first row) and only the workgroup's last segment (ownsBottom)
// evaluates its bottom cell row itself -- 4 R + 1 cell rows per 4 R dst rows where one wave per strip paid R + 1 per R, and
// waves that live a quarter as long at the same overhead (the drain of a launch is one wave's life).
// Cell rows outside [liveLo, liveHi] cannot touch the image (cell_live_rows): they only finish the pixels waiting above.
// look(px, py) runs at the top of the iteration that will finish pixel (px, py) and its result is handed to emit: whatever emit
// needs from memory (the flag word of the pixel) is requested before the cell is evaluated, not waited for after it.
// Every wave of the workgroup must call this (the barrier), also with an empty segment (y0 >= y1).
// NC: interleaved channels (every sum of area x value once per channel; sVA[t * NC + c], emit's VA[c])
template <int NC, typename Look, typename Eval, typename Emit>
__device__ __forceinline__ void cell_walk(int dW, int x0, int y0, int y1, bool ownsBottom, int wave, CellHandoffN<NC> &hand, int liveLo, int liveHi, int lane,
                                          Look look, Eval eval, Emit emit)
{
    const int cx = x0 + lane;
    const bool column = lane < 63 && cx < dW;                   // this lane's column is one the wave completes
    float ownA = 0.f, ownVA[NC], belowA = 0.f, belowVA[NC];
    int rowU = 0;
    // one cell row: own + W of this row (own*), N + NW for the row above (below*)
    bool rowLive = false, aboveLive = false;                     // wave-uniform: this cell row / the one above can touch the image
    auto row = [&](int cy, bool upOnly) {
        ownA = 0.f; belowA = 0.f; rowU = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { ownVA[c] = 0.f; belowVA[c] = 0.f; }
        rowLive = !(cy > liveHi || cy < liveLo);
        if (!rowLive) return;                                    // wave-uniform: every cell of this row misses the image
        float sA[4], sVA[4 * NC];
        // (lanes beyond cell column dW -- the last strip's -- evaluate that column once more instead of sitting out: nobody reads their
        // sums, and the wave has no divergent branch around the cell)
        const int unc = eval(cx < dW ? cx : dW, cy, sA, sVA, upOnly) ? 1 : 0;
        ownA = sA[CELL_O] + from_next_lane(sA[CELL_W]);
        belowA = sA[CELL_N] + from_next_lane(sA[CELL_NW]);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ownVA[c] = sVA[CELL_O * NC + c] + from_next_lane(sVA[CELL_W * NC + c]);
            belowVA[c] = sVA[CELL_N * NC + c] + from_next_lane(sVA[CELL_NW * NC + c]);
        }
        rowU = unc | from_next_lane(unc);                        // the two cells of this row that feed column cx
    };
    const bool active = y0 < y1;
    float carryA = 0.f, carryVA[NC];                             // own + W of the row above, waiting for this row's N / NW parts
#pragma unroll
    for (int c = 0; c < NC; ++c) { carryVA[c] = 0.f; ownVA[c] = 0.f; belowVA[c] = 0.f; }
    int carryU = 0;
    // ONE loop over the cell rows y0 .. y1 (one copy of the cell evaluation in the code): the first iteration parks its N / NW parts
    // and meets the other waves at the barrier, the last one takes them from the wave below unless this wave owns the bottom row
    for (int cy = y0;; ++cy) {
        const bool first = cy == y0, last = cy == y1;             // (an empty segment: both)
        decltype(look(cx, cy, false)) seen = {};
        if (!first) seen = look(cx, cy - 1, column);
        if (active && (!last || ownsBottom)) row(cy, last);
        else if (active) {
            belowA = hand.a[wave + 1][lane]; rowU = hand.u[wave + 1][lane]; rowLive = true;
#pragma unroll
            for (int c = 0; c < NC; ++c) belowVA[c] = hand.va[wave + 1][c][lane];
        }
        if (first) {
            if (active) {
                hand.a[wave][lane] = belowA; hand.u[wave][lane] = rowU;
#pragma unroll
                for (int c = 0; c < NC; ++c) hand.va[wave][c][lane] = belowVA[c];
            }
            __syncthreads();
            if (!active) break;
        } else if (column) {
            // (a pixel row between two cell rows that miss the image -- the corners of a rotated canvas, half of config 5's rows --
            // has no area: its zeros go out without the sums and the division; the knife scan may still have listed such a pixel)
            float VA[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) VA[c] = carryVA[c] + belowVA[c];
            if (!rowLive && !aboveLive) {
#pragma unroll
                for (int c = 0; c < NC; ++c) VA[c] = 0.f;
                emit(cx, cy - 1, 0.f, VA, 0, seen);
            } else emit(cx, cy - 1, carryA + belowA, VA, carryU | rowU, seen);
        }
        if (last) break;
        carryA = ownA; carryU = rowU; aboveLive = rowLive;
#pragma unroll
        for (int c = 0; c < NC; ++c) carryVA[c] = ownVA[c];
    }
}

// The same walk with a wave of 32 cell columns x 2 cell rows: lanes 0-31 evaluate cell row c, lanes 32-63 cell row c + 1 of the same 32
// columns, c = y0, y0 + 2, ... y1.  Every lane finishes the dst pixel ABOVE its cell, (cx, cy - 1) = the own + W parts of cell row cy - 1
// (held by the other half-wave: one exchange across the halves per step serves both directions -- the upper half uses the lower half's
// sums of THIS step, the lower half keeps the upper half's for the NEXT one) + the N + NW parts of its own cell.  Cell row y1 comes
// from the wave below (its first row's parts, parked in LDS before the workgroup's one barrier) unless this wave owns the bottom row.
// Per-cell arithmetic, the order of the additions and therefore every result are those of cell_walk.
// (WR = 2; the code is written for WR sub-rows of 64 / WR lanes: sub-row j takes the sums of sub-row j - 1 through a rotation by 64 / WR
// lanes, sub-row 0 keeps what the last sub-row sends it for the next step.)
template <int NC, int WR, typename Look, typename Eval, typename Emit>
__device__ __forceinline__ void cell_walk2(int dW, int x0, int y0, int y1, bool ownsBottom, int wave, CellHandoffN<NC> &hand, int liveLo, int liveHi, int lane,
                                           Look look, Eval eval, Emit emit)
{
    static_assert(NC == 1, "the multi-row waves serve plain images");
    constexpr int LANES = 64 / WR;
    const int half = lane / LANES, hl = lane & (LANES - 1);     // (half: the lane's sub-row of the step)
    const int cx = x0 + hl;
    const bool column = hl < LANES - 1 && cx < dW;              // this lane's column is one the wave completes
    const int last = ownsBottom ? y1 : y1 - 1;                  // last cell row this wave evaluates itself
    const bool active = y0 < y1;
    const int other = ((lane - LANES) & 63) << 2;                // ds_bpermute address of the same column in the sub-row above (sub-row 0: the last one)
    float keptA = 0.f, keptVA = 0.f;                             // lower half: own + W of cell row c - 1, handed down by the upper half a step ago
    int keptU = 0;
    bool prevDead = false;                                       // wave-uniform: the previous step's two cell rows cannot touch the image (cell_live_rows)
    for (int c = y0;; c += WR) {
        const bool first = c == y0;
        const int cy = c + half;
        const int py = cy - 1;                                   // the dst row this lane finishes in this step
        const bool emits = column && py >= y0 && py < y1;
        decltype(look(cx, cy, false)) seen = {};
        if (active) seen = look(cx, py, emits);
        float ownA = 0.f, ownVA = 0.f, belowA = 0.f, belowVA = 0.f;
        int rowU = 0;
        // (wave-uniform: neither cell row of this step is one the wave evaluates, or both miss the image -- cell_live_rows)
        const bool dead = c > liveHi || c + WR - 1 < liveLo;
        const bool live = active && c <= last && !dead;
        if (live) {
            float sA[4], sVA[4];
            // (lanes beyond cell column dW, or whose row is not the wave's to evaluate, repeat a cell of the wave instead of sitting out:
            // nobody reads their sums, and the wave has no divergent branch around the cell)
            const bool mine = cy <= last;
            const int unc = eval(cx < dW ? cx : dW, mine ? cy : last, sA, sVA, cy == y1) ? 1 : 0;
            ownA = sA[CELL_O] + from_next_lane(sA[CELL_W]); ownVA = sVA[CELL_O] + from_next_lane(sVA[CELL_W]);
            belowA = sA[CELL_N] + from_next_lane(sA[CELL_NW]); belowVA = sVA[CELL_N] + from_next_lane(sVA[CELL_NW]);
            rowU = unc | from_next_lane(unc);
            if (!mine) { ownA = 0.f; ownVA = 0.f; belowA = 0.f; belowVA = 0.f; rowU = 0; }
        }
        if (first) {
            if (active && half == 0) { hand.a[wave][hl] = belowA; hand.va[wave][0][hl] = belowVA; hand.u[wave][hl] = rowU; }
            __syncthreads();
            if (!active) break;
        }
        // cell row y1 of a wave that does not own the bottom row: the parts the wave below parked (after the barrier, also in the first step)
        if (cy == y1 && !ownsBottom) { belowA = hand.a[wave + 1][hl]; belowVA = hand.va[wave + 1][0][hl]; rowU = hand.u[wave + 1][hl]; }
        // own + W of the other half-wave's cell row: the upper half needs the lower half's of THIS step, the lower half keeps the upper
        // half's for the next step
        const float swapA = __int_as_float(__builtin_amdgcn_ds_bpermute(other, __float_as_int(ownA)));
        const float swapVA = __int_as_float(__builtin_amdgcn_ds_bpermute(other, __float_as_int(ownVA)));
        const int swapU = __builtin_amdgcn_ds_bpermute(other, rowU);
        if (emits) {
            const float aboveA = half ? swapA : keptA, aboveVA = half ? swapVA : keptVA;
            const int aboveU = half ? swapU : keptU;
            // (a pixel row between cell rows that all miss the image -- the corners of a rotated canvas -- has no area: its zeros go
            // out without the sums and the division; the knife scan may still have listed such a pixel)
            const float zero[1] = {0.f}, total[1] = {aboveVA + belowVA};
            if (dead && prevDead) emit(cx, py, 0.f, zero, 0, seen);
            else emit(cx, py, aboveA + belowA, total, aboveU | rowU, seen);
        }
        keptA = swapA; keptVA = swapVA; keptU = swapU; prevDead = dead;
        if (c + WR > y1) break;
    }
}

// dst rows [y0, y1) of wave `wave` of the workgroup whose rows start at blockY0 and end before blockY1; ownsBottom: no wave below
// it in the workgroup has rows
__device__ __forceinline__ void cell_segment(int blockY0, int blockY1, int rowsPerWave, int wave, int &y0, int &y1, bool &ownsBottom)
{
    y0 = min(blockY0 + wave * rowsPerWave, blockY1);
    y1 = min(y0 + rowsPerWave, blockY1);
    ownsBottom = wave == kCellWaves - 1 || y1 >= blockY1;
}

// WR: the wave's shape (cell_wave_rows): 1 = 64 cell columns x 1 cell row per step, 2 = 32 x 2
template <typename T, int WIN, bool SCALED, bool HP, int WR>
__global__ __launch_bounds__(kQuadBlock, cell_min_waves(WIN)) __attribute__((amdgpu_num_sgpr(AAI_CELL_SGPRS))) void aai_cell_kernel(
    RotLaunch r, QuadConsts<float> q, CellConsts<float> z, CellLive live, QuadMap m, const T *__restrict__ src, ImageView sv, float *__restrict__ dst, ImageView dv,
    const unsigned long long *__restrict__ skipMasks, int tilesX, int rowsPerWave, int bigBlocks, int tailRows, int xcdRows, int rowBlocks)
{
    constexpr int kCellLanes = cell_wave_lanes(WR), kCellCols = cell_wave_cols(WR);
    __shared__ float window[WIN * WIN][kQuadBlock];
    __shared__ CellHandoff hand;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y;
    // XCD-aware order (xcd_tile, aai_quad_src.hpp): XCD x owns bands of `xcdRows` row blocks and walks them strip by strip, so that the
    // source lines neighbouring strips share are fetched through one L2
    xcd_tile(xcdRows, bx, by);
    if (by >= rowBlocks) return;                               // (block-uniform; rows the padded grid adds)
    const int x0 = bx * kCellCols;                             // (block-uniform: every wave reaches the walk's barrier)
    // the last workgroups of a launch are shorter (tailRows rows per wave instead of rowsPerWave): the waves that finish it live a
    // fraction as long, and the chip drains in a fraction of the time
    const int rpw = by < bigBlocks ? rowsPerWave : tailRows;
    const int blockY0 = r.dyBase + (by < bigBlocks ? by * (kCellWaves * rowsPerWave) : bigBlocks * (kCellWaves * rowsPerWave) + (by - bigBlocks) * (kCellWaves * tailRows));
    const int blockY1 = min(blockY0 + kCellWaves * rpw, r.dyEnd);
    int y0, y1;
    bool ownsBottom;
    cell_segment(blockY0, blockY1, rpw, wave, y0, y1, ownsBottom);           // dst rows [y0, y1); cell rows y0 .. y1
    float *image = dst + (int64_t)blockIdx.z * dv.imageStride;
    const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
    const CellColumn col = cell_column(r, z, min(x0 + (lane & (kCellLanes - 1)), r.dW));      // (cell_walk: lanes beyond column dW repeat it)
    int liveLo, liveHi;
    cell_live_rows(live, x0, x0 + kCellCols, liveLo, liveHi);  // wave-uniform
    // Images of 4 GiB and more (QuadMap::rebaseWaves): this wave's base pointer moves to the first source row its cells can touch -- the
    // zone centres are affine in the cell, so the extremes along the strided axis sit at the wave's corner cells (first / last lane, first
    // / last cell row); +- WIN + 2 lattice points for the windows; clamped onto the lattice like the positions themselves
    uint32_t rebase = 0;
    if (m.rebaseWaves) {
        const bool rowsAlongX = m.strideX > m.strideY;         // the window axis that walks the source rows
        int Zx, Zy;
        double dfx, dfy;
        cell_anchor<true>(r, col, y0, Zx, Zy, dfx, dfy);
        const int a0 = rowsAlongX ? Zx : Zy;
        cell_anchor<true>(r, col, y1, Zx, Zy, dfx, dfy);
        const int a1 = rowsAlongX ? Zx : Zy;
        const int lo = min(a0, a1), hi = max(a0, a1);
        const int mN = rowsAlongX ? r.mW : r.mH, nS = rowsAlongX ? m.nX : m.nY;
        const int wlo = min(__builtin_amdgcn_readlane(lo, 0), __builtin_amdgcn_readlane(lo, kCellLanes - 1)) - (WIN + 2);
        const int whi = max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, kCellLanes - 1)) + (WIN + 2);
        const int clo = min(max(wlo, 0), mN - 1) / m.scale, chi = min(max(whi, 0), mN - 1) / m.scale;          // source rows (replication)
        const int first = (rowsAlongX ? m.flipX : m.flipY) ? nS - 1 - chi : clo;
        const int64_t bytes = (int64_t)first * (rowsAlongX ? m.strideX : m.strideY) * (int64_t)sizeof(T);
        img += bytes;
        rebase = (uint32_t)bytes;
    }
    // per-pixel masks only where a 16 x 16 tile this segment touches holds a flagged pixel (QuadMap::tileFlags; wave-uniform, scalar loads)
    const unsigned long long *masks = skipMasks;
    if (masks && m.tileFlags) {
        bool any = false;
        for (int tr = y0 >> 4; tr <= (y1 - 1) >> 4; ++tr) any = any || tiles_flagged(m.tileFlags, m.tileFlagWords, tr, x0 >> 4, kCellCols / 16 + 2);
        if (!any) masks = nullptr;
    }
    auto look = [&](int px, int py, bool wanted) -> bool {
            // is the pixel one the plan's scans left to the fix-up pass?  (requested here, used after the cell is evaluated)
            return masks && wanted && ((masks[flag_word(px, py, tilesX)] >> flag_bit(px, py)) & 1ull);
        };
    auto eval = [&](int cx, int cy, float (&sA)[4], float (&sVA)[4], bool upOnly) -> bool {
            int Zx, Zy;
            double dfx, dfy;
            cell_anchor<true>(r, col, cy, Zx, Zy, dfx, dfy);
            QuadSrc<T, WIN, SCALED, true> s;
            s.img = img; s.m = &m; s.mW = r.mW; s.mH = r.mH; s.lds = window; s.tid = tid; s.rebase = rebase;
            // (the workgroup's bottom cell row only feeds its last pixel row: its interior / left-edge zones are skipped)
            cell_eval<float, WIN, false, HP>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA, upOnly);
            return false;
        };
    auto emit = [&](int px, int py, float A, const float (&VA)[1], int, bool skip) {
            if (skip) return;                                  // a pixel the plan's scans left to the fix-up pass is not written here
            // (written once, never read back: around the caches -- 1 % at configs 3 and 5)
            __builtin_nontemporal_store(A > 0.f ? VA[0] / A : 0.f, image + ((int64_t)(py - r.dyBase) * dv.rowStride + px));         // Source.cpp:577
        };
    if constexpr (WR >= 2) cell_walk2<1, WR>(r.dW, x0, y0, y1, ownsBottom, wave, hand, liveLo, liveHi, lane, look, eval, emit);
    else cell_walk<1>(r.dW, x0, y0, y1, ownsBottom, wave, hand, liveLo, liveHi, lane, look, eval, emit);
}

// Interleaved channels (2 .. 4 per pixel): the same walk with one sum of area x value per channel -- the areas of a (dst, src) pair are
// computed once and applied to every channel -- over QuadSrcMulti's window, whose slots hold all channels of their pixel as WORDS raw
// words (8-bit RGB(A): one word, like a plain image).  Wherever cell_can_serve says no, interleaved requests keep aai_quad_multi_kernel;
// no rebasing (below 4 GiB).  Dynamic LDS: WIN * WIN * WORDS KiB.
constexpr int cell_multi_min_waves(int win, int words)
{
    return 160 / (win * win * words + 7) >= 6 ? 6 : (160 / (win * win * words + 7) >= 1 ? 160 / (win * win * words + 7) : 1);
}
template <typename T, int WIN, bool SCALED, bool HP, int WORDS>
__global__ __launch_bounds__(kQuadBlock, cell_multi_min_waves(WIN, WORDS)) void aai_cell_multi_kernel(
    RotLaunch r, QuadConsts<float> q, CellConsts<float> z, CellLive live, QuadMap m, const T *__restrict__ src, ImageView sv, float *__restrict__ dst, ImageView dv,
    const unsigned long long *__restrict__ skipMasks, int tilesX, int rowsPerWave, int xcdRows, int rowBlocks)
{
    constexpr int kCellCols = cell_wave_cols(1);
    extern __shared__ unsigned windowWords[];
    __shared__ CellHandoffN<kQuadMaxChan> hand;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y;
    xcd_tile(xcdRows, bx, by);
    if (by >= rowBlocks) return;                               // (block-uniform; rows the padded grid adds)
    const int x0 = bx * kCellCols;
    const int blockY0 = r.dyBase + by * (kCellWaves * rowsPerWave);
    const int blockY1 = min(blockY0 + kCellWaves * rowsPerWave, r.dyEnd);
    int y0, y1;
    bool ownsBottom;
    cell_segment(blockY0, blockY1, rowsPerWave, wave, y0, y1, ownsBottom);
    const int chan = r.chan;
    float *image = dst + (int64_t)blockIdx.z * dv.imageStride;
    const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
    const CellColumn col = cell_column(r, z, min(x0 + lane, r.dW));
    int liveLo, liveHi;
    cell_live_rows(live, x0, x0 + kCellCols, liveLo, liveHi);
    const unsigned long long *masks = skipMasks;
    if (masks && m.tileFlags) {
        bool any = false;
        for (int tr = y0 >> 4; tr <= (y1 - 1) >> 4; ++tr) any = any || tiles_flagged(m.tileFlags, m.tileFlagWords, tr, x0 >> 4, kCellCols / 16 + 2);
        if (!any) masks = nullptr;
    }
    cell_walk<kQuadMaxChan>(r.dW, x0, y0, y1, ownsBottom, wave, hand, liveLo, liveHi, lane,
        [&](int px, int py, bool wanted) -> bool {
            return masks && wanted && ((masks[flag_word(px, py, tilesX)] >> flag_bit(px, py)) & 1ull);
        },
        [&](int cx, int cy, float (&sA)[4], float (&sVA)[4 * kQuadMaxChan], bool upOnly) -> bool {
            int Zx, Zy;
            double dfx, dfy;
            cell_anchor<true>(r, col, cy, Zx, Zy, dfx, dfy);
            QuadSrcMulti<T, WIN, SCALED, WORDS> s;
            s.img = img; s.m = &m; s.mW = r.mW; s.mH = r.mH; s.chan = chan; s.lds = windowWords; s.tid = tid;
            cell_eval<float, WIN, false, HP, kQuadMaxChan>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA, upOnly);
            return false;
        },
        [&](int px, int py, float A, const float (&VA)[kQuadMaxChan], int, bool skip) {
            if (skip) return;                                  // a pixel the plan's scans left to the fix-up pass is not written here
            float *out = image + ((int64_t)(py - r.dyBase) * dv.rowStride + (int64_t)px * chan);
#pragma unroll
            for (int c = 0; c < kQuadMaxChan; ++c)
                if (c < chan) out[c] = A > 0.f ? VA[c] / A : 0.f;                                  // Source.cpp:577
        });
}

// Once per geometry: the same walk without pixel loads.  Sets the bit of every dst pixel one of whose four cells has a
// decision too close to its threshold, or whose total area is too small for fp32 weights (QuadConsts::minArea), in the
// lane masks of the 16 x 16 tiling (on top of the knife-edge scan's bits) and counts the newly set bits in counter[0].
template <int WIN, bool HP>
__global__ __launch_bounds__(kQuadBlock) void aai_cell_scan_kernel(RotLaunch r, QuadConsts<float> q, CellConsts<float> z, CellLive live, unsigned long long *__restrict__ laneMasks,
                                                                  unsigned *__restrict__ counter, int tilesX, int rowsPerWave, int band0)
{
    constexpr int kCellLanes = cell_wave_lanes(1), kCellCols = cell_wave_cols(1);      // (the flags do not depend on the wave's shape: the scan keeps 64 x 1)
    __shared__ CellHandoff hand;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * kCellCols;
    const int blockY0 = (band0 + blockIdx.y) * (kCellWaves * rowsPerWave);
    const int blockY1 = min(blockY0 + kCellWaves * rowsPerWave, r.dH);
    int y0, y1;
    bool ownsBottom;
    cell_segment(blockY0, blockY1, rowsPerWave, wave, y0, y1, ownsBottom);
    const CellColumn col = cell_column(r, z, min(x0 + (lane & (kCellLanes - 1)), r.dW));
    int liveLo, liveHi;
    cell_live_rows(live, x0, x0 + kCellCols, liveLo, liveHi);
    cell_walk<1>(r.dW, x0, y0, y1, ownsBottom, wave, hand, liveLo, liveHi, lane,
        [&](int, int, bool) -> int { return 0; },
        [&](int cx, int cy, float (&sA)[4], float (&sVA)[4], bool) -> bool {
            int Zx, Zy;
            double dfx, dfy;
            if (!cell_anchor(r, col, cy, Zx, Zy, dfx, dfy)) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { sA[t] = 0.f; sVA[t] = 0.f; }
                return false;
            }
            NoSrc s;
            return cell_eval<float, WIN, true, HP>(q, z, Zx, Zy, dfx, dfy, r.mW, r.mH, s, sA, sVA);
        },
        [&](int px, int py, float A, const float (&)[1], int uncertain, int) {
            if (uncertain | ((A > 0.f && A < q.minArea) ? 1 : 0)) {
                const unsigned long long bit = 1ull << flag_bit(px, py);
                const unsigned long long old = atomicOr(laneMasks + flag_word(px, py, tilesX), bit);
                if (!(old & bit)) atomicAdd(counter, 1u);
            }
        });
}

// HPSEL: which precision variants this translation unit holds: 0 = plain fp32 only, 1 = hiPrec only, 2 = both
template <typename T, int WIN, int HPSEL>
hipError_t launch_cell_tile(const RotLaunch &r, const QuadConsts<float> &q, const CellConsts<float> &z, const QuadMap &m, const T *src, ImageView sv,
                            float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    const int rows = r.dyEnd - r.dyBase;
    // the wave's shape (cell_wave_rows; experiments build: AAI_CELL_WAVE=1 / 2)
    int waveRows = cell_wave_rows(r.side, m.scale, r.c, r.s);
    {
        const char *e = experiment_env("AAI_CELL_WAVE");
        if (e && m.scale <= 1 && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4)) waveRows = atoi(e);      // (4: 16 x 4, experiments build only)
    }
    const int rowsPerWave = cell_rows_per_wave(r.dW, rows, batch, r.side / (m.scale > 0 ? m.scale : 1), waveRows);
    const int strips = (r.dW + cell_wave_cols(waveRows) - 1) / cell_wave_cols(waveRows);
    const int blockRows = kCellWaves * rowsPerWave;
    // (Shorter segments for the last rows of a launch -- waves that live half as long, so that the chip drains sooner -- paid with
    // 8-row strips per wave; with four waves per strip segment they measure nothing: config 3 160.7 us without, 161.0 ... 166.3 with.
    // Experiments (-DAAI_EXPERIMENTS): AAI_CELL_TAIL="<percent of the rows>,<rows per wave in the tail>".)
    int tailPct = 0, tailR = 2;
    {
        const char *e = experiment_env("AAI_CELL_TAIL");
        if (e) { tailPct = atoi(e); const char *c = strchr(e, ','); tailR = c ? atoi(c + 1) : 2; }
    }
    int bigBlocks = (rows + blockRows - 1) / blockRows, tailRows = rowsPerWave, tailBlocks = 0;
    if (tailPct > 0 && tailR > 0 && tailR < rowsPerWave) {
        bigBlocks = (int)((int64_t)rows * (100 - tailPct) / 100 / blockRows);
        tailRows = tailR;
        tailBlocks = (rows - bigBlocks * blockRows + kCellWaves * tailRows - 1) / (kCellWaves * tailRows);
        if (bigBlocks + tailBlocks > 65535) { bigBlocks = (rows + blockRows - 1) / blockRows; tailRows = rowsPerWave; tailBlocks = 0; }      // grid.y
    }
    const int rowBlocks = bigBlocks + tailBlocks;
    // XCD-aware workgroup order: config 3 is bound by instruction issue and does not move (170.0 us at 0 / 1 / 2 / 4 row blocks per
    // band), but geometries whose waves wait for memory do: 2:1 at 45 degrees 312 -> 275 / 260 / 257 us, 4:1 at 30 degrees 275 -> 270;
    // replicated sources (config 5: one source row feeds four dst rows) lose 1-4 % beyond one block and gain nothing: off there.
    // (profiles/r04_fast_xcd.txt; experiments build: AAI_XCD_ROWS.)
    const int band = xcd_band(m.scale > 1 ? 0 : kCellXcdRowsDefault);
    const int gy = xcd_grid_rows(rowBlocks, band);
    const int xcdRows = gy ? band : 0;
    const dim3 grid(strips, gy ? gy : rowBlocks, batch);
    const int tilesX = (r.dW + 15) / 16;
    const CellLive live = make_cell_live(r, z);
#define AAI_CELL_LAUNCH(SCALED, HP, WR) \
    hipLaunchKernelGGL((aai_cell_kernel<T, WIN, SCALED, HP, WR>), grid, dim3(kQuadBlock), 0, stream, r, q, z, live, m, src, sv, dst, dv, skipMasks, tilesX, rowsPerWave, bigBlocks, tailRows, xcdRows, rowBlocks)
    if ((q.hiPrec != 0) != (HPSEL == 1) && HPSEL != 2) return hipErrorInvalidValue;       // (the dispatcher picks the unit that holds the variant)
    if (m.scale > 1) {                                          // (replicated sources: the 64 x 1 wave only)
        if (HPSEL != 0 && q.hiPrec) AAI_CELL_LAUNCH(true, true, 1);
        if (HPSEL != 1 && !q.hiPrec) AAI_CELL_LAUNCH(true, false, 1);
    } else if (waveRows == 2) {
        if (HPSEL != 0 && q.hiPrec) AAI_CELL_LAUNCH(false, true, 2);
        if (HPSEL != 1 && !q.hiPrec) AAI_CELL_LAUNCH(false, false, 2);
#if defined(AAI_EXPERIMENTS)
    } else if (waveRows == 4) {
        if (HPSEL != 0 && q.hiPrec) AAI_CELL_LAUNCH(false, true, 4);
        if (HPSEL != 1 && !q.hiPrec) AAI_CELL_LAUNCH(false, false, 4);
#endif
    } else {
        if (HPSEL != 0 && q.hiPrec) AAI_CELL_LAUNCH(false, true, 1);
        if (HPSEL != 1 && !q.hiPrec) AAI_CELL_LAUNCH(false, false, 1);
    }
#undef AAI_CELL_LAUNCH
    return hipGetLastError();
}

template <typename T, int HPSEL>
hipError_t launch_cell_typed(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch,
                             const unsigned long long *skipMasks, hipStream_t stream)
{
    const QuadConsts<float> q = make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    switch (z.win) {
    case 2: return launch_cell_tile<T, 2, HPSEL>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 3: return launch_cell_tile<T, 3, HPSEL>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 4: return launch_cell_tile<T, 4, HPSEL>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 5: return launch_cell_tile<T, 5, HPSEL>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 6: return launch_cell_tile<T, 6, HPSEL>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 7: return launch_cell_tile<T, 7, HPSEL>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 8: return launch_cell_tile<T, 8, HPSEL>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    default: return hipErrorInvalidValue;
    }
}

// LDS words one window slot takes with `chan` interleaved channels of T (QuadSrcMulti)
template <typename T> constexpr int cell_slot_words(int chan) { return sizeof(T) == 4 ? chan : (sizeof(T) == 2 ? (chan + 1) / 2 : 1); }

template <typename T, int WIN, int WORDS>
hipError_t launch_cell_multi_words(const RotLaunch &r, const QuadConsts<float> &q, const CellConsts<float> &z, const QuadMap &m, const T *src, ImageView sv,
                                   float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    if constexpr (WIN * WIN * WORDS > kCellMultiMaxKiB) return hipErrorInvalidValue;          // (cell_can_serve keeps such windows on the quad kernel)
    else {
        const int rows = r.dyEnd - r.dyBase;
        const int rowsPerWave = cell_rows_per_wave(r.dW, rows, batch, r.side / (m.scale > 0 ? m.scale : 1));
        const int strips = (r.dW + cell_wave_cols(1) - 1) / cell_wave_cols(1);
        const int rowBlocks = (rows + kCellWaves * rowsPerWave - 1) / (kCellWaves * rowsPerWave);
        const int band = xcd_band(m.scale > 1 ? 0 : kCellXcdRowsDefault);
        const int gy = xcd_grid_rows(rowBlocks, band);
        const int xcdRows = gy ? band : 0;
        const dim3 grid(strips, gy ? gy : rowBlocks, batch);
        const int tilesX = (r.dW + 15) / 16;
        const CellLive live = make_cell_live(r, z);
        const size_t lds = (size_t)WIN * WIN * WORDS * kQuadBlock * sizeof(unsigned);
#define AAI_CELL_MULTI_LAUNCH(SCALED, HP)                                                                                                                     \
    {                                                                                                                                                         \
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&aai_cell_multi_kernel<T, WIN, SCALED, HP, WORDS>),                  \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);                                         \
        (void)once;                                                                                                                                           \
        hipLaunchKernelGGL((aai_cell_multi_kernel<T, WIN, SCALED, HP, WORDS>), grid, dim3(kQuadBlock), lds, stream, r, q, z, live, m, src, sv, dst, dv,         \
                           skipMasks, tilesX, rowsPerWave, xcdRows, rowBlocks);                                                                               \
    }
        if (m.scale > 1) {
            if (q.hiPrec) AAI_CELL_MULTI_LAUNCH(true, true) else AAI_CELL_MULTI_LAUNCH(true, false)
        } else {
            if (q.hiPrec) AAI_CELL_MULTI_LAUNCH(false, true) else AAI_CELL_MULTI_LAUNCH(false, false)
        }
#undef AAI_CELL_MULTI_LAUNCH
        return hipGetLastError();
    }
}

template <typename T, int WIN>
hipError_t launch_cell_multi_win(const RotLaunch &r, const QuadConsts<float> &q, const CellConsts<float> &z, const QuadMap &m, const T *src, ImageView sv,
                                 float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    const int words = cell_slot_words<T>(r.chan);
    if constexpr (sizeof(T) == 1) return launch_cell_multi_words<T, WIN, 1>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    else if constexpr (sizeof(T) == 2) {
        if (words == 1) return launch_cell_multi_words<T, WIN, 1>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
        return launch_cell_multi_words<T, WIN, 2>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    } else {
        if (words == 2) return launch_cell_multi_words<T, WIN, 2>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
        if (words == 3) return launch_cell_multi_words<T, WIN, 3>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
        return launch_cell_multi_words<T, WIN, 4>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    }
}

template <typename T>
hipError_t launch_cell_multi_typed(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch,
                                   const unsigned long long *skipMasks, hipStream_t stream)
{
    const QuadConsts<float> q = make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    switch (z.win) {
    case 2: return launch_cell_multi_win<T, 2>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 3: return launch_cell_multi_win<T, 3>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 4: return launch_cell_multi_win<T, 4>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 5: return launch_cell_multi_win<T, 5>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 6: return launch_cell_multi_win<T, 6>(r, q, z, m, src, sv, dst, dv, batch, skipMasks, stream);
    default: return hipErrorInvalidValue;                      // (cell_can_serve: windows up to 6 x 6)
    }
}

}  // namespace

// ---- translation units ----------------------------------------------------------------------------------------------------------
// The runtime loads a translation unit's code object when one of its kernels is first launched -- ~1 ms per 100 KB of code, once per
// process, and the first call of a process is the only one the reference's user makes (Source.cpp:1565).  All 84 instantiations of
// aai_cell_kernel in one unit made that 14 ms (profiles/r03_plan_time.txt: config 3's plan 15.4 ms cold, 0.8 ms with the unit loaded).
// The Makefile therefore compiles this file once per AAI_CELL_PART: 1 = dispatch + scan kernels, 2 = fp32 sources, 3 = fp32 sources
// with hiPrec (rotations within a few degrees of an axis), 4 = 8-bit, 5 = 16-bit sources; a process pays for the unit it uses.
// (AAI_CELL_PART undefined = everything in one unit.)
hipError_t launch_cell_f32(const RotLaunch &r, const QuadMap &m, const float *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream);
hipError_t launch_cell_f32_hp(const RotLaunch &r, const QuadMap &m, const float *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream);
hipError_t launch_cell_u8(const RotLaunch &r, const QuadMap &m, const unsigned char *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream);
hipError_t launch_cell_u16(const RotLaunch &r, const QuadMap &m, const unsigned short *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream);

#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 2
hipError_t launch_cell_f32(const RotLaunch &r, const QuadMap &m, const float *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    return launch_cell_typed<float, 0>(r, m, src, sv, dst, dv, batch, skipMasks, stream);
}
#endif
#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 3
hipError_t launch_cell_f32_hp(const RotLaunch &r, const QuadMap &m, const float *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    return launch_cell_typed<float, 1>(r, m, src, sv, dst, dv, batch, skipMasks, stream);
}
#endif
#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 4
hipError_t launch_cell_u8(const RotLaunch &r, const QuadMap &m, const unsigned char *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    return launch_cell_typed<unsigned char, 2>(r, m, src, sv, dst, dv, batch, skipMasks, stream);
}
#endif
#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 5
hipError_t launch_cell_u16(const RotLaunch &r, const QuadMap &m, const unsigned short *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    return launch_cell_typed<unsigned short, 2>(r, m, src, sv, dst, dv, batch, skipMasks, stream);
}
#endif

hipError_t launch_cell_multi_f32(const RotLaunch &r, const QuadMap &m, const float *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream);
hipError_t launch_cell_multi_u8(const RotLaunch &r, const QuadMap &m, const unsigned char *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream);
hipError_t launch_cell_multi_u16(const RotLaunch &r, const QuadMap &m, const unsigned short *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream);
#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 6
hipError_t launch_cell_multi_f32(const RotLaunch &r, const QuadMap &m, const float *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    return launch_cell_multi_typed<float>(r, m, src, sv, dst, dv, batch, skipMasks, stream);
}
#endif
#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 7
hipError_t launch_cell_multi_u8(const RotLaunch &r, const QuadMap &m, const unsigned char *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    return launch_cell_multi_typed<unsigned char>(r, m, src, sv, dst, dv, batch, skipMasks, stream);
}
#endif
#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 8
hipError_t launch_cell_multi_u16(const RotLaunch &r, const QuadMap &m, const unsigned short *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    return launch_cell_multi_typed<unsigned short>(r, m, src, sv, dst, dv, batch, skipMasks, stream);
}
#endif

#if !defined(AAI_CELL_PART) || AAI_CELL_PART == 1
bool cell_can_serve(const RotLaunch &r, int srcType, ImageView sv)
{
    // plain images below 4 GiB (lanes address their pixels with unsigned 32-bit byte offsets from the image's first element)
    static const bool enabled = [] { const char *e = experiment_env("AAI_CELL"); return !(e && atoi(e) == 0); }();      // experiments: AAI_CELL=0 keeps the quad kernel
    if (!enabled || !r.cell || r.mode != AAI_MODE_AREA) return false;
    // Small outputs stay on the quad kernel: a cell wave lives for rows + 1 cell rows, and an image of fewer than ~1000 such
    // waves (about 720 x 720 dst pixels) cannot fill the chip with them -- the reference's own example call (158 x 158 dst
    // pixels at 5.9 : 1) takes 71 us on 60 cell waves and 36 us on 390 one-shot quad waves.
    // (AAI_POLICY_PREFER_CELL asks for the cell kernel all the same)
    if (!r.preferCell && (int64_t)((r.dW + 62) / 63) * ((r.dH + 7) / 8) < 1024) return false;      // (in waves of 63 columns x 8 rows, whatever the wave's shape)
    const int64_t esz = srcType == SRC_U8 ? 1 : srcType == SRC_U16 ? 2 : 4;
    if (r.chan > 1) {
        // interleaved channels (aai_cell_multi_kernel, 64 x 1 wave): windows whose slots fit 64 KiB of LDS, below 4 GiB
        if (r.chan > kQuadMaxChan) return false;
        // Where it wins (tools/interleaved_bench.py, profiles/r04_interleaved.txt): replicated sources of any type (x2 up-sampling of RGB
        // fp32 at 30 degrees: 1.48 ms against the quad kernel's 2.78) and fp32 pixels (config 3's geometry, RGB: 0.72 against 0.77);
        // 8- / 16-bit pixels without replication stay on aai_quad_multi_kernel (RGB 8-bit: 0.49 ms there, 0.68 here -- one lane's window
        // of packed words is unpacked for four dst pixels' sums, and four channels' sums cross lanes and rows)
        if (esz != 4 && r.scale <= 1 && !r.preferCell) return false;
        const int words = esz == 4 ? r.chan : (esz == 2 ? (r.chan + 1) / 2 : 1);
        const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
        if (z.win > 6 || z.win * z.win * words > kCellMultiMaxKiB) return false;
        return (int64_t)r.H * sv.rowStride * esz < ((int64_t)1 << 32);
    }
    if ((int64_t)r.H * sv.rowStride * esz < ((int64_t)1 << 32)) return true;
    // 4 GiB and more: every wave rebases its offsets on its own first source row (aai_cell_kernel, QuadMap::rebaseWaves); the rows one
    // wave can touch -- 64 cell columns and up to 33 cell rows of `side` lattice points each, plus its windows -- must span less than
    // 4 GiB, and the 24-bit multiplies of the window addresses need a pitch below 8 MiB
    const int64_t span = (int64_t)((97.0 * r.side + 24.0) / (r.scale > 0 ? r.scale : 1)) + 2;
    return sv.rowStride * esz < ((int64_t)1 << 23) && span * sv.rowStride * esz < ((int64_t)1 << 32);
}

hipError_t launch_cell(const RotLaunch &r, const QuadMap &map, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    if (r.dW <= 0 || r.dyEnd <= r.dyBase || batch <= 0) return hipSuccess;
    QuadMap m = map;
    m.anchorRows = 0;
    {
        const int64_t esz = srcType == SRC_U8 ? 1 : srcType == SRC_U16 ? 2 : 4;
        m.rebaseWaves = (int64_t)r.H * sv.rowStride * esz >= ((int64_t)1 << 32) ? 1 : 0;
    }
    if (r.chan > 1) {
        m.rebaseWaves = 0;
        switch (srcType) {
        case SRC_U8: return launch_cell_multi_u8(r, m, static_cast<const unsigned char *>(src), sv, dst, dv, batch, skipMasks, stream);
        case SRC_U16: return launch_cell_multi_u16(r, m, static_cast<const unsigned short *>(src), sv, dst, dv, batch, skipMasks, stream);
        default: return launch_cell_multi_f32(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream);
        }
    }
    switch (srcType) {
    case SRC_U8: return launch_cell_u8(r, m, static_cast<const unsigned char *>(src), sv, dst, dv, batch, skipMasks, stream);
    case SRC_U16: return launch_cell_u16(r, m, static_cast<const unsigned short *>(src), sv, dst, dv, batch, skipMasks, stream);
    default:
        if (make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy).hiPrec) return launch_cell_f32_hp(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream);
        return launch_cell_f32(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream);
    }
}

hipError_t launch_cell_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream)
{
    if (r.dW <= 0 || r.dH <= 0) return hipSuccess;
    const QuadConsts<float> q = make_cell_quad_consts<float>(r.side, r.c, r.s, r.policy);
    const CellConsts<float> z = make_cell_consts<float>(r.side, r.c, r.s);
    const CellLive live = make_cell_live(r, z);
    const int rows = 4;                                        // per wave: workgroups of 16 dst rows
    const int strips = (r.dW + cell_wave_cols(1) - 1) / cell_wave_cols(1);
    const int tilesX = (r.dW + 15) / 16;
    const int bands = (r.dH + kCellWaves * rows - 1) / (kCellWaves * rows);
    for (int b0 = 0; b0 < bands; b0 += 65535) {                // grid.y carries at most 65535 bands
        const dim3 grid(strips, bands - b0 < 65535 ? bands - b0 : 65535, 1);
#define AAI_CELL_SCAN(W)                                                                                                                               \
    case W:                                                                                                                                            \
        if (q.hiPrec) hipLaunchKernelGGL((aai_cell_scan_kernel<W, true>), grid, dim3(kQuadBlock), 0, stream, r, q, z, live, laneMasks, counter, tilesX, rows, b0); \
        else hipLaunchKernelGGL((aai_cell_scan_kernel<W, false>), grid, dim3(kQuadBlock), 0, stream, r, q, z, live, laneMasks, counter, tilesX, rows, b0);    \
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

#endif      // AAI_CELL_PART 1

}  // namespace aai
