// aai_axis.hip -- K1: the axis-aligned (reduced rotation == 0) area-average kernel for gfx950.
//
// Replaces the reference's per-output-pixel loop (Source.cpp:413-579, and 868-907 for the fast mode)
// for rotations that are multiples of 90 degrees, where every (dst,src) pair the reference classifies is
// "whole", "none", a straight cut or a corner box, and the overlap area factors into
// (x overlap) * (y overlap).  The result is two normalised 1-D box filters whose windows and weights the
// host tabulates in double precision (aai_plan.cpp).  This is the bandwidth-bound, roofline-graded kernel:
// the source is read exactly once, coalesced, and the output written exactly once.
//
// Work decomposition (wave64, no MFMA -- this is a streaming weighted gather, not a contraction):
//   * one WAVE owns a strip = up to 256 consecutive source columns (one 16-byte load per lane per source
//     row = 1 KiB per wave-instruction) and the output pixels whose windows lie inside those columns;
//   * for each output row it accumulates the vertical pass in registers (one FMA per loaded element), with
//     all source rows of that output row in flight together;
//   * the row of vertical sums goes through a 1-KiB per-wave LDS line so that lanes can re-index it by
//     OUTPUT pixel for the horizontal pass (windows are not lane-aligned: e.g. 8192->2048 with the
//     reference's isocenter-anchored grid starts every window 2 columns into a float4);
//   * stores are one contiguous run per wave per output row.
//   A 256-thread workgroup = 4 adjacent strips (4 KiB contiguous per source row); waves never synchronise
//   with each other, so there is no s_barrier anywhere.
#include "aai_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace aai {

namespace {

constexpr int kWaves = 4;
constexpr int VEC = 4;         // source columns per lane and load: STRIP_COLS = 64 * VEC

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access
typedef float f4 __attribute__((ext_vector_type(4)));

// Raw<T>: VEC = 4 consecutive source columns of one row, as loaded (one 16-, 8- or 4-byte load per lane for fp32,
// 16-bit and 8-bit sources), and their widening to fp32.
// Loads only assume element alignment (x0 and the row stride are arbitrary).
// (Fatter strips for the narrow types -- 8 / 16 columns per lane so that every load is 16 bytes, with a padded LDS
// line and outputs interleaved across the wave -- were built and measured: 8192^2 u8 -> 2048^2 took 21.0 us per
// image against 21.2 us for this layout at 8 rows per workgroup; 16-bit sources got slower.  These kernels are
// bound by issue and wait time of the whole pipeline, not by load width; see profiles/r01_typed_sources.txt.)
// NT = nontemporal (streaming) load: the source is read exactly once, so it should not displace the
// tables and the output lines in L2 / Infinity Cache.  Measured on MI355X (tools/membw.hip): plain
// 16-byte reads stream at ~6.3 TB/s, nontemporal ones at ~7.1 TB/s.
template <typename T> struct Raw;

template <> struct Raw<float> {
    f4 q;
    __device__ __forceinline__ void zero() { q = (f4){0.f, 0.f, 0.f, 0.f}; }
    template <bool NT> __device__ __forceinline__ void load(const float *p)
    {
        const f4u *v = reinterpret_cast<const f4u *>(p);
        q = NT ? __builtin_nontemporal_load(v) : *v;
    }
    __device__ __forceinline__ float get(int i) const { return q[i]; }
};

template <> struct Raw<unsigned char> {
    typedef unsigned int word __attribute__((aligned(1)));
    unsigned w;
    __device__ __forceinline__ void zero() { w = 0u; }
    template <bool NT> __device__ __forceinline__ void load(const unsigned char *p)
    {
        const word *v = reinterpret_cast<const word *>(p);
        w = NT ? __builtin_nontemporal_load(v) : *v;
    }
    __device__ __forceinline__ float get(int i) const { return (float)((w >> (8 * i)) & 255u); }     // v_cvt_f32_ubyte<i>
};

template <> struct Raw<unsigned short> {
    typedef unsigned int word2 __attribute__((ext_vector_type(2), aligned(2)));
    unsigned w0, w1;
    __device__ __forceinline__ void zero() { w0 = 0u; w1 = 0u; }
    template <bool NT> __device__ __forceinline__ void load(const unsigned short *p)
    {
        const word2 *v = reinterpret_cast<const word2 *>(p);
        const word2 x = NT ? __builtin_nontemporal_load(v) : *v;
        w0 = x.x; w1 = x.y;
    }
    __device__ __forceinline__ float get(int i) const { return (float)(((i < 2 ? w0 : w1) >> (16 * (i & 1))) & 65535u); }
};

// A table entry unpacked into plain scalars (keeps it in registers: a struct copy of AxisEntry would be
// turned into a private array and promoted to LDS).
struct Win { int s0, s1; float wF, wM, wL; };

__device__ __forceinline__ Win load_win(const AxisEntry *__restrict__ tab, int k)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4 q = reinterpret_cast<const i4 *>(tab)[2 * k];
    const float wl = reinterpret_cast<const float *>(tab)[8 * k + 4];
    Win w;
    w.s0 = q.x; w.s1 = q.y; w.wF = __int_as_float(q.z); w.wM = __int_as_float(q.w); w.wL = wl;
    return w;
}

__device__ __forceinline__ float row_weight(const Win e, int y)
{
    return y == e.s0 ? e.wF : (y == e.s1 ? e.wL : e.wM);
}

struct Cols { float v[VEC]; };

// Vertical pass for one output row: sum_y w(y) * src[y][colc .. colc+VEC-1].  Up to four source rows are issued
// together; the window is wave-uniform, so the "row exists" branches are scalar and rows past the window are
// simply not loaded.
template <bool NT, typename T>
__device__ __forceinline__ Cols vertical_pass(const T *__restrict__ img, int64_t rowStride, int colc, const Win e)
{
    Cols acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] = 0.f;
    for (int y = e.s0; y <= e.s1; y += 4) {
        Raw<T> r0, r1, r2, r3;
        const T *p = img + (int64_t)y * rowStride + colc;
        r0.template load<NT>(p);
        r1.zero(); r2.zero(); r3.zero();
        if (y + 1 <= e.s1) r1.template load<NT>(p + rowStride);
        if (y + 2 <= e.s1) r2.template load<NT>(p + 2 * rowStride);
        if (y + 3 <= e.s1) r3.template load<NT>(p + 3 * rowStride);
        const float w0 = row_weight(e, y), w1 = row_weight(e, y + 1), w2 = row_weight(e, y + 2), w3 = row_weight(e, y + 3);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            acc.v[i] += w0 * r0.get(i);
            acc.v[i] += w1 * r1.get(i);
            acc.v[i] += w2 * r2.get(i);
            acc.v[i] += w3 * r3.get(i);
        }
    }
    return acc;
}

// The same in two halves, for callers that keep several output rows' loads in flight: issue_rows starts the loads of a window of
// at most four source rows (the caller checks), finish_rows does vertical_pass's arithmetic on them, operation for operation.
template <typename T, int NR> struct RowsN { Raw<T> r[NR]; };      // NR = 4 or 8 source rows
template <bool NT, typename T, int NR>
__device__ __forceinline__ void issue_rows(const T *__restrict__ img, int64_t rowStride, int colc, const Win e, RowsN<T, NR> &r)
{
    const T *p = img + (int64_t)e.s0 * rowStride + colc;
    r.r[0].template load<NT>(p);
#pragma unroll
    for (int k = 1; k < NR; ++k) {
        r.r[k].zero();
        if (e.s0 + k <= e.s1) r.r[k].template load<NT>(p + k * rowStride);      // (wave-uniform: rows past the window are not loaded)
    }
}
template <typename T, int NR>
__device__ __forceinline__ Cols finish_rows(const RowsN<T, NR> &r, const Win e)
{
    Cols acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] = 0.f;
#pragma unroll
    for (int g = 0; g < NR; g += 4) {
        // (vertical_pass stops after the group that holds the window's last row: groups past it add nothing there, and w x 0 here)
        if (g > 0 && e.s0 + g > e.s1) break;
        const int y = e.s0 + g;
        const float w0 = row_weight(e, y), w1 = row_weight(e, y + 1), w2 = row_weight(e, y + 2), w3 = row_weight(e, y + 3);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            acc.v[i] += w0 * r.r[g].get(i);
            acc.v[i] += w1 * r.r[g + 1].get(i);
            acc.v[i] += w2 * r.r[g + 2].get(i);
            acc.v[i] += w3 * r.r[g + 3].get(i);
        }
    }
    return acc;
}

// Park the lane's VEC vertical sums in the wave's LDS line (position = column - strip origin).  Near the right
// image edge a lane loaded the last in-range vector instead of its own (colc = min(col, W - VEC), shift =
// col - colc): it then holds columns [colc, colc + VEC) and writes only those at or right of its own first column
// (the others belong to the lane before it, which writes them itself).
__device__ __forceinline__ void park(float *line, int lane, int shift, const Cols &c)
{
    if (shift == 0) {
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const f4 v = {c.v[j], c.v[j + 1], c.v[j + 2], c.v[j + 3]};
            *reinterpret_cast<f4 *>(line + VEC * lane + j) = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (i >= shift) line[VEC * lane + i - shift] = c.v[i];
    }
}

// Horizontal pass for one output pixel from the wave's LDS line (indices relative to the strip origin).
// `step` = distance between taps in elements (1, or the channel count of an interleaved source).
__device__ __forceinline__ float horizontal_pass(const float *line, int off, int span, float wF, float wM, float wL, int step = 1)
{
    float s = wF * line[off];
    if (span > 0) {
        float mid = 0.f;
        for (int i = 1; i < span; ++i) mid += line[off + i * step];
        s += wM * mid + wL * line[off + span * step];
    }
    return s;
}

// NT: nontemporal source loads.
// A workgroup owns `rowsPerBlock` consecutive output rows of its 4 strips (or, with `interleave`, the rows
// blockIdx.y, blockIdx.y + gridDim.y, ...).  Measured on MI355X (tools/tune_axis.py, profiles/): ONE output
// row per workgroup is fastest by a wide margin (8192^2 -> 2048^2: 6.8 TB/s at 1 row, 6.3 at 2, 5.3 at 8,
// 4.8 at 32): workgroups are dispatched in grid order, so small workgroups make the whole chip sweep the
// image top to bottom together and the HBM pages of a source row are read by all CUs at about the same
// time, whereas tall workgroups open 2048 independent streams 32 KiB apart.  A software-pipelined variant
// (two output rows in flight per wave) bought 2 % at 2+ rows per workgroup and nothing at 1, so the kernel
// keeps the simple form: 4 source rows (4 KiB per wave) in flight, latency covered by 8 waves per SIMD.
// CH: interleaved channels (AxisLaunch::tapStep / outChan); false compiles the single-channel kernel unchanged.
template <bool NT, typename T, bool CH>
__global__ __launch_bounds__(kWaves * 64) void aai_axis_kernel(AxisLaunch a, const AxisEntry *__restrict__ laneTab,
                                                                const AxisEntry *__restrict__ rowTab, const AxisStrip *__restrict__ strips,
                                                                const T *__restrict__ src, ImageView sv,
                                                                float *__restrict__ dst, ImageView dv,
                                                                int rowsPerBlock, int interleave, int swapXY)
{
    __shared__ __attribute__((aligned(16))) float lds[kWaves][64 * VEC + 8];   // +8 keeps the lines 16-byte aligned and apart

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: strip/row tables load through the scalar cache
    // swapXY: the grid is launched (rows, strip blocks) so that consecutive workgroups -- which go to consecutive XCDs --
    // take consecutive output ROWS of one strip block instead of the strip blocks of one row
    const int bX = swapXY ? (int)blockIdx.y : (int)blockIdx.x, bY = swapXY ? (int)blockIdx.x : (int)blockIdx.y;
    const int gY = swapXY ? (int)gridDim.x : (int)gridDim.y;
    const int strip = bX * kWaves + wave;
    if (strip >= a.nStrips) return;   // waves are independent: no barrier is skipped by leaving early

    typedef int i4s __attribute__((ext_vector_type(4)));
    const i4s stq = reinterpret_cast<const i4s *>(strips)[strip];
    struct { int k0, k1, x0; } st = {stq.x, stq.y, stq.z};
    const T *img = src + (int64_t)blockIdx.z * sv.imageStride;
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + a.outBase;
    float *line = lds[wave];
    const int col = st.x0 + VEC * lane;
    const int colc = min(col, a.srcW - VEC);        // srcW >= VEC here (narrower images use the wide kernel)
    const int shift = col - colc;                    // 0 away from the right edge

    const int rowStart = interleave ? bY : bY * rowsPerBlock;
    const int rowStep = interleave ? gY : 1;
    const int rowEnd = interleave ? a.nB : min(rowStart + rowsPerBlock, a.nB);
    const int nOut = st.k1 - st.k0;
    // interleaved channels: taps `step` elements apart, dst element of lane entry ka split into (pixel, channel)
    const int step = CH ? a.tapStep : 1;
    auto taps = [&](const Win &c) { return CH ? (c.s1 - c.s0) / step : c.s1 - c.s0; };        // number of taps - 1
    auto out_off = [&](int ka) -> int64_t {
        return CH ? (int64_t)(ka / a.outChan) * a.outStrideA + ka % a.outChan : (int64_t)ka * a.outStrideA;
    };

    if (!CH && nOut <= 64 && (a.outStrideB == 1 || a.outStrideB == -1) && rowStep == 1) {
        // Quadrants 1 and 3 (pre-rotation by 90 / 270 degrees): the lane axis runs along dst y, so the output
        // rows of this workgroup are CONSECUTIVE DST COLUMNS of each lane's dst row.  Keep eight of them in
        // registers and write them as two 16-byte stores per lane instead of eight 4-byte stores a row pitch
        // apart.
        const bool live = lane < nOut;
        const Win c = load_win(laneTab, live ? st.k0 + lane : st.k0);
        const int off = c.s0 - st.x0, span = c.s1 - c.s0;
        float *orow = out + (int64_t)(st.k0 + lane) * a.outStrideA;
        for (int kb0 = rowStart; kb0 < rowEnd; kb0 += 8) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = 0.f;
                if (kb0 + j < rowEnd) {          // wave-uniform
                    const Win e = load_win(rowTab, kb0 + j);
                    const Cols v = vertical_pass<NT, T>(img, sv.rowStride, colc, e);
                    __builtin_amdgcn_wave_barrier();
                    park(line, lane, shift, v);
                    __builtin_amdgcn_wave_barrier();
                    acc[j] = horizontal_pass(line, off, span, c.wF, c.wM, c.wL);
                }
            }
            if (!live) continue;
            if (kb0 + 8 <= rowEnd) {
                // ascending dst columns: kb0..kb0+7 when outStrideB = +1, reversed when -1
                const bool fwd = a.outStrideB == 1;
                float *p = orow + (int64_t)(fwd ? kb0 : kb0 + 7) * a.outStrideB;
                f4 lo, hi;
                lo.x = fwd ? acc[0] : acc[7]; lo.y = fwd ? acc[1] : acc[6]; lo.z = fwd ? acc[2] : acc[5]; lo.w = fwd ? acc[3] : acc[4];
                hi.x = fwd ? acc[4] : acc[3]; hi.y = fwd ? acc[5] : acc[2]; hi.z = fwd ? acc[6] : acc[1]; hi.w = fwd ? acc[7] : acc[0];
                *reinterpret_cast<f4u *>(p) = lo;
                *reinterpret_cast<f4u *>(p + 4) = hi;
            } else if (kb0 + 4 == rowEnd) {
                // four dst columns (four output rows per workgroup): one 16-byte store
                const bool fwd = a.outStrideB == 1;
                float *p = orow + (int64_t)(fwd ? kb0 : kb0 + 3) * a.outStrideB;
                f4 v;
                v.x = fwd ? acc[0] : acc[3]; v.y = fwd ? acc[1] : acc[2]; v.z = fwd ? acc[2] : acc[1]; v.w = fwd ? acc[3] : acc[0];
                *reinterpret_cast<f4u *>(p) = v;
            } else if (kb0 + 2 == rowEnd) {
                typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                const bool fwd = a.outStrideB == 1;
                float *p = orow + (int64_t)(fwd ? kb0 : kb0 + 1) * a.outStrideB;
                f2u v;
                v.x = fwd ? acc[0] : acc[1]; v.y = fwd ? acc[1] : acc[0];
                *reinterpret_cast<f2u *>(p) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (kb0 + j < rowEnd) orow[(int64_t)(kb0 + j) * a.outStrideB] = acc[j];
            }
        }
    } else if (!CH && nOut <= 256 && (a.outStrideB == 1 || a.outStrideB == -1) && rowStep == 1) {
        // Quadrants 1 and 3 at ratios below 4 (65..256 outputs per strip): up to four outputs per lane, interleaved
        // across the wave (k = k0 + lane + 64 q, so that neighbouring lanes read neighbouring windows of the LDS line),
        // each holding FOUR consecutive dst columns in registers = one 16-byte store per output and chunk instead of
        // four 4-byte stores a dst row pitch apart.
        const int kl = st.k0 + lane;
        const int nq = (nOut + 63) >> 6;                         // wave-uniform: 2..4
        const Win c0 = load_win(laneTab, kl < st.k1 ? kl : st.k0), c1 = load_win(laneTab, kl + 64 < st.k1 ? kl + 64 : st.k0);
        const Win c2 = load_win(laneTab, kl + 128 < st.k1 ? kl + 128 : st.k0), c3 = load_win(laneTab, kl + 192 < st.k1 ? kl + 192 : st.k0);
        const bool fwd = a.outStrideB == 1;
        for (int kb0 = rowStart; kb0 < rowEnd; kb0 += 4) {
            float acc[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q][j] = 0.f;
                if (kb0 + j < rowEnd) {          // wave-uniform
                    const Win e = load_win(rowTab, kb0 + j);
                    const Cols v = vertical_pass<NT, T>(img, sv.rowStride, colc, e);
                    __builtin_amdgcn_wave_barrier();
                    park(line, lane, shift, v);
                    __builtin_amdgcn_wave_barrier();
                    acc[0][j] = horizontal_pass(line, c0.s0 - st.x0, c0.s1 - c0.s0, c0.wF, c0.wM, c0.wL);
                    acc[1][j] = horizontal_pass(line, c1.s0 - st.x0, c1.s1 - c1.s0, c1.wF, c1.wM, c1.wL);
                    if (nq > 2) acc[2][j] = horizontal_pass(line, c2.s0 - st.x0, c2.s1 - c2.s0, c2.wF, c2.wM, c2.wL);
                    if (nq > 3) acc[3][j] = horizontal_pass(line, c3.s0 - st.x0, c3.s1 - c3.s0, c3.wF, c3.wM, c3.wL);
                }
            }
            const bool full = kb0 + 4 <= rowEnd;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kl + 64 * q;
                if (k >= st.k1) continue;
                float *orow = out + (int64_t)k * a.outStrideA;
                if (full) {
                    // ascending dst columns: kb0..kb0+3 when outStrideB = +1, reversed when -1
                    f4 v;
                    v.x = fwd ? acc[q][0] : acc[q][3]; v.y = fwd ? acc[q][1] : acc[q][2];
                    v.z = fwd ? acc[q][2] : acc[q][1]; v.w = fwd ? acc[q][3] : acc[q][0];
                    *reinterpret_cast<f4u *>(orow + (int64_t)(fwd ? kb0 : kb0 + 3) * a.outStrideB) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kb0 + j < rowEnd) orow[(int64_t)(kb0 + j) * a.outStrideB] = acc[q][j];
                }
            }
        }
    } else if (nOut <= 64) {
        // Common case (any down-sampling ratio >= 4 source columns per output): one output per lane,
        // its window description stays in registers for all rows.
        const bool live = lane < nOut;
        const Win c = load_win(laneTab, live ? st.k0 + lane : st.k0);
        const int off = c.s0 - st.x0, span = taps(c);
        const int64_t outCol = out_off(st.k0 + lane);
        for (int kb = rowStart; kb < rowEnd; kb += rowStep) {
            const Win e = load_win(rowTab, kb);
            const Cols v = vertical_pass<NT, T>(img, sv.rowStride, colc, e);
            __builtin_amdgcn_wave_barrier();
            park(line, lane, shift, v);
            __builtin_amdgcn_wave_barrier();
            if (live) out[outCol + (int64_t)kb * a.outStrideB] = horizontal_pass(line, off, span, c.wF, c.wM, c.wL, step);
        }
    } else if (nOut <= 256 && a.outStrideA == 1) {
        // Ratios between 1 and 4 (65..256 outputs per strip) with dst x along the lanes: four consecutive
        // outputs per lane, window descriptions in registers, one 16-byte store per lane and output row.
        const int kq = st.k0 + 4 * lane;
        const int nq = min(max(st.k1 - kq, 0), 4);                   // how many of this lane's four outputs exist
        const Win c0 = load_win(laneTab, nq > 0 ? kq : st.k0), c1 = load_win(laneTab, nq > 1 ? kq + 1 : st.k0);
        const Win c2 = load_win(laneTab, nq > 2 ? kq + 2 : st.k0), c3 = load_win(laneTab, nq > 3 ? kq + 3 : st.k0);
        for (int kb = rowStart; kb < rowEnd; kb += rowStep) {
            const Win e = load_win(rowTab, kb);
            const Cols v = vertical_pass<NT, T>(img, sv.rowStride, colc, e);
            __builtin_amdgcn_wave_barrier();
            park(line, lane, shift, v);
            __builtin_amdgcn_wave_barrier();
            f4 r;
            r.x = horizontal_pass(line, c0.s0 - st.x0, taps(c0), c0.wF, c0.wM, c0.wL, step);
            r.y = horizontal_pass(line, c1.s0 - st.x0, taps(c1), c1.wF, c1.wM, c1.wL, step);
            r.z = horizontal_pass(line, c2.s0 - st.x0, taps(c2), c2.wF, c2.wM, c2.wL, step);
            r.w = horizontal_pass(line, c3.s0 - st.x0, taps(c3), c3.wF, c3.wM, c3.wL, step);
            float *o = out + kq + (int64_t)kb * a.outStrideB;           // outStrideA == 1: lane order = dst element order
            if (nq == 4) *reinterpret_cast<f4u *>(o) = r;
            else {
                if (nq > 0) o[0] = r.x;
                if (nq > 1) o[1] = r.y;
                if (nq > 2) o[2] = r.z;
            }
        }
    } else {
        // Many outputs per strip (up-sampling, transposed quadrants at small ratios): lanes walk the outputs.
        for (int kb = rowStart; kb < rowEnd; kb += rowStep) {
            const Win e = load_win(rowTab, kb);
            const Cols v = vertical_pass<NT, T>(img, sv.rowStride, colc, e);
            __builtin_amdgcn_wave_barrier();
            park(line, lane, shift, v);
            __builtin_amdgcn_wave_barrier();
            for (int k = st.k0 + lane; k < st.k1; k += 64) {
                const Win c = load_win(laneTab, k);
                out[out_off(k) + (int64_t)kb * a.outStrideB] =
                    horizontal_pass(line, c.s0 - st.x0, taps(c), c.wF, c.wM, c.wL, step);
            }
        }
    }
}

// Quadrants 1 and 3 (rotation by 90 / 270 degrees) at ratios below 2, i.e. more than 128 outputs per strip: the lane axis
// runs along dst y, so the outputs a wave produces for one output row kb are one dst COLUMN -- 64+ stores a dst row
// pitch apart.  Measured (profiles/r01_axis_transposed.txt): even 16-byte stores per lane leave 1:1 at 1.7 TB/s,
// whatever the grid order, because every store instruction writes into 64 different lines.  Here a wave gathers
// kTileCols consecutive output rows (= dst columns) of its strip in an LDS tile and then stores the tile with lanes
// running along dst x: 64 contiguous bytes per dst row and store.
constexpr int kTileCols = 16;
constexpr int kTilePitch = kTileCols + 1;              // odd pitch: the transposed reads hit 64 different banks
constexpr int kTileWaveFloats = 64 * VEC + 8 + 256 * kTilePitch;

// The output rows (columns of the tile) of one wave in a software pipeline kDepth deep: with 72 KiB of LDS per workgroup a SIMD
// holds two waves, and a wave that waits for each output row's source rows before it asks for the next spends its life in load
// latency (1:1 at 270 degrees: 2.6 TB/s so, 3.3 with the pipeline).  NR = source rows a window may have; taller ones: the plain path.
template <bool NT, typename T, int NR, int kDepth, typename Finish>
__device__ __forceinline__ void tile_columns(const T *__restrict__ img, int64_t rowStride, int colc, const AxisEntry *__restrict__ rowTab, int kb0, int nCols,
                                             Finish &&finish_column)
{
    // (the trailing waves of the last cooperative workgroup have no output rows: row kb0 then lies past the table's end and must
    // not even be looked at; their caller still meets the other waves at its barrier)
    if (nCols <= 0) return;
    RowsN<T, NR> ring[kDepth];
    Win wring[kDepth];
    bool tall = false;
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
        wring[d] = load_win(rowTab, kb0 + (d < nCols ? d : 0));
        tall = tall || wring[d].s1 - wring[d].s0 >= NR;
    }
    for (int j = kDepth; j < nCols; ++j) { const Win e = load_win(rowTab, kb0 + j); tall = tall || e.s1 - e.s0 >= NR; }
    if (tall) {
        for (int j = 0; j < nCols; ++j) finish_column(j, vertical_pass<NT, T>(img, rowStride, colc, load_win(rowTab, kb0 + j)));
        return;
    }
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if (d < nCols) issue_rows<NT, T, NR>(img, rowStride, colc, wring[d], ring[d]);
    for (int j0 = 0; j0 < nCols; j0 += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int j = j0 + d;
            if (j < nCols) {                                  // wave-uniform
                const Cols v = finish_rows<T, NR>(ring[d], wring[d]);
                if (j + kDepth < nCols) {                     // the slot's next output row: asked for before this one is parked
                    wring[d] = load_win(rowTab, kb0 + j + kDepth);
                    issue_rows<NT, T, NR>(img, rowStride, colc, wring[d], ring[d]);
                }
                finish_column(j, v);
            }
        }
    }
}

// COOP: the four waves of a workgroup take ONE strip and four consecutive groups of 16 output rows, meet at a barrier and store
// together -- lane = one of 64 consecutive dst x, read from the four waves' tiles: 256 contiguous bytes per dst row and store
// instruction instead of four 64-byte segments.
constexpr int kTileWaveFloatsCoop = (kTileWaveFloats + 31) / 32 * 32 + 16;      // tiles 16 banks apart: the joint reads are conflict free
template <bool NT, typename T, bool COOP>
__global__ __launch_bounds__(kWaves * 64) void aai_axis_tile_kernel(AxisLaunch a, const AxisEntry *__restrict__ laneTab,
                                                                     const AxisEntry *__restrict__ rowTab, const AxisStrip *__restrict__ strips,
                                                                     const T *__restrict__ src, ImageView sv,
                                                                     float *__restrict__ dst, ImageView dv)
{
    constexpr int kWaveFloats = COOP ? kTileWaveFloatsCoop : kTileWaveFloats;
    __shared__ __attribute__((aligned(16))) float smem[kWaves * kWaveFloats];     // 72 KiB: two workgroups per CU

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = COOP ? (int)blockIdx.x : (int)blockIdx.x * kWaves + wave;
    if (strip >= a.nStrips) return;   // (not COOP: waves are independent, no s_barrier anywhere; COOP: the whole workgroup leaves)

    typedef int i4s __attribute__((ext_vector_type(4)));
    const i4s stq = reinterpret_cast<const i4s *>(strips)[strip];
    struct { int k0, k1, x0; } st = {stq.x, stq.y, stq.z};
    const T *img = src + (int64_t)blockIdx.z * sv.imageStride;
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + a.outBase;
    float *line = smem + wave * kWaveFloats;
    float *tile = line + 64 * VEC + 8;
    const int col = st.x0 + VEC * lane;
    const int colc = min(col, a.srcW - VEC);
    const int shift = col - colc;
    const int nOut = st.k1 - st.k0;                              // <= 256
    const int nq = (nOut + 63) >> 6;                             // wave-uniform
    const int kb0 = COOP ? ((int)blockIdx.y * kWaves + wave) * kTileCols : (int)blockIdx.y * kTileCols;
    const int nCols = max(0, min(kTileCols, a.nB - kb0));

    const int kl = st.k0 + lane;
    const Win c0 = load_win(laneTab, kl < st.k1 ? kl : st.k0), c1 = load_win(laneTab, kl + 64 < st.k1 ? kl + 64 : st.k0);
    const Win c2 = load_win(laneTab, kl + 128 < st.k1 ? kl + 128 : st.k0), c3 = load_win(laneTab, kl + 192 < st.k1 ? kl + 192 : st.k0);
    // The output rows of a tile in a software pipeline kDepth deep: with 72 KiB of LDS per workgroup a SIMD holds two waves, and a wave
    // that waits for each output row's source rows before it asks for the next spends its life in load latency (1:1 at 270 degrees:
    // 2.6 TB/s).  Windows here are at most four source rows tall (ratios below 2); a taller one takes the plain path.
    auto finish_column = [&](int j, const Cols &v) {
        __builtin_amdgcn_wave_barrier();
        park(line, lane, shift, v);
        __builtin_amdgcn_wave_barrier();
        // outputs interleaved across the wave (k = k0 + lane + 64 q): neighbouring lanes read neighbouring windows
        tile[lane * kTilePitch + j] = horizontal_pass(line, c0.s0 - st.x0, c0.s1 - c0.s0, c0.wF, c0.wM, c0.wL);
        if (nq > 1) tile[(lane + 64) * kTilePitch + j] = horizontal_pass(line, c1.s0 - st.x0, c1.s1 - c1.s0, c1.wF, c1.wM, c1.wL);
        if (nq > 2) tile[(lane + 128) * kTilePitch + j] = horizontal_pass(line, c2.s0 - st.x0, c2.s1 - c2.s0, c2.wF, c2.wM, c2.wL);
        if (nq > 3) tile[(lane + 192) * kTilePitch + j] = horizontal_pass(line, c3.s0 - st.x0, c3.s1 - c3.s0, c3.wF, c3.wM, c3.wL);
    };
    // windows of at most four source rows (ratios up to 3): eight output rows in flight; up to eight rows (ratios up to 4, where a
    // strip still holds more than 64 outputs): four; anything taller takes the plain path
    if (a.maxRowSpan <= 4) tile_columns<NT, T, 4, 8>(img, sv.rowStride, colc, rowTab, kb0, nCols, finish_column);
    else tile_columns<NT, T, 8, 4>(img, sv.rowStride, colc, rowTab, kb0, nCols, finish_column);
    if (COOP) {
        __syncthreads();
        // lane = dst x within the workgroup's 64 output rows (kb), read from the tile of the wave that computed it; wave w stores
        // the dst rows r = w, w + 4, ...
        const int kbBase = (int)blockIdx.y * kWaves * kTileCols;
        const float *from = smem + (lane >> 4) * kWaveFloats + 64 * VEC + 8 + (lane & (kTileCols - 1));
        if (kbBase + lane < a.nB)
            for (int r = wave; r < nOut; r += kWaves)
                out[(int64_t)(st.k0 + r) * a.outStrideA + (int64_t)(kbBase + lane) * a.outStrideB] = from[r * kTilePitch];
        return;
    }
    __builtin_amdgcn_wave_barrier();
    // store: four dst rows x 16 dst columns per instruction
    const int jr = lane & (kTileCols - 1), rr = lane >> 4;
    if (jr < nCols)
        for (int r = rr; r < nOut; r += 64 / kTileCols)
            out[(int64_t)(st.k0 + r) * a.outStrideA + (int64_t)(kb0 + jr) * a.outStrideB] = tile[r * kTilePitch + jr];
}

// Fallback for footprints wider than one strip (down-sampling by more than ~250:1): one thread per output
// pixel walks its whole window.  Correct, not fast; such ratios leave almost no output to write.
template <typename T>
__global__ __launch_bounds__(256) void aai_axis_wide_kernel(AxisLaunch a, const T *__restrict__ src, ImageView sv,
                                                             float *__restrict__ dst, ImageView dv, int kb0)
{
    const int ka = blockIdx.x * 64 + (threadIdx.x & 63);
    const int kb = kb0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ka >= a.nA || kb >= a.nB) return;
    const Win c = load_win(a.laneTab, ka), e = load_win(a.rowTab, kb);
    const T *img = src + (int64_t)blockIdx.z * sv.imageStride;
    const int step = a.tapStep > 1 ? a.tapStep : 1;              // interleaved channels: taps `step` elements apart
    const int oc = a.outChan > 1 ? a.outChan : 1;
    float acc = 0.f;
    for (int y = e.s0; y <= e.s1; ++y) {
        const T *row = img + (int64_t)y * sv.rowStride;
        float h = 0.f;
        for (int x = c.s0; x <= c.s1; x += step)
            h += row_weight(c, x) * (float)row[x];
        acc += row_weight(e, y) * h;
    }
    dst[(int64_t)blockIdx.z * dv.imageStride + a.outBase + (int64_t)(ka / oc) * a.outStrideA + ka % oc + (int64_t)kb * a.outStrideB] = acc;
}

}  // namespace

// Launch-shape overrides for experiments (tools/tune_axis.py), in the experiments build only (make exp, -DAAI_EXPERIMENTS):
// AAI_AXIS_TUNE="nt=1,rows=1,interleave=0,gy=0,swap=0,tile=1" is read once per process; aai_debug_axis_tune() replaces it at run time.
struct AxisTune { int nt = -1, rows = 0, interleave = -1, gy = -1, swap = -1, tile = -1; };
static AxisTune parse_axis_tune(const char *spec)
{
    AxisTune t;
    if (!spec) return t;
    auto get = [&](const char *key, int &v) {
        const char *p = strstr(spec, key);
        if (p) v = atoi(p + strlen(key));
    };
    get("nt=", t.nt); get("rows=", t.rows); get("interleave=", t.interleave); get("gy=", t.gy); get("swap=", t.swap); get("tile=", t.tile);
    return t;
}
static AxisTune &axis_tune()
{
    static AxisTune t = parse_axis_tune(experiment_env("AAI_AXIS_TUNE"));
    return t;
}
void set_axis_tune(const char *spec) { axis_tune() = parse_axis_tune(spec); }

template <typename T>
static hipError_t launch_axis_typed(const AxisLaunch &a, const T *src, ImageView sv, float *dst, ImageView dv,
                                    int batch, hipStream_t stream, const char **kernelName);

hipError_t launch_axis(const AxisLaunch &a, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, hipStream_t stream, const char **kernelName)
{
    switch (srcType) {
    case SRC_U8: return launch_axis_typed(a, static_cast<const unsigned char *>(src), sv, dst, dv, batch, stream, kernelName);
    case SRC_U16: return launch_axis_typed(a, static_cast<const unsigned short *>(src), sv, dst, dv, batch, stream, kernelName);
    default: return launch_axis_typed(a, static_cast<const float *>(src), sv, dst, dv, batch, stream, kernelName);
    }
}

template <typename T>
static hipError_t launch_axis_typed(const AxisLaunch &a, const T *src, ImageView sv, float *dst, ImageView dv,
                                    int batch, hipStream_t stream, const char **kernelName)
{
    if (a.nA <= 0 || a.nB <= 0 || batch <= 0) return hipSuccess;
    if (a.wide) {
        if (kernelName) *kernelName = "aai_axis_wide_kernel";
        for (int kb0 = 0; kb0 < a.nB; kb0 += 65535 * 4) {          // grid.y carries at most 65535 blocks of 4 rows
            dim3 grid((a.nA + 63) / 64, (std::min(a.nB - kb0, 65535 * 4) + 3) / 4, batch);
            hipLaunchKernelGGL((aai_axis_wide_kernel<T>), grid, dim3(256), 0, stream, a, src, sv, dst, dv, kb0);
        }
        return hipGetLastError();
    }
    // Launch shape (see the kernel comment for the measurements behind the defaults).  Each knob can be
    // overridden for experiments: AAI_AXIS_TUNE="nt=1,rows=1,interleave=0,gy=0".
    // Defaults from the sweeps in profiles/r01_axis_tune_*.txt (8192^2 sources):
    //   * windows that share source rows between output rows (grid not pixel-aligned) want cached loads and
    //     two rows per workgroup (5.8 vs 5.2 TB/s at 4:1); disjoint windows want nontemporal loads;
    //   * >= 4 source rows per output row: one output row per workgroup (6.8 TB/s at 4:1; 5.3 at 8 rows);
    //   * 2-3 source rows: four (5.0-5.2 TB/s vs 3.7-4.6 at one or two);
    //   * ratios below 2 (129..256 outputs per strip, write-heavy): 32 rows per workgroup (4.2 TB/s at 1:1
    //     vs 2.5 at four and 1.4 at one); the up-sampling path (more than 256 outputs per strip) prefers 4;
    //   * 8- and 16-bit sources move 4x / 2x fewer bytes and are bound by per-wave latency, not by HBM: eight / four
    //     rows per workgroup (8192^2 u8 -> 2048^2: 28 us at one row, 21 at eight; profiles/r01_typed_sources.txt).
    int nt = a.rowsShared ? 0 : 1, interleave = 0, gy = 0, swapXY = 0;
    int rows = a.maxRowSpan >= 4 ? (a.rowsShared ? 2 : 1) : 4;
    if (sizeof(T) < 4) {
        // ... as long as that leaves a few workgroups per CU
        const int want = sizeof(T) == 1 ? 8 : 4;
        const int64_t stripBlocks = (int64_t)((a.nStrips + kWaves - 1) / kWaves) * batch;
        while (rows < want && stripBlocks * ((a.nB + 2 * rows - 1) / (2 * rows)) >= 2048) rows *= 2;
    }
    if (a.transposed) {
        // transposed quadrants: 2, 4 or 8 dst columns per lane and store; about 16 source rows per workgroup is the
        // sweet spot between DRAM page locality and store width (8192^2 at 90 degrees: 4:1 5.9 TB/s at 4 rows vs 5.2 at
        // 8; 8:1 6.0 at 2 vs 4.8 at 8; profiles/r01_axis_transposed.txt)
        rows = a.maxRowSpan >= 8 ? 2 : (a.maxRowSpan >= 3 ? 4 : 8);
        if (a.maxOutputsPerStrip > 64) rows = a.maxRowSpan >= 2 ? 4 : 16;     // the four-column path (ratios below 4)
    }
    if (!a.transposed && a.maxOutputsPerStrip > 128 && a.maxOutputsPerStrip <= 256) rows = 32;      // ratios below 2: write-heavy
    // the plan's measured choice for this (geometry, device) -- fp32 sources only (aai_engine.cpp: tune_axis_plan)
    if (sizeof(T) == 4 && a.tuneRows > 0) { rows = a.tuneRows; nt = a.tuneNt; swapXY = a.tuneSwap; }
    const AxisTune &tune = axis_tune();
    if (tune.nt >= 0) nt = tune.nt;
    if (tune.rows > 0) rows = tune.rows;
    if (tune.interleave >= 0) interleave = tune.interleave;
    if (tune.gy >= 0) gy = tune.gy;
    if (tune.swap >= 0) swapXY = tune.swap;
    // (measured, profiles/r01_axis_transposed.txt: wins below 2:1 -- 1:1 1.8 -> 2.3 TB/s, x4 up-sampling 1.2 -> 1.6 --
    // and loses 5-14 % to the four-column register path between 2:1 and 4:1)
    // ... and, since its output rows run in a software pipeline and its waves store together, down to 64 outputs per strip (ratios up
    // to 4) wherever no output row needs more than the pipeline's eight source rows: 3:1 at 270 degrees 300 -> 242 us per 4 images,
    // 2.5:1 351 -> 271, 2:1 285 -> 275 (profiles/r03_axis_tile.txt)
    const bool tileable = a.transposed && a.tapStep <= 1 && (a.outStrideB == 1 || a.outStrideB == -1) && a.maxOutputsPerStrip <= 256;
    int tile = tileable && (a.maxOutputsPerStrip > 128 || (a.maxOutputsPerStrip > 64 && a.maxRowSpan <= 8));
    if (tune.tile == 0) tile = 0;
    if (tune.tile == 1) tile = tileable && a.maxOutputsPerStrip > 128;      // experiments: the round-1 rule and the independent-waves form
    if (tile && (a.nB + kTileCols - 1) / kTileCols <= 65535) {
        // transposed quadrants at ratios below 4: LDS tile, stores along dst x (see aai_axis_tile_kernel)
        if (kernelName) *kernelName = "aai_axis_tile_kernel";
        const bool coop = tune.tile != 1 && a.nStrips <= 2147483647;      // (experiments: tile=1 keeps the independent-waves form)
        if (coop) {
            dim3 grid(a.nStrips, (a.nB + kWaves * kTileCols - 1) / (kWaves * kTileCols), batch), block(kWaves * 64);
            if (nt) hipLaunchKernelGGL((aai_axis_tile_kernel<true, T, true>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv);
            else hipLaunchKernelGGL((aai_axis_tile_kernel<false, T, true>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv);
            return hipGetLastError();
        }
        dim3 grid((a.nStrips + kWaves - 1) / kWaves, (a.nB + kTileCols - 1) / kTileCols, batch), block(kWaves * 64);
        if (nt) hipLaunchKernelGGL((aai_axis_tile_kernel<true, T, false>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv);
        else hipLaunchKernelGGL((aai_axis_tile_kernel<false, T, false>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv);
        return hipGetLastError();
    }
    const int blocksX = (a.nStrips + kWaves - 1) / kWaves;
    int blocksY = (a.nB + rows - 1) / rows;
    if (interleave && gy > 0) blocksY = gy < a.nB ? gy : a.nB;
    while (blocksY > 65535) { rows *= 2; blocksY = (a.nB + rows - 1) / rows; }
    if (swapXY && blocksX > 65535) swapXY = 0;
    dim3 grid(swapXY ? blocksY : blocksX, swapXY ? blocksX : blocksY, batch), block(kWaves * 64);
    if (kernelName) *kernelName = "aai_axis_kernel";
    if (a.tapStep > 1) {
        if (nt) hipLaunchKernelGGL((aai_axis_kernel<true, T, true>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv, rows, interleave, swapXY);
        else hipLaunchKernelGGL((aai_axis_kernel<false, T, true>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv, rows, interleave, swapXY);
    } else {
        if (nt) hipLaunchKernelGGL((aai_axis_kernel<true, T, false>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv, rows, interleave, swapXY);
        else hipLaunchKernelGGL((aai_axis_kernel<false, T, false>), grid, block, 0, stream, a, a.laneTab, a.rowTab, a.strips, src, sv, dst, dv, rows, interleave, swapXY);
    }
    return hipGetLastError();
}

}  // namespace aai
