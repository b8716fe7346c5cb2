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

namespace aai {

namespace {

constexpr int kWaves = 4;
constexpr int kLdsLine = STRIP_COLS + 8;   // floats per wave; +8 keeps lines 16-byte aligned and apart

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access
typedef float f4 __attribute__((ext_vector_type(4)));

// One 16-byte load of source columns [col, col+4) of a row, zero-filled past the image edge.
__device__ __forceinline__ f4 load_cols(const float *__restrict__ row, int col, int W)
{
    if (col + 3 < W) return *reinterpret_cast<const f4u *>(row + col);
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < W) v.x = row[col];
    if (col + 1 < W) v.y = row[col + 1];
    if (col + 2 < W) v.z = row[col + 2];
    return v;
}

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

// Vertical pass for one output row: sum_y w(y) * src[y][col..col+3], rows issued four at a time.
__device__ __forceinline__ f4 vertical_pass(const float *__restrict__ img, int64_t rowStride, int col, int W,
                                            const Win e)
{
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int y = e.s0; y <= e.s1; y += 4) {
        f4 r0, r1 = {0.f, 0.f, 0.f, 0.f}, r2 = r1, r3 = r1;
        const float *p = img + (int64_t)y * rowStride;
        // e is wave-uniform, so these branches are scalar; the taken loads issue back to back.
        r0 = load_cols(p, col, W);
        if (y + 1 <= e.s1) r1 = load_cols(p + rowStride, col, W);
        if (y + 2 <= e.s1) r2 = load_cols(p + 2 * rowStride, col, W);
        if (y + 3 <= e.s1) r3 = load_cols(p + 3 * rowStride, col, W);
        const float w0 = row_weight(e, y), w1 = row_weight(e, y + 1), w2 = row_weight(e, y + 2), w3 = row_weight(e, y + 3);
        acc += w0 * r0;
        acc += w1 * r1;
        acc += w2 * r2;
        acc += w3 * r3;
    }
    return acc;
}

// Horizontal pass for one output pixel from the wave's LDS line (indices relative to the strip origin).
__device__ __forceinline__ float horizontal_pass(const float *line, int off, int span, float wF, float wM, float wL)
{
    float s = wF * line[off];
    if (span > 0) {
        float mid = 0.f;
        for (int i = 1; i < span; ++i) mid += line[off + i];
        s += wM * mid + wL * line[off + span];
    }
    return s;
}

__global__ __launch_bounds__(kWaves * 64) void aai_axis_kernel(AxisLaunch a, const float *__restrict__ src, ImageView sv,
                                                                float *__restrict__ dst, ImageView dv, int rowsPerBlock)
{
    __shared__ __attribute__((aligned(16))) float lds[kWaves][kLdsLine];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int strip = blockIdx.x * kWaves + wave;
    if (strip >= a.nStrips) return;   // waves are independent: no barrier is skipped by leaving early

    typedef int i4s __attribute__((ext_vector_type(4)));
    const i4s stq = reinterpret_cast<const i4s *>(a.strips)[strip];
    struct { int k0, k1, x0; } st = {stq.x, stq.y, stq.z};
    const float *img = src + (int64_t)blockIdx.z * sv.imageStride;
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + a.outBase;
    float *line = lds[wave];
    const int col = st.x0 + 4 * lane;

    const int kb0 = blockIdx.y * rowsPerBlock;
    const int kb1 = min(kb0 + rowsPerBlock, a.nB);
    const int nOut = st.k1 - st.k0;

    if (nOut <= 64) {
        // Common case (any down-sampling ratio >= 4 source columns per output): one output per lane,
        // its window description stays in registers for all rows.
        const bool live = lane < nOut;
        const Win c = load_win(a.laneTab, live ? st.k0 + lane : st.k0);
        const int off = c.s0 - st.x0, span = c.s1 - c.s0;
        const int64_t outCol = (int64_t)(st.k0 + lane) * a.outStrideA;
        for (int kb = kb0; kb < kb1; ++kb) {
            const Win e = load_win(a.rowTab, kb);
            const f4 v = vertical_pass(img, sv.rowStride, col, a.srcW, e);
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<f4 *>(line + 4 * lane) = v;
            __builtin_amdgcn_wave_barrier();
            if (live) out[outCol + (int64_t)kb * a.outStrideB] = horizontal_pass(line, off, span, c.wF, c.wM, c.wL);
        }
    } else {
        // Many outputs per strip (up-sampling, or ratios below 4): lanes walk the strip's outputs.
        for (int kb = kb0; kb < kb1; ++kb) {
            const Win e = load_win(a.rowTab, kb);
            const f4 v = vertical_pass(img, sv.rowStride, col, a.srcW, e);
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<f4 *>(line + 4 * lane) = v;
            __builtin_amdgcn_wave_barrier();
            for (int k = st.k0 + lane; k < st.k1; k += 64) {
                const Win c = load_win(a.laneTab, k);
                out[(int64_t)k * a.outStrideA + (int64_t)kb * a.outStrideB] =
                    horizontal_pass(line, c.s0 - st.x0, c.s1 - c.s0, c.wF, c.wM, c.wL);
            }
        }
    }
}

// Fallback for footprints wider than one strip (down-sampling by more than ~250:1): one thread per output
// pixel walks its whole window.  Correct, not fast; such ratios leave almost no output to write.
__global__ __launch_bounds__(256) void aai_axis_wide_kernel(AxisLaunch a, const float *__restrict__ src, ImageView sv,
                                                             float *__restrict__ dst, ImageView dv)
{
    const int ka = blockIdx.x * 64 + (threadIdx.x & 63);
    const int kb = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ka >= a.nA || kb >= a.nB) return;
    const Win c = load_win(a.laneTab, ka), e = load_win(a.rowTab, kb);
    const float *img = src + (int64_t)blockIdx.z * sv.imageStride;
    float acc = 0.f;
    for (int y = e.s0; y <= e.s1; ++y) {
        const float *row = img + (int64_t)y * sv.rowStride;
        float h = 0.f;
        for (int x = c.s0; x <= c.s1; ++x)
            h += row_weight(c, x) * row[x];
        acc += row_weight(e, y) * h;
    }
    dst[(int64_t)blockIdx.z * dv.imageStride + a.outBase + (int64_t)ka * a.outStrideA + (int64_t)kb * a.outStrideB] = acc;
}

}  // namespace

hipError_t launch_axis(const AxisLaunch &a, const float *src, ImageView sv, float *dst, ImageView dv,
                       int batch, hipStream_t stream, const char **kernelName)
{
    if (a.nA <= 0 || a.nB <= 0 || batch <= 0) return hipSuccess;
    if (a.wide) {
        dim3 grid((a.nA + 63) / 64, (a.nB + 3) / 4, batch);
        if (kernelName) *kernelName = "aai_axis_wide_kernel";
        hipLaunchKernelGGL(aai_axis_wide_kernel, grid, dim3(256), 0, stream, a, src, sv, dst, dv);
        return hipGetLastError();
    }
    // Rows per workgroup: enough waves to fill 256 CUs several times over, few enough rows that the
    // tail wave-round stays short.
    const int blocksX = (a.nStrips + kWaves - 1) / kWaves;
    int rowsPerBlock = 8;
    while (rowsPerBlock > 1 && (int64_t)blocksX * ((a.nB + rowsPerBlock - 1) / rowsPerBlock) * batch < 4096) rowsPerBlock >>= 1;
    dim3 grid(blocksX, (a.nB + rowsPerBlock - 1) / rowsPerBlock, batch);
    if (kernelName) *kernelName = "aai_axis_kernel";
    hipLaunchKernelGGL(aai_axis_kernel, grid, dim3(kWaves * 64), 0, stream, a, src, sv, dst, dv, rowsPerBlock);
    return hipGetLastError();
}

}  // namespace aai
