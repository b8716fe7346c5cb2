// aai_rotated_wide.hip -- K2 for WIDE footprints (dst pixels of more than ~5.5 source pixels a side at a general rotation):
// the fp32 quad formulation (aai_rot_quad.hpp) over a window of up to 32 x 32 source pixels, split into 2 x 2 or 4 x 4
// parts of at most 8 x 8 positions, ONE LANE PER PART.
//
// Replaces Source.cpp:413-579 + 986-1431 of the reference for these geometries, which the double-precision runs kernel
// (aai_rotated_runs_kernel) served before: it read the source at a tenth of the HBM rate because every one of the ~5 L
// boundary pairs of a dst pixel cost ~200 double-precision instructions in ONE lane, while a dst image of (W / L)^2 pixels has
// too few lanes to fill the chip (8 : 1 on 8192^2: 16 k waves).  Here a dst pixel is 4 or 16 lanes -- four / sixteen times the
// waves, each part a window the existing passes classify and evaluate in fp32 (an interior part is 36 ... 64 pixels of
// area 1) -- and the partial sums meet in a butterfly of lane exchanges (quad_parts_sum spells the order out for the CPU replay).
//
// Decisions: as for the other fp32 kernels the plan's scan (aai_wide_scan_kernel: the same code and lane layout, no pixel
// loads) flags the dst pixels with a decision too close to its threshold, the production kernel skips them and the
// double-precision fix-up pass computes them beside it.
//
// Fast mode (the reference's default mode) takes the same decomposition with memberships instead of areas: aai_wide_fast_kernel.
//
// Plain single-channel images without replication (scale 1: wide footprints never have any); interleaved channels and
// the double-precision policy keep the double-precision kernels.
#include "aai_kernels.hpp"
#include "aai_rot_quad.hpp"
#include "aai_quad_src.hpp"

namespace aai {

namespace {

constexpr int wide_waves_per_simd(int win, bool hp)
{
    const int w = 160 / (win * win) >= 8 ? 8 : 160 / (win * win);
    return hp && w > 2 ? w - 1 : w;
}

// lane -> (dst pixel, part).  A block of 256 lanes is 256 / PARTS^2 dst pixels of ONE 16-pixel-wide mask word region:
// PARTS = 2: the 16 x 4 pixels of lane-mask word `sub` (0..3) of tile (bx, blockIdx.y); PARTS = 4: row `sub` (0..15) of the tile
template <int PARTS>
struct WideLane {
    int dx, dy, partI, partJ;
    size_t word;
    int bit;
    __device__ __forceinline__ WideLane(const RotLaunch &r, int tileRow0)
    {
        constexpr int LANES = PARTS * PARTS;
        const int tid = threadIdx.x;
        const int tilesX = (r.dW + 15) / 16;
        const int sub = blockIdx.x / tilesX, bx = blockIdx.x - sub * tilesX;
        const int p = tid / LANES, part = tid & (LANES - 1);
        partI = part % PARTS; partJ = part / PARTS;
        const int tileRow = tileRow0 + blockIdx.y;
        if (PARTS == 2) {
            dx = bx * 16 + (p & 15);
            dy = tileRow * 16 + sub * 4 + (p >> 4);
            word = ((size_t)tileRow * tilesX + bx) * 4 + sub;
            bit = p;
        } else {
            dx = bx * 16 + p;
            dy = tileRow * 16 + sub;
            word = ((size_t)tileRow * tilesX + bx) * 4 + (sub >> 2);
            bit = (sub & 3) * 16 + p;
        }
    }
};

// the parts' sums of one dst pixel: every lane of the pixel ends with the total (quad_parts_sum's order)
template <int PARTS>
__device__ __forceinline__ float wide_total(float v)
{
#pragma unroll
    for (int o = 1; o < PARTS * PARTS; o <<= 1) v = v + __shfl_xor(v, o);
    return v;
}

template <typename T, int WIN, bool HP, int PARTS>
__global__ __launch_bounds__(kQuadBlock, wide_waves_per_simd(WIN, HP)) void aai_wide_kernel(RotLaunch r, QuadConsts<float> q, QuadMap m, const T *__restrict__ src,
                                                                                           ImageView sv, float *__restrict__ dst, ImageView dv,
                                                                                           const unsigned long long *__restrict__ skipMasks)
{
    __shared__ float window[WIN * WIN][kQuadBlock];
    const WideLane<PARTS> l(r, r.dyBase / 16);
    if (!(l.dx < r.dW && l.dy < r.dyEnd)) return;                     // (whole pixels: all lanes of a dst pixel decide alike)
    if (skipMasks && ((skipMasks[l.word] >> l.bit) & 1ull)) return;   // flagged: the double-precision pass's, beside this kernel
    double px, py;
    pixel_centre(r, l.dx, l.dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    float sumA = 0.f, sumVA[1] = {0.f};
    // a centre further than the window's reach from the lattice touches nothing (and stays inside int range)
    if (cx > -40.0 && cx < (double)r.mW + 40.0 && cy > -40.0 && cy < (double)r.mH + 40.0) {
        QuadSrc<T, WIN, false, true> s;
        s.img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
        s.m = &m; s.mW = r.mW; s.mH = r.mH; s.lds = window; s.tid = threadIdx.x;
        quad_pixel<float, WIN, false, HP, 1, QuadSrc<T, WIN, false, true>, true>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA, l.partI, l.partJ);
    }
    const float A = wide_total<PARTS>(sumA), VA = wide_total<PARTS>(sumVA[0]);
    if ((threadIdx.x & (PARTS * PARTS - 1)) == 0)
        dst[(int64_t)blockIdx.z * dv.imageStride + (int64_t)(l.dy - r.dyBase) * dv.rowStride + l.dx] = A > 0.f ? VA / A : 0.f;      // Source.cpp:577
}

// Once per geometry: the same lanes and arithmetic without pixel loads; a pixel one of whose parts has a decision within
// QuadConsts::margin of its threshold, or whose total area is too small for fp32 weights, gets its bit in the lane masks
template <int WIN, bool HP, int PARTS>
__global__ __launch_bounds__(kQuadBlock) void aai_wide_scan_kernel(RotLaunch r, QuadConsts<float> q, unsigned long long *__restrict__ laneMasks,
                                                                  unsigned *__restrict__ counter, int tileRow0)
{
    const WideLane<PARTS> l(r, tileRow0);
    if (!(l.dx < r.dW && l.dy < r.dH)) return;
    double px, py;
    pixel_centre(r, l.dx, l.dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    float sumA = 0.f, sumVA[1] = {0.f};
    bool uncertain = false;
    if (cx > -40.0 && cx < (double)r.mW + 40.0 && cy > -40.0 && cy < (double)r.mH + 40.0) {
        NoSrc s;
        uncertain = quad_pixel<float, WIN, true, HP, 1, NoSrc, true>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA, l.partI, l.partJ);
    }
    const float A = wide_total<PARTS>(sumA);
    if (A > 0.f && A < q.minArea) uncertain = true;
    if (uncertain) {
        const unsigned long long bit = 1ull << l.bit;
        const unsigned long long old = atomicOr(laneMasks + l.word, bit);
        if (!(old & bit)) atomicAdd(counter, 1u);
    }
}

// Fast mode over a wide footprint (Source.cpp:868-907 + 837-864): the mean of the pixels whose centres lie in the dst square, the window
// of centres (QuadConsts::winFastFull positions a side) in PARTS x PARTS parts, a lane per part as above; memberships are two
// compares per position on values held in registers (no LDS), sums and counts meet in the same butterfly.
template <int PARTS>
__device__ __forceinline__ int wide_total_int(int v)
{
#pragma unroll
    for (int o = 1; o < PARTS * PARTS; o <<= 1) v = v + __shfl_xor(v, o);
    return v;
}

template <typename T, int WIN, int PARTS>
__global__ __launch_bounds__(kQuadBlock) void aai_wide_fast_kernel(RotLaunch r, QuadConsts<float> q, QuadMap m, const T *__restrict__ src, ImageView sv,
                                                                  float *__restrict__ dst, ImageView dv, const unsigned long long *__restrict__ skipMasks)
{
    const WideLane<PARTS> l(r, r.dyBase / 16);
    if (!(l.dx < r.dW && l.dy < r.dyEnd)) return;
    if (skipMasks && ((skipMasks[l.word] >> l.bit) & 1ull)) return;
    double px, py;
    pixel_centre(r, l.dx, l.dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    float sum = 0.f;
    int count = 0;
    if (cx > -40.0 && cx < (double)r.mW + 40.0 && cy > -40.0 && cy < (double)r.mH + 40.0) {
        const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
        fast_window_sum<T, WIN, false>(r, q, m, img, (int)cx, (int)cy, px - cx, py - cy, threadIdx.x, l.partI, l.partJ, sum, count);
    }
    const float S = wide_total<PARTS>(sum);
    const int N = wide_total_int<PARTS>(count);
    if ((threadIdx.x & (PARTS * PARTS - 1)) == 0)
        dst[(int64_t)blockIdx.z * dv.imageStride + (int64_t)(l.dy - r.dyBase) * dv.rowStride + l.dx] = N > 0 ? S / (float)N : 0.f;      // Source.cpp:905
}

template <int WIN, int PARTS>
__global__ __launch_bounds__(kQuadBlock) void aai_wide_fast_scan_kernel(RotLaunch r, QuadConsts<float> q, unsigned long long *__restrict__ laneMasks,
                                                                       unsigned *__restrict__ counter, int tileRow0)
{
    const WideLane<PARTS> l(r, tileRow0);
    if (!(l.dx < r.dW && l.dy < r.dH)) return;
    double px, py;
    pixel_centre(r, l.dx, l.dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    bool uncertain = false;
    if (cx > -40.0 && cx < (double)r.mW + 40.0 && cy > -40.0 && cy < (double)r.mH + 40.0) {
        NoSrc s;
        float sum;
        int count;
        uncertain = quad_fast_pixel<float, WIN, true>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sum, count, l.partI, l.partJ);
    }
    if (uncertain) {
        const unsigned long long bit = 1ull << l.bit;
        const unsigned long long old = atomicOr(laneMasks + l.word, bit);
        if (!(old & bit)) atomicAdd(counter, 1u);
    }
}

// FAMILIES: 1 = area mode, 2 = fast mode (which kernels this translation unit holds: "translation units" below)
template <typename T, int WIN, int PARTS, int FAMILIES>
hipError_t launch_wide_win(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv,
                           int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    const int tilesX = (r.dW + 15) / 16;
    const dim3 grid(tilesX * (PARTS == 2 ? 4 : 16), (r.dyEnd - r.dyBase + 15) / 16, batch);      // at most 65535 tile rows: the caller bands taller outputs
    if (r.mode == AAI_MODE_FAST) {
        if constexpr ((FAMILIES & 2) == 0) return hipErrorInvalidValue;
        else {
            hipLaunchKernelGGL((aai_wide_fast_kernel<T, WIN, PARTS>), grid, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks);
            return hipGetLastError();
        }
    }
    if constexpr ((FAMILIES & 1) == 0) return hipErrorInvalidValue;
    else {
        if (q.hiPrec) hipLaunchKernelGGL((aai_wide_kernel<T, WIN, true, PARTS>), grid, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks);
        else hipLaunchKernelGGL((aai_wide_kernel<T, WIN, false, PARTS>), grid, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks);
        return hipGetLastError();
    }
}

template <typename T, int PARTS, int FAMILIES>
hipError_t launch_wide_parts(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv,
                             int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    switch (r.mode == AAI_MODE_FAST ? q.winFast : q.win) {
    case 5: return launch_wide_win<T, 5, PARTS, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 6: return launch_wide_win<T, 6, PARTS, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 7: return launch_wide_win<T, 7, PARTS, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    case 8: return launch_wide_win<T, 8, PARTS, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    default: return hipErrorInvalidValue;
    }
}

template <typename T, int FAMILIES>
hipError_t launch_wide_typed(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch,
                             const unsigned long long *skipMasks, hipStream_t stream)
{
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    const int parts = r.mode == AAI_MODE_FAST ? q.partsFast : q.parts;
    if (parts != r.wide) return hipErrorInvalidValue;
    if (parts == 2) return launch_wide_parts<T, 2, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    return launch_wide_parts<T, 4, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
}

template <int PARTS>
hipError_t launch_wide_scan_parts(const RotLaunch &r, const QuadConsts<float> &q, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream)
{
    const int tilesX = (r.dW + 15) / 16, tileRows = (r.dH + 15) / 16;
    for (int t0 = 0; t0 < tileRows; t0 += 65535) {         // grid.y carries at most 65535 tiles
        const dim3 grid(tilesX * (PARTS == 2 ? 4 : 16), tileRows - t0 < 65535 ? tileRows - t0 : 65535, 1);
#define AAI_WIDE_SCAN(W)                                                                                                                       \
    case W:                                                                                                                                    \
        if (r.mode == AAI_MODE_FAST) hipLaunchKernelGGL((aai_wide_fast_scan_kernel<W, PARTS>), grid, dim3(kQuadBlock), 0, stream, r, q, laneMasks, counter, t0); \
        else if (q.hiPrec) hipLaunchKernelGGL((aai_wide_scan_kernel<W, true, PARTS>), grid, dim3(kQuadBlock), 0, stream, r, q, laneMasks, counter, t0); \
        else hipLaunchKernelGGL((aai_wide_scan_kernel<W, false, PARTS>), grid, dim3(kQuadBlock), 0, stream, r, q, laneMasks, counter, t0);     \
        break;
        switch (r.mode == AAI_MODE_FAST ? q.winFast : q.win) {
            AAI_WIDE_SCAN(5) AAI_WIDE_SCAN(6) AAI_WIDE_SCAN(7) AAI_WIDE_SCAN(8)
        default: return hipErrorInvalidValue;
        }
#undef AAI_WIDE_SCAN
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

// ---- translation units (as in aai_rotated_cell.hip: a process loads the code object of the kernels it uses) ------------------------
// AAI_WIDE_PART: 1 = dispatch + scan kernels, 2 = fp32 area mode, 3 = fp32 fast mode, 4 = 8-bit, 5 = 16-bit sources; undefined = all
#define AAI_WIDE_ENTRY(name, T) \
    hipError_t name(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream)
AAI_WIDE_ENTRY(launch_wide_f32_area, float);
AAI_WIDE_ENTRY(launch_wide_f32_fast, float);
AAI_WIDE_ENTRY(launch_wide_u8, unsigned char);
AAI_WIDE_ENTRY(launch_wide_u16, unsigned short);
#if !defined(AAI_WIDE_PART) || AAI_WIDE_PART == 2
AAI_WIDE_ENTRY(launch_wide_f32_area, float) { return launch_wide_typed<float, 1>(r, m, src, sv, dst, dv, batch, skipMasks, stream); }
#endif
#if !defined(AAI_WIDE_PART) || AAI_WIDE_PART == 3
AAI_WIDE_ENTRY(launch_wide_f32_fast, float) { return launch_wide_typed<float, 2>(r, m, src, sv, dst, dv, batch, skipMasks, stream); }
#endif
#if !defined(AAI_WIDE_PART) || AAI_WIDE_PART == 4
AAI_WIDE_ENTRY(launch_wide_u8, unsigned char) { return launch_wide_typed<unsigned char, 3>(r, m, src, sv, dst, dv, batch, skipMasks, stream); }
#endif
#if !defined(AAI_WIDE_PART) || AAI_WIDE_PART == 5
AAI_WIDE_ENTRY(launch_wide_u16, unsigned short) { return launch_wide_typed<unsigned short, 3>(r, m, src, sv, dst, dv, batch, skipMasks, stream); }
#endif
#undef AAI_WIDE_ENTRY

#if !defined(AAI_WIDE_PART) || AAI_WIDE_PART == 1
bool wide_can_serve(const RotLaunch &r, int srcType, ImageView sv)
{
    static const bool off = [] { const char *e = experiment_env("AAI_WIDE"); return e && atoi(e) == 0; }();      // experiments: AAI_WIDE=0 keeps the runs kernel
    if (off || !r.wide || (r.mode != AAI_MODE_AREA && r.mode != AAI_MODE_FAST) || r.chan != 1 || r.scale != 1 || (r.dyBase & 15) != 0) return false;
    // (tilesX * 16 blocks along grid.x)
    if ((int64_t)((r.dW + 15) / 16) * 16 > 2147483647ll) return false;
    return quad_can_address(r, srcType, sv);
}

hipError_t launch_wide(const RotLaunch &r, const QuadMap &map, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    if (r.dW <= 0 || r.dyEnd <= r.dyBase || batch <= 0) return hipSuccess;
    QuadMap m = map;
    const int64_t esz = srcType == SRC_U8 ? 1 : srcType == SRC_U16 ? 2 : 4;
    m.anchorRows = (int64_t)r.H * sv.rowStride * esz >= ((int64_t)1 << 32) ? quad_anchor_rows(r) : 0;
    switch (srcType) {
    case SRC_U8: return launch_wide_u8(r, m, static_cast<const unsigned char *>(src), sv, dst, dv, batch, skipMasks, stream);
    case SRC_U16: return launch_wide_u16(r, m, static_cast<const unsigned short *>(src), sv, dst, dv, batch, skipMasks, stream);
    default:
        if (r.mode == AAI_MODE_FAST) return launch_wide_f32_fast(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream);
        return launch_wide_f32_area(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream);
    }
}

hipError_t launch_wide_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream)
{
    if (r.dW <= 0 || r.dH <= 0 || !r.wide) return hipSuccess;
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    const int parts = r.mode == AAI_MODE_FAST ? q.partsFast : q.parts;
    if (parts != r.wide) return hipErrorInvalidValue;
    return parts == 2 ? launch_wide_scan_parts<2>(r, q, laneMasks, counter, stream) : launch_wide_scan_parts<4>(r, q, laneMasks, counter, stream);
}

#endif      // AAI_WIDE_PART 1

}  // namespace aai
