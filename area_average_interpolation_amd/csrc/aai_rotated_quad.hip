// aai_rotated_quad.hip -- K2, the area average at a general rotation, in its fp32 "quad" formulation for gfx950, and K3,
// fast mode, in the same frame (the arithmetic lives in aai_rot_quad.hpp, shared with the CPU replay of the test-suite).
//
// Replaces Source.cpp:413-579 + 986-1431 (area) and 868-907 + 837-864 (fast) of the reference.  One lane per dst pixel, a wave covers a 16 x 4 dst
// tile, no barrier.  The dst pixel's centre is the only double-precision quantity (coordinates reach ~3e4 virtual
// pixels); everything after it is relative to the nearest virtual pixel and runs in fp32 at twice the fp64 issue
// rate and half the registers.
//
// Source pixels: every lane fetches its own WIN x WIN window with independent loads (all in flight while the
// window is classified), parks it in one LDS column of its own (slot-major, so a wave's accesses are conflict
// free and no barrier is needed) and the area passes read values at LDS latency instead of waiting for a
// dependent global load per (dst, src) pair -- with loads inside the passes the kernel spent 42 % of its wave
// cycles in s_waitcnt (profiles/r02_cfg3_sq_counters.txt).  The bound is VALU issue, not HBM: cfg3 moves 26.9
// algorithmic bytes per dst pixel through ~1.1 k vector instructions per wave.
//
// The production kernel takes no numerically delicate decision by itself: the plan runs the SCAN kernel once per
// geometry (same code, no pixel loads); pixels it flags -- a decision within QuadConsts::margin of its threshold,
// or a border pixel with too little total area for fp32 weights -- are recomputed afterwards by the
// double-precision fix-up pass (aai_rotated_kernel<STRICT>), exactly like the knife-edge pixels.
//
// Kernels: aai_quad_kernel (plain images), aai_quad_multi_kernel (2..4 interleaved channels, packed LDS slots),
// aai_quad_fast_kernel (fast mode: window in registers, no LDS), aai_quad_scan_kernel (the plan's scan for all three),
// aai_flag_list_kernel (flag words -> pixel list).
#include <algorithm>
#include <cstdlib>
#include "aai_kernels.hpp"
#include "aai_rot_quad.hpp"
#include "aai_quad_src.hpp"

#include <type_traits>

namespace aai {

namespace {

// Tile order: launch order (x fastest).  Workgroups are dealt round-robin over the 8 XCDs, so neighbouring tiles sit on
// different L2s and the fast kernel fetches 1.75 x the source at config 3 -- yet giving each XCD contiguous bands, or
// cyclically dealt 8 x 8 super-tiles, made every kernel SLOWER (profiles/r02_xcd_tile_order.txt): these kernels are bound
// by instruction issue, and the empty corners of a rotated output are spread evenly only in launch order.

// waves per SIMD that the staged windows leave room for (160 KiB of LDS per CU, WIN * WIN KiB per 256-lane block):
// the register budget follows it
constexpr int quad_waves_per_simd(int win) { return 160 / (win * win) >= 8 ? 8 : 160 / (win * win); }

template <typename T, int WIN, bool SCALED, bool HP>
__global__ __launch_bounds__(kQuadBlock, quad_waves_per_simd(WIN) - (HP && quad_waves_per_simd(WIN) > 2 ? 1 : 0)) void aai_quad_kernel(RotLaunch r, QuadConsts<float> q, QuadMap m, const T *__restrict__ src,
                                                             ImageView sv, float *__restrict__ dst, ImageView dv,
                                                             const unsigned long long *__restrict__ skipMasks, const int *__restrict__ live, int xcdRows)
{
    __shared__ float window[WIN * WIN][kQuadBlock];
    const int tid = threadIdx.x;
    int tx = blockIdx.x, ty = blockIdx.y;
    xcd_tile(xcdRows, tx, ty);
    const int dx = tx * 16 + (tid & 15);
    const int dy = r.dyBase + ty * 16 + (tid >> 4);
    if (!(dx < r.dW && dy < r.dyEnd)) return;
    if (skipMasks) {
        // pixels the plan's scans flagged belong to the double-precision pass, which runs beside this kernel
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned long long mask = skipMasks[((size_t)(ty + r.dyBase / 16) * gridDim.x + tx) * (kQuadBlock / 64) + wave];
        if ((mask >> (tid & 63)) & 1ull) return;
    }
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride + dx;
    if (live) {
        // a tile in a corner of the rotated canvas: every pixel is 0 (rot_live_cols), no centre is computed
        const int first = live[2 * (ty + r.dyBase / 16)], last = live[2 * (ty + r.dyBase / 16) + 1];      // (scalar loads: wave-uniform)
        if (tx < first || tx > last) { *out = 0.f; return; }
    }

    double px, py;
    quad_centre(r, dx, dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    float value = 0.f;
    // a centre further than the window's reach from the lattice touches nothing (and stays inside int range)
    if (cx > -16.0 && cx < (double)r.mW + 16.0 && cy > -16.0 && cy < (double)r.mH + 16.0) {
        QuadSrc<T, WIN, SCALED, true> s;
        s.img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
        s.m = &m; s.mW = r.mW; s.mH = r.mH; s.lds = window; s.tid = tid;
        float sumA, sumVA[1];
        quad_pixel<float, WIN, false, HP, 1>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA);
        value = sumA > 0.f ? sumVA[0] / sumA : 0.f;                   // Source.cpp:577
    }
    *out = value;
}

constexpr int kFastXcdRowsDefault = 1;       // tile rows per XCD band of the 16 x 4-wave kernels (profiles/r04_fast_xcd.txt)

// K3 in the same frame: fast mode (Source.cpp:868-907), the mean of the virtual pixels whose centres lie in the dst
// square.  The window stays in the registers it was fetched into; no LDS.
template <typename T, int WIN, bool SCALED>
__global__ __launch_bounds__(kQuadBlock) void aai_quad_fast_kernel(RotLaunch r, QuadConsts<float> q, QuadMap m, const T *__restrict__ src, ImageView sv,
                                                                  float *__restrict__ dst, ImageView dv, const unsigned long long *__restrict__ skipMasks,
                                                                  const int *__restrict__ live, int xcdRows)
{
    const int tid = threadIdx.x;
    int tx = blockIdx.x, ty = blockIdx.y;
    xcd_tile(xcdRows, tx, ty);                                 // XCD-aware tile order (aai_quad_src.hpp); rows beyond the image leave below
    const int dx = tx * 16 + (tid & 15);
    const int dy = r.dyBase + ty * 16 + (tid >> 4);
    if (!(dx < r.dW && dy < r.dyEnd)) return;
    if (skipMasks) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned long long mask = skipMasks[((size_t)(ty + r.dyBase / 16) * gridDim.x + tx) * (kQuadBlock / 64) + wave];
        if ((mask >> (tid & 63)) & 1ull) return;
    }
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride + dx;
    if (live) {
        // a tile in a corner of the rotated canvas: every pixel is 0 (rot_live_cols), no centre is computed
        const int first = live[2 * (ty + r.dyBase / 16)], last = live[2 * (ty + r.dyBase / 16) + 1];      // (scalar loads: wave-uniform)
        if (tx < first || tx > last) { *out = 0.f; return; }
    }

    double px, py;
    quad_centre(r, dx, dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    float value = 0.f;
    if (cx > -16.0 && cx < (double)r.mW + 16.0 && cy > -16.0 && cy < (double)r.mH + 16.0) {
        const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
        value = fast_window_value<T, WIN, SCALED>(r, q, m, img, (int)cx, (int)cy, px - cx, py - cy, tid);
    }
    *out = value;
}

// K3 for outputs as large as their source or larger (replication = up-sampling, and ratios up to 1.6), where the STORES set the time:
// 2.15 GB of config 5 went out in 64-byte pieces (a 16 x 4 tile per wave) that start at any multiple of 4 bytes.  Here a wave
// owns 64 consecutive pixels of a dst row, shifted left to the 256-byte boundary below them (two whole 128-byte lines per
// store, written around the caches), and walks the four rows its wave of the 16 x 16 tile owns -- so the skip masks, the
// live tile spans and the fix-up list keep their tiling.  Per-pixel arithmetic is untouched: results are identical.
template <typename T, int WIN, bool SCALED>
// (at most 96 scalar registers: with 97 ... 112 a CU admits six 256-lane workgroups instead of seven -- MI355X_MICROARCH.md, Residency --
// and this store-bound kernel loses a tenth of its rate: config 5 fast 1.50 -> 1.66 ms when a change of the source map took it to 99)
__global__ __launch_bounds__(kQuadBlock) __attribute__((amdgpu_num_sgpr(96))) void aai_quad_fast_rows_kernel(RotLaunch r, QuadConsts<float> q, QuadMap m, const T *__restrict__ src, ImageView sv,
                                                                       float *__restrict__ dst, ImageView dv, const unsigned long long *__restrict__ skipMasks,
                                                                       const int *__restrict__ live, int tilesX)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tileRow = (int)blockIdx.y + r.dyBase / 16;
    int first = 0, last = 0x7fffffff;
    if (live) { first = live[2 * tileRow]; last = live[2 * tileRow + 1]; }
    for (int j = 0; j < 4; ++j) {
        const int dy = tileRow * 16 + wave * 4 + j;
        if (dy >= r.dyEnd) break;                                     // (wave-uniform)
        float *row = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride;
        const int shift = (int)((reinterpret_cast<uintptr_t>(row) >> 2) & 63);
        const int dx = (int)blockIdx.x * 64 + lane - shift;
        if (dx < 0 || dx >= r.dW) continue;
        const int tx = dx >> 4;
        if (skipMasks) {
            const unsigned long long mask = skipMasks[((size_t)tileRow * tilesX + tx) * (kQuadBlock / 64) + wave];
            if ((mask >> ((j << 4) | (dx & 15))) & 1ull) continue;
        }
        float *out = row + dx;
        // a tile in a corner of the rotated canvas: every pixel is 0 (rot_live_cols), no centre is computed
        if (tx < first || tx > last) { __builtin_nontemporal_store(0.f, out); continue; }
        double px, py;
        quad_centre(r, dx, dy, px, py);
        const double cx = floor(px + 0.5), cy = floor(py + 0.5);
        float value = 0.f;
        if (cx > -16.0 && cx < (double)r.mW + 16.0 && cy > -16.0 && cy < (double)r.mH + 16.0) {
            const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
            value = fast_window_value<T, WIN, SCALED>(r, q, m, img, (int)cx, (int)cy, px - cx, py - cy, tid);
        }
        __builtin_nontemporal_store(value, out);
    }
}

// K3 with the workgroup's source footprint staged through LDS -- MEASURED AND NOT SHIPPED (round 4, profiles/r04_fast_lds.txt): it exists
// in the experiments build only (make exp; AAI_FAST_LDS=1 selects it).  Idea: the 256 lanes of a workgroup bring
// the bounding box of the windows of their 16 x 16 dst tile into LDS with coalesced loads (a wave instruction reads 64 consecutive
// elements of a source row), meet at one barrier and take their windows from LDS.  Result: config 3's fast mode 158 us against 105 for
// the register-window kernel (2:1 at 45 degrees 256 / 148, 1.5:1 at 61 degrees 471 / 302, 4:1 164 / 129, 5:1 118 / 117): the per-lane
// window loads of the shipped kernel hit the CU's L1 88 % of the time, so there was no texture-path time to win, while the staged form
// fetches the slanted tile's whole bounding box (1.6 x the windows' footprint), serialises load -> barrier -> compute inside a
// workgroup and pays LDS traffic twice.  Results are identical to aai_quad_fast_kernel's, bit for bit (same quad_fast_pixel).
#if defined(AAI_EXPERIMENTS)
constexpr int kFastLdsUnroll = 4;            // box rows a wave has in flight per round of the staging loop
// the window of one lane, read from the staged box; `reg` protocol of quad_fast_pixel
template <int WIN>
struct FastLdsSrc {
    const float *lds;
    int pitch, A0, B0, A1, B1;               // box [A0, A1] x [B0, B1] along the contiguous (A) and the strided (B) virtual axis
    bool alongX;                             // A is virtual X (quadrants 0 / 2)
    float v[WIN * WIN];
    __device__ __forceinline__ void issue(int xg0, int yg0, unsigned long long, bool allInside)
    {
        const int a0 = (alongX ? xg0 : yg0) - A0, b0 = (alongX ? yg0 : xg0) - B0;
        if (allInside) {
            const float *p = lds + b0 * pitch + a0;
            if (alongX) {
#pragma unroll
                for (int j = 0; j < WIN; ++j)
#pragma unroll
                    for (int i = 0; i < WIN; ++i) v[j * WIN + i] = p[j * pitch + i];
            } else {
#pragma unroll
                for (int i = 0; i < WIN; ++i)
#pragma unroll
                    for (int j = 0; j < WIN; ++j) v[j * WIN + i] = p[i * pitch + j];
            }
            return;
        }
        // at the lattice's border: positions outside it are never used (quad_fast_pixel gives them a far-away coordinate); they are
        // read from the nearest staged element
        const int nA = A1 - A0, nB = B1 - B0;
#pragma unroll
        for (int j = 0; j < WIN; ++j)
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int a = min(max(a0 + (alongX ? i : j), 0), nA), b = min(max(b0 + (alongX ? j : i), 0), nB);
                v[j * WIN + i] = lds[b * pitch + a];
            }
    }
    __device__ __forceinline__ float reg(int slot) const { return v[slot]; }
};

template <int WIN>
__global__ __launch_bounds__(kQuadBlock) void aai_quad_fast_lds_kernel(RotLaunch r, QuadConsts<float> q, QuadMap m, FastTile ft, const float *__restrict__ src, ImageView sv,
                                                                      float *__restrict__ dst, ImageView dv, const unsigned long long *__restrict__ skipMasks,
                                                                      const int *__restrict__ live)
{
    extern __shared__ float staged[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = blockIdx.x, ty = blockIdx.y;
    const int dx = tx * 16 + (tid & 15);
    const int dy = r.dyBase + ty * 16 + (tid >> 4);
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride + dx;
    const bool inImage = dx < r.dW && dy < r.dyEnd;
    if (live) {
        // a tile in a corner of the rotated canvas: every pixel is 0 (rot_live_cols), nothing is staged (block-uniform: no barrier is missed)
        const int first = live[2 * (ty + r.dyBase / 16)], last = live[2 * (ty + r.dyBase / 16) + 1];
        if (tx < first || tx > last) {
            bool flagged = false;
            if (skipMasks && inImage) flagged = (skipMasks[((size_t)(ty + r.dyBase / 16) * gridDim.x + tx) * (kQuadBlock / 64) + wave] >> (tid & 63)) & 1ull;
            if (inImage && !flagged) *out = 0.f;
            return;
        }
    }
    // ---- the tile's box on the virtual lattice (block-uniform), clipped to the lattice --------------------------------------------
    int X0, X1, Y0, Y1;
    const bool any = fast_tile_box(r, ft, tx * 16, r.dyBase + ty * 16, X0, X1, Y0, Y1);      // (false: the box misses the lattice)
    const bool alongX = m.strideX == 1;
    const int A0 = alongX ? X0 : Y0, A1 = alongX ? X1 : Y1, B0 = alongX ? Y0 : X0, B1 = alongX ? Y1 : X1;
    const int nA = any ? A1 - A0 + 1 : 0, nB = any ? B1 - B0 + 1 : 0;
    const int pitch = nA | 1;
    // ---- staging: wave w takes rows w, w + 4, ... of the box, a lane the elements lane, lane + 64, ... of a row ----------------------
    {
        const char *img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
        const int flipA = alongX ? m.flipX : m.flipY, flipB = alongX ? m.flipY : m.flipX;
        const int nAimg = alongX ? m.nX : m.nY, nBimg = alongX ? m.nY : m.nX;
        const unsigned strideB = (unsigned)(alongX ? m.strideY : m.strideX) * 4u;
        for (int b = wave; b < nB; b += 4 * kFastLdsUnroll) {
            float t[kFastLdsUnroll][2];
#pragma unroll
            for (int u = 0; u < kFastLdsUnroll; ++u) {
                const int bb = b + 4 * u;
                const int uB = flipB ? nBimg - 1 - (B0 + bb) : B0 + bb;      // the strided axis: uB IS the source row
                // (a row band's buffer holds source rows [srcRow0, srcRow1): rows of the box outside it -- the corners of a slanted
                // box that no window touches -- are not fetched)
                const bool rowOk = bb < nB && uB >= r.srcRow0 && uB < r.srcRow1;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int a = lane + 64 * k;
                    const int uA = flipA ? nAimg - 1 - (A0 + a) : A0 + a;
                    t[u][k] = (rowOk && a < nA) ? *reinterpret_cast<const float *>(img + ((unsigned)uB * strideB + (unsigned)uA * 4u)) : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < kFastLdsUnroll; ++u) {
                const int bb = b + 4 * u;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int a = lane + 64 * k;
                    if (bb < nB && a < nA) staged[bb * pitch + a] = t[u][k];
                }
            }
        }
    }
    __syncthreads();
    if (!inImage) return;
    if (skipMasks) {
        const unsigned long long mask = skipMasks[((size_t)(ty + r.dyBase / 16) * gridDim.x + tx) * (kQuadBlock / 64) + wave];
        if ((mask >> (tid & 63)) & 1ull) return;
    }
    double px, py;
    quad_centre(r, dx, dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    float value = 0.f;
    if (cx > -16.0 && cx < (double)r.mW + 16.0 && cy > -16.0 && cy < (double)r.mH + 16.0) {
        FastLdsSrc<WIN> s;
        s.lds = staged; s.pitch = pitch; s.A0 = A0; s.B0 = B0; s.A1 = A1; s.B1 = B1; s.alongX = alongX;
        float sum;
        int count;
        quad_fast_pixel<float, WIN, false>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sum, count);
        value = count > 0 ? sum / (float)count : 0.f;                 // Source.cpp:905
    }
    *out = value;
}

#endif      // AAI_EXPERIMENTS

// Interleaved channels: areas once per (dst, src) pair, applied to every channel (four accumulators).  Dynamic LDS:
// WIN * WIN * words KiB per block.
template <typename T, int WIN, bool SCALED, int WORDS>
__global__ __launch_bounds__(kQuadBlock, 160 / (WIN * WIN * WORDS) >= 6 ? 6 : (160 / (WIN * WIN * WORDS) >= 2 ? 160 / (WIN * WIN * WORDS) : 2))
void aai_quad_multi_kernel(RotLaunch r, QuadConsts<float> q, QuadMap m, const T *__restrict__ src, ImageView sv, float *__restrict__ dst, ImageView dv,
                           const unsigned long long *__restrict__ skipMasks, int xcdRows)
{
    extern __shared__ unsigned windowWords[];
    const int tid = threadIdx.x;
    int tx = blockIdx.x, ty = blockIdx.y;
    xcd_tile(xcdRows, tx, ty);                                 // XCD-aware tile order (aai_quad_src.hpp); rows beyond the image leave below
    const int dx = tx * 16 + (tid & 15);
    const int dy = r.dyBase + ty * 16 + (tid >> 4);
    if (!(dx < r.dW && dy < r.dyEnd)) return;
    if (skipMasks) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned long long mask = skipMasks[((size_t)(ty + r.dyBase / 16) * gridDim.x + tx) * (kQuadBlock / 64) + wave];
        if ((mask >> (tid & 63)) & 1ull) return;
    }
    const int chan = r.chan;
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride + (int64_t)dx * chan;

    double px, py;
    quad_centre(r, dx, dy, px, py);
    const double cx = floor(px + 0.5), cy = floor(py + 0.5);
    float value[kQuadMaxChan] = {0.f, 0.f, 0.f, 0.f};
    if (cx > -16.0 && cx < (double)r.mW + 16.0 && cy > -16.0 && cy < (double)r.mH + 16.0) {
        QuadSrcMulti<T, WIN, SCALED, WORDS> s;
        s.img = reinterpret_cast<const char *>(src + (int64_t)blockIdx.z * sv.imageStride + m.base);
        s.m = &m; s.mW = r.mW; s.mH = r.mH; s.chan = chan; s.lds = windowWords; s.tid = tid;
        float sumA, sumVA[kQuadMaxChan];
        if (q.hiPrec) quad_pixel<float, WIN, false, true, kQuadMaxChan>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA);
        else quad_pixel<float, WIN, false, false, kQuadMaxChan>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA);
#pragma unroll
        for (int c = 0; c < kQuadMaxChan; ++c) value[c] = sumA > 0.f ? sumVA[c] / sumA : 0.f;      // Source.cpp:577
    }
#pragma unroll
    for (int c = 0; c < kQuadMaxChan; ++c)
        if (c < chan) out[c] = value[c];
}

// Once per geometry: the same arithmetic without pixel loads.  Flags one bit per dst pixel in the 64-bit word of its
// wave (the 16 x 16 tiling of aai_knife_scan_kernel, whose bits this kernel adds to) and counts the newly flagged
// pixels in counter[0].
// FAST: the scan of aai_quad_fast_kernel (HP unused)
template <int WIN, bool HP, bool FAST = false>
__global__ __launch_bounds__(kQuadBlock) void aai_quad_scan_kernel(RotLaunch r, QuadConsts<float> q, unsigned long long *__restrict__ laneMasks,
                                                                  unsigned *__restrict__ counter, int tileRow0)
{
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dx = blockIdx.x * 16 + (tid & 15);
    const int dy = (tileRow0 + blockIdx.y) * 16 + (tid >> 4);
    bool uncertain = false;
    if (dx < r.dW && dy < r.dH) {
        double px, py;
        quad_centre(r, dx, dy, px, py);
        const double cx = floor(px + 0.5), cy = floor(py + 0.5);
        if (cx > -16.0 && cx < (double)r.mW + 16.0 && cy > -16.0 && cy < (double)r.mH + 16.0) {
            NoSrc s;
            float sumA, sumVA[1];
            int count;
            if (FAST) uncertain = quad_fast_pixel<float, WIN, true>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, count);
            else uncertain = quad_pixel<float, WIN, true, HP, 1>(q, (int)cx, (int)cy, px - cx, py - cy, r.mW, r.mH, s, sumA, sumVA);
        }
    }
    const unsigned long long any = __ballot(uncertain);
    if ((tid & 63) == 0 && any != 0ull) {
        unsigned long long *f = laneMasks + ((size_t)(tileRow0 + blockIdx.y) * gridDim.x + blockIdx.x) * (kQuadBlock / 64) + wave;
        const unsigned long long fresh = any & ~*f;
        if (fresh) { *f |= fresh; atomicAdd(counter, (unsigned)__popcll(fresh)); }
    }
}

#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 1
// flagged dst pixels as a list of (dy << 32 | dx)... kept as two 32-bit words per pixel; any order (the fix-up pass
// writes each listed pixel once)
__global__ __launch_bounds__(256) void aai_flag_list_kernel(const unsigned long long *__restrict__ laneMasks, size_t waves, unsigned tilesX,
                                                           uint2 *__restrict__ list, unsigned *__restrict__ cursor, unsigned capacity)
{
    for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < waves; w += (size_t)gridDim.x * 256) {
        unsigned long long mask = laneMasks[w];
        if (!mask) continue;
        unsigned k = atomicAdd(cursor, (unsigned)__popcll(mask));
        const size_t tile = w >> 2;
        const unsigned x0 = (unsigned)(tile % tilesX) * 16u, y0 = (unsigned)(tile / tilesX) * 16u + (unsigned)(w & 3) * 4u;
        while (mask) {
            const int lane = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            if (k < capacity) list[k] = make_uint2(x0 + (unsigned)(lane & 15), y0 + (unsigned)(lane >> 4));
            ++k;
        }
    }
}

#endif

// AAI_FAST_ROWS (A/B switch of tools/, experiments build only): 0 = always the 16 x 4 wave, 1 = row-shaped wave under replication and
// at ratios up to 1.6 (default), 2 = row-shaped wave for every plain image below 4 GiB
static int quad_fast_rows_mode()
{
    static const int mode = [] { const char *e = experiment_env("AAI_FAST_ROWS"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1; }();
    return mode;
}

#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 1

// summary of the lane masks: one bit per 16 x 16 tile that holds a flagged pixel (QuadMap::tileFlags)
__global__ __launch_bounds__(256) void aai_tile_flags_kernel(const unsigned long long *__restrict__ laneMasks, unsigned tilesX, unsigned tilesY, unsigned rowWords,
                                                            unsigned *__restrict__ out)
{
    const size_t tiles = (size_t)tilesX * tilesY;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < tiles; t += (size_t)gridDim.x * 256) {
        if (laneMasks[4 * t] | laneMasks[4 * t + 1] | laneMasks[4 * t + 2] | laneMasks[4 * t + 3]) {
            const unsigned tx = (unsigned)(t % tilesX), ty = (unsigned)(t / tilesX);
            atomicOr(out + (size_t)ty * rowWords + (tx >> 5), 1u << (tx & 31u));
        }
    }
}
#endif

template <typename T, int WIN, int FAMILIES>
hipError_t launch_quad_win(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv,
                           int batch, const unsigned long long *skipMasks, hipStream_t stream, const int *live)
{
    const dim3 grid((r.dW + 15) / 16, (r.dyEnd - r.dyBase + 15) / 16, batch);      // at most 65535 tile rows: the caller bands taller outputs
    if (r.mode == AAI_MODE_FAST) {
      if constexpr ((FAMILIES & 2) == 0) return hipErrorInvalidValue;
      else {
        // ... and without replication while a dst pixel is at most 1.6 source pixels wide (1.5:1 at 17.5 degrees 237 -> 189 us, 1:1 at
        // 45 degrees 480 -> 426; from 2:1 on the slanted line of source pixels a row-shaped wave reads costs more than its stores
        // save: 2:1 at 45 degrees 148 -> 206 us, 3:1 at 17.5 degrees 100 -> 158) -- profiles/r03_store_paths.txt
#if defined(AAI_EXPERIMENTS)
        if constexpr (std::is_same<T, float>::value) {
            // experiments build: AAI_FAST_LDS=1 sends plain fp32 images without replication whose tile footprint fits LDS to the staged form
            static const bool ldsOn = [] { const char *e = experiment_env("AAI_FAST_LDS"); return e && atoi(e) != 0; }();
            FastTile ft;
            if (ldsOn && m.scale == 1 && m.anchorRows == 0 && make_fast_tile(r, q, ft)) {
                const size_t lds = (size_t)(ft.maxSide | 1) * (size_t)ft.maxSide * sizeof(float);
                static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&aai_quad_fast_lds_kernel<WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, kFastLdsBytes);
                (void)once;
                hipLaunchKernelGGL((aai_quad_fast_lds_kernel<WIN>), grid, dim3(kQuadBlock), lds, stream, r, q, m, ft, src, sv, dst, dv, skipMasks, live);
                set_quad_kernel_note("aai_quad_fast_kernel<lds>");
                return hipGetLastError();
            }
        }
#endif
        const bool rowShaped = quad_fast_rows_mode() >= 2 || (quad_fast_rows_mode() == 1 && (m.scale > 1 || r.side <= 1.6));
        if (rowShaped && m.anchorRows == 0) {
            // (images of 4 GiB and more keep the 16 x 4 wave: their anchor rows are sized for it)
            const dim3 rows((r.dW + 63) / 64 + 1, grid.y, batch);
            if (m.scale > 1) hipLaunchKernelGGL((aai_quad_fast_rows_kernel<T, WIN, true>), rows, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, (int)grid.x);
            else hipLaunchKernelGGL((aai_quad_fast_rows_kernel<T, WIN, false>), rows, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, (int)grid.x);
        } else {
            // XCD-aware tile order (xcd_tile): the fast-mode kernel is bound by its fabric traffic, and with one tile row per XCD band
            // config 3 asks for 1.23 x its source instead of 1.75 x: 98.5 -> 91.6 us (2 rows 95.1, 4 rows 96.6; every geometry tried
            // equal or faster: profiles/r04_fast_xcd.txt).  Experiments build: AAI_XCD_ROWS.
            const int band = xcd_band(kFastXcdRowsDefault);
            const int gy = xcd_grid_rows((int)grid.y, band);
            const dim3 g(grid.x, gy ? gy : grid.y, grid.z);
            const int on = gy ? band : 0;
            if (m.scale > 1) hipLaunchKernelGGL((aai_quad_fast_kernel<T, WIN, true>), g, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, on);
            else hipLaunchKernelGGL((aai_quad_fast_kernel<T, WIN, false>), g, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, on);
        }
        return hipGetLastError();
      }
    }
    // (the one-lane-per-pixel area kernel takes the same XCD-aware tile order as the fast-mode kernel)
    const int bandA = xcd_band(kFastXcdRowsDefault);
    const int gyA = xcd_grid_rows((int)grid.y, bandA);
    const dim3 gridA(grid.x, gyA ? gyA : grid.y, grid.z);
    const int onA = gyA ? bandA : 0;
    if constexpr ((FAMILIES & 1) == 0) return hipErrorInvalidValue;
    else if (m.scale > 1) {
        if (q.hiPrec) hipLaunchKernelGGL((aai_quad_kernel<T, WIN, true, true>), gridA, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, onA);
        else hipLaunchKernelGGL((aai_quad_kernel<T, WIN, true, false>), gridA, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, onA);
    } else {
        if (q.hiPrec) hipLaunchKernelGGL((aai_quad_kernel<T, WIN, false, true>), gridA, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, onA);
        else hipLaunchKernelGGL((aai_quad_kernel<T, WIN, false, false>), gridA, dim3(kQuadBlock), 0, stream, r, q, m, src, sv, dst, dv, skipMasks, live, onA);
    }
    return hipGetLastError();
}

// LDS words one window slot takes with `chan` interleaved channels of T
template <typename T> int quad_slot_words(int chan) { return sizeof(T) == 4 ? chan : (sizeof(T) == 2 ? (chan + 1) / 2 : 1); }

template <typename T, int WIN, int WORDS>
hipError_t launch_quad_multi_words(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv,
                                   int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    const dim3 tiles((r.dW + 15) / 16, (r.dyEnd - r.dyBase + 15) / 16, batch);
    // (the same XCD-aware tile order as the plain kernels of this file)
    const int band = xcd_band(kFastXcdRowsDefault);
    const int gy = xcd_grid_rows((int)tiles.y, band);
    const dim3 grid(tiles.x, gy ? gy : tiles.y, tiles.z);
    const int xcdRows = gy ? band : 0;
    const size_t lds = (size_t)WIN * WIN * WORDS * kQuadBlock * sizeof(unsigned);
    if (m.scale > 1) {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&aai_quad_multi_kernel<T, WIN, true, WORDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)once;
        hipLaunchKernelGGL((aai_quad_multi_kernel<T, WIN, true, WORDS>), grid, dim3(kQuadBlock), lds, stream, r, q, m, src, sv, dst, dv, skipMasks, xcdRows);
    } else {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void *>(&aai_quad_multi_kernel<T, WIN, false, WORDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)once;
        hipLaunchKernelGGL((aai_quad_multi_kernel<T, WIN, false, WORDS>), grid, dim3(kQuadBlock), lds, stream, r, q, m, src, sv, dst, dv, skipMasks, xcdRows);
    }
    return hipGetLastError();
}

template <typename T, int WIN>
hipError_t launch_quad_multi_win(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv,
                                 int batch, const unsigned long long *skipMasks, hipStream_t stream)
{
    // quad_can_address() keeps WIN * WIN * words <= 80 KiB of LDS per block
    const int words = quad_slot_words<T>(r.chan);
    if (sizeof(T) == 1) return launch_quad_multi_words<T, WIN, 1>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    if (sizeof(T) == 2) {
        if (words == 1) return launch_quad_multi_words<T, WIN, 1>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
        return launch_quad_multi_words<T, WIN, (sizeof(T) == 2 ? 2 : 1)>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    }
    if (WIN * WIN * words > 80) return hipErrorInvalidValue;
    if (words == 2) return launch_quad_multi_words<T, WIN, (sizeof(T) == 4 && WIN * WIN * 2 <= 80 ? 2 : 1)>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    if (words == 3) return launch_quad_multi_words<T, WIN, (sizeof(T) == 4 && WIN * WIN * 3 <= 80 ? 3 : 1)>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
    return launch_quad_multi_words<T, WIN, (sizeof(T) == 4 && WIN * WIN * 4 <= 80 ? 4 : 1)>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
}

// FAMILIES: which kernel families this translation unit holds (see "translation units" below): 1 = area mode, plain images;
// 2 = fast mode; 4 = interleaved channels
template <typename T, int FAMILIES>
hipError_t launch_quad_typed(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch,
                             const unsigned long long *skipMasks, hipStream_t stream, const int *live)
{
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    if (r.chan > 1) {
        if constexpr ((FAMILIES & 4) == 0) return hipErrorInvalidValue;
        else switch (q.win) {
        case 3: return launch_quad_multi_win<T, 3>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
        case 4: return launch_quad_multi_win<T, 4>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
        case 5: return launch_quad_multi_win<T, 5>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
        case 6: return launch_quad_multi_win<T, 6>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
        case 7: return launch_quad_multi_win<T, 7>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
        default: return launch_quad_multi_win<T, 8>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream);
        }
    }
    if constexpr ((FAMILIES & 3) == 0) return hipErrorInvalidValue;
    else switch (r.mode == AAI_MODE_FAST ? q.winFast : q.win) {
    case 2: return launch_quad_win<T, 2, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream, live);
    case 3: return launch_quad_win<T, 3, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream, live);
    case 4: return launch_quad_win<T, 4, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream, live);
    case 5: return launch_quad_win<T, 5, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream, live);
    case 6: return launch_quad_win<T, 6, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream, live);
    case 7: return launch_quad_win<T, 7, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream, live);
    default: return launch_quad_win<T, 8, FAMILIES>(r, q, m, src, sv, dst, dv, batch, skipMasks, stream, live);
    }
}

}  // namespace

// ---- translation units ----------------------------------------------------------------------------------------------------------
// As in aai_rotated_cell.hip: the runtime loads a translation unit's code object on the first launch of one of its kernels, so the
// Makefile compiles this file once per AAI_QUAD_PART and a process pays for the kernels it uses: 1 = dispatch, scans, list / tile
// flag kernels; 2 = fp32 plain images, area mode; 3 = fp32, fast mode (the reference's default mode); 4 = fp32 interleaved
// channels; 5 = 8-bit, 6 = 16-bit sources (all families).  (AAI_QUAD_PART undefined = everything in one unit.)
#define AAI_QUAD_ENTRY(name, T) \
    hipError_t name(const RotLaunch &r, const QuadMap &m, const T *src, ImageView sv, float *dst, ImageView dv, int batch, const unsigned long long *skipMasks, hipStream_t stream, const int *live)
AAI_QUAD_ENTRY(launch_quad_f32_area, float);
AAI_QUAD_ENTRY(launch_quad_f32_fast, float);
AAI_QUAD_ENTRY(launch_quad_f32_multi, float);
AAI_QUAD_ENTRY(launch_quad_u8, unsigned char);
AAI_QUAD_ENTRY(launch_quad_u16, unsigned short);
#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 2
AAI_QUAD_ENTRY(launch_quad_f32_area, float) { return launch_quad_typed<float, 1>(r, m, src, sv, dst, dv, batch, skipMasks, stream, live); }
#endif
#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 3
AAI_QUAD_ENTRY(launch_quad_f32_fast, float) { return launch_quad_typed<float, 2>(r, m, src, sv, dst, dv, batch, skipMasks, stream, live); }
#endif
#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 4
AAI_QUAD_ENTRY(launch_quad_f32_multi, float) { return launch_quad_typed<float, 4>(r, m, src, sv, dst, dv, batch, skipMasks, stream, live); }
#endif
#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 5
AAI_QUAD_ENTRY(launch_quad_u8, unsigned char) { return launch_quad_typed<unsigned char, 7>(r, m, src, sv, dst, dv, batch, skipMasks, stream, live); }
#endif
#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 6
AAI_QUAD_ENTRY(launch_quad_u16, unsigned short) { return launch_quad_typed<unsigned short, 7>(r, m, src, sv, dst, dv, batch, skipMasks, stream, live); }
#endif
#undef AAI_QUAD_ENTRY

#if !defined(AAI_QUAD_PART) || AAI_QUAD_PART == 1
// How many source rows apart two lanes of one wave (a 16 x 4 dst tile) can read: their centres differ by at most
// 15.6 dst pixel sides, each reaches half a window further, plus slack for rounding and clamping.
int quad_anchor_rows(const RotLaunch &r)
{
    const int win = (int)floor(2.0 * (r.h * (r.c + r.s) - 0.5 + 1e-5)) + 3;
    return (int)ceil(15.6 * 2.0 * r.h / r.scale) + win + 4;
}

bool quad_can_address(const RotLaunch &r, int srcType, ImageView sv)
{
    // lanes address their pixels with unsigned 32-bit byte offsets from the image's first element -- or, for plain images
    // of 4 GiB and more, from an anchor row of their wave (quad_anchor_rows)
    const int64_t esz = srcType == SRC_U8 ? 1 : srcType == SRC_U16 ? 2 : 4;
    if ((int64_t)r.H * sv.rowStride * esz >= ((int64_t)1 << 32)) {
        if (r.chan > 1) return false;
        const int win = (int)floor(2.0 * (r.h * (r.c + r.s) - 0.5 + 1e-5)) + 3;
        if ((int64_t)(2 * quad_anchor_rows(r) + win + 2) * sv.rowStride * esz >= ((int64_t)1 << 32)) return false;
    }
    if (r.chan > 1) {
        // interleaved channels: the staged window (win^2 slots of `words` LDS words per lane) must leave room for two
        // workgroups per CU
        const int win = (int)floor(2.0 * (r.h * (r.c + r.s) - 0.5 + 1e-5)) + 3;
        const int words = esz == 4 ? r.chan : (esz == 2 ? (r.chan + 1) / 2 : 1);
        if (win * win * words > 80) return false;
    }
    return true;
}

static thread_local const char *g_quadKernelNote = nullptr;
void set_quad_kernel_note(const char *note) { g_quadKernelNote = note; }
const char *quad_kernel_note() { return g_quadKernelNote; }

hipError_t launch_quad(const RotLaunch &r, const QuadMap &map, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, const unsigned long long *skipMasks, hipStream_t stream, const int *live)
{
    set_quad_kernel_note(nullptr);
    if (r.dW <= 0 || r.dyEnd <= r.dyBase || batch <= 0) return hipSuccess;
    QuadMap m = map;
    const int64_t esz = srcType == SRC_U8 ? 1 : srcType == SRC_U16 ? 2 : 4;
    m.anchorRows = (int64_t)r.H * sv.rowStride * esz >= ((int64_t)1 << 32) ? quad_anchor_rows(r) : 0;
    switch (srcType) {
    case SRC_U8: return launch_quad_u8(r, m, static_cast<const unsigned char *>(src), sv, dst, dv, batch, skipMasks, stream, live);
    case SRC_U16: return launch_quad_u16(r, m, static_cast<const unsigned short *>(src), sv, dst, dv, batch, skipMasks, stream, live);
    default:
        if (r.chan > 1) return launch_quad_f32_multi(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream, live);
        if (r.mode == AAI_MODE_FAST) return launch_quad_f32_fast(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream, live);
        return launch_quad_f32_area(r, m, static_cast<const float *>(src), sv, dst, dv, batch, skipMasks, stream, live);
    }
}

hipError_t launch_quad_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream)
{
    if (r.dW <= 0 || r.dH <= 0) return hipSuccess;
    const QuadConsts<float> q = make_quad_consts<float>(r.side, r.c, r.s, r.policy, r.scale);
    const int tileRows = (r.dH + 15) / 16;
    for (int t0 = 0; t0 < tileRows; t0 += 65535) {         // grid.y carries at most 65535 tiles
        const dim3 grid((r.dW + 15) / 16, tileRows - t0 < 65535 ? tileRows - t0 : 65535, 1);
#define AAI_SCAN_WIN(W)                                                                                                                        \
    case W:                                                                                                                                    \
        if (r.mode == AAI_MODE_FAST) hipLaunchKernelGGL((aai_quad_scan_kernel<W, false, true>), grid, dim3(kQuadBlock), 0, stream, r, q, laneMasks, counter, t0); \
        else if (q.hiPrec) hipLaunchKernelGGL((aai_quad_scan_kernel<W, true>), grid, dim3(kQuadBlock), 0, stream, r, q, laneMasks, counter, t0); \
        else hipLaunchKernelGGL((aai_quad_scan_kernel<W, false>), grid, dim3(kQuadBlock), 0, stream, r, q, laneMasks, counter, t0);            \
        break;
        switch (r.mode == AAI_MODE_FAST ? q.winFast : (q.win < 3 ? 3 : (q.win > 8 ? 8 : q.win))) {
            AAI_SCAN_WIN(2) AAI_SCAN_WIN(3) AAI_SCAN_WIN(4) AAI_SCAN_WIN(5) AAI_SCAN_WIN(6) AAI_SCAN_WIN(7) AAI_SCAN_WIN(8)
        }
#undef AAI_SCAN_WIN
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_tile_flags(const unsigned long long *laneMasks, unsigned tilesX, unsigned tilesY, unsigned *out, hipStream_t stream)
{
    const size_t tiles = (size_t)tilesX * tilesY;
    if (!tiles) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((tiles + 255) / 256, 65536);
    hipLaunchKernelGGL(aai_tile_flags_kernel, dim3(blocks), dim3(256), 0, stream, laneMasks, tilesX, tilesY, tile_flag_row_words(tilesX), out);
    return hipGetLastError();
}

hipError_t launch_flag_list(const unsigned long long *laneMasks, size_t waves, unsigned tilesX, void *list, unsigned *cursor, unsigned capacity,
                            hipStream_t stream)
{
    if (!waves) return hipSuccess;
    const size_t blocks = (waves + 255) / 256;
    hipLaunchKernelGGL(aai_flag_list_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, laneMasks, waves, tilesX,
                       static_cast<uint2 *>(list), cursor, capacity);
    return hipGetLastError();
}

#endif      // AAI_QUAD_PART 1

}  // namespace aai
