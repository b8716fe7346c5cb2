// aai_rotated_strict.hip -- the knife-edge fix-up pass of the rotated-lattice kernels.
//
// Compiled with -ffp-contract=off (see the Makefile): the strict replay (aai_strict.hpp) reproduces the
// reference's DBL_EPSILON decisions only if every product and sum rounds exactly like the reference build's.
// It runs over the plan's list of flagged dst pixels (see aai_rotated_kernel.hpp), one lane per list entry, or over
// the whole image when the list would be longer than the plan keeps.
#include "aai_rotated_kernel.hpp"
#include "aai_axis_verify.hpp"

namespace aai {

// Plan-time scan of an axis-aligned geometry (K1): one bit per dst pixel whose weights the separable model gets wrong
// (aai_axis_verify.hpp), in the 16 x 16 tiling and mask layout of aai_knife_scan_kernel; counter[0] counts them.
template <bool FAST>
__global__ __launch_bounds__(kRotBlock) void aai_axis_verify_kernel(RotLaunch r, unsigned long long *__restrict__ laneMasks, unsigned *__restrict__ counter, int tileRow0)
{
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dx = blockIdx.x * 16 + (tid & 15);
    const int dy = (tileRow0 + blockIdx.y) * 16 + (tid >> 4);
    const bool differs = dx < r.dW && dy < r.dH && (FAST ? axis_pixel_differs_fast(r, dx, dy) : axis_pixel_differs(r, dx, dy));
    const unsigned long long any = __ballot(differs);
    if ((tid & 63) == 0) {
        laneMasks[((size_t)(tileRow0 + blockIdx.y) * gridDim.x + blockIdx.x) * (kRotBlock / 64) + wave] = any;
        if (any != 0ull) atomicAdd(counter, (unsigned)__popcll(any));
    }
}

hipError_t launch_axis_verify(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream)
{
    if (r.dW <= 0 || r.dH <= 0) return hipSuccess;
    const int tileRows = (r.dH + 15) / 16;
    for (int t0 = 0; t0 < tileRows; t0 += 65535) {         // grid.y carries at most 65535 tiles
        const dim3 grid((r.dW + 15) / 16, tileRows - t0 < 65535 ? tileRows - t0 : 65535, 1);
        if (r.mode == AAI_MODE_FAST) hipLaunchKernelGGL(aai_axis_verify_kernel<true>, grid, dim3(kRotBlock), 0, stream, r, laneMasks, counter, t0);
        else hipLaunchKernelGGL(aai_axis_verify_kernel<false>, grid, dim3(kRotBlock), 0, stream, r, laneMasks, counter, t0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The fix-up pass over the plan's LIST of flagged dst pixels, 16 lanes per pixel.  aai_rotated_kernel<STRICT> gives every
// listed pixel ONE lane, which then walks its whole window alone -- 100 pairs of double-precision replay at 5.9 : 1: a
// latency-bound pass of 40-120 us however few pixels are listed, which set the time of every small rotated request (the
// reference's own example call: 72 us, config 5 at 1/8 scale: 125 us) and shared the chip with the production kernel on larger
// ones.  Here the 16 lanes of a group take the window's pairs in turn (the same per-pair code: classify_pair, the closed
// forms, the strict replay of the reference's classifier at knife edges) and their partial sums are added in a four-step
// butterfly; four groups per wave.  Sums are reassociated (double precision: ~1e-16 relative).
constexpr int kFixGroup = 16;

template <int MODE, typename T, bool MULTI>
__global__ __launch_bounds__(kRotBlock) void aai_fixup_group_kernel(RotLaunch r, const T *__restrict__ src, ImageView sv, float *__restrict__ dst, ImageView dv,
                                                                   const uint2 *__restrict__ pixelList, unsigned nList)
{
    constexpr int NC = MULTI ? kMaxChan : 1;
    const int tid = threadIdx.x, sub = tid & (kFixGroup - 1);
    const unsigned e = blockIdx.x * (kRotBlock / kFixGroup) + (unsigned)(tid / kFixGroup);
    bool valid = e < nList;
    const uint2 p = valid ? pixelList[e] : make_uint2(0u, 0u);
    const int dx = (int)p.x, dy = (int)p.y;
    valid = valid && dy >= r.dyBase && dy < r.dyEnd;                 // (uniform within a group)
    const T *img = src + (int64_t)blockIdx.z * sv.imageStride;
    const int chan = MULTI ? r.chan : 1;

    double sumA = 0.0, acc[NC] = {};
    int count = 0;
    if (valid) {
        double px, py;
        pixel_centre(r, dx, dy, px, py);
        int x0, x1, y0, y1;
        rot_window(r, px, py, x0, x1, y0, y1);
        const int nW = x1 - x0 + 1;
        SVec sv4[4];
        bool haveVertices = false;
        const double lim = r.h + DBL_EPSILON * r.side;
        // pair k of the window (row-major) belongs to lane k mod 16
        int X = x0 + sub, Y = y0;
        while (X > x1 && Y <= y1) { X -= nW; ++Y; }
        while (nW > 0 && Y <= y1) {
            const double ex = X - px, ey = Y - py;
            double w = 0.0;
            if (MODE == AAI_MODE_FAST) {
                const double a = fabs(ex * r.c - ey * r.s), b = fabs(ex * r.s + ey * r.c);
                bool in = a <= lim && b <= lim;
                const bool edgy = (fabs(a - r.h) < AAI_KNIFE_GUARD && b <= r.h + AAI_KNIFE_GUARD) || (fabs(b - r.h) < AAI_KNIFE_GUARD && a <= r.h + AAI_KNIFE_GUARD);
                if (edgy) {                      // a centre on an edge: the reference's ray cast decides
                    if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                    SVec pc; pc.x = X; pc.y = Y;
                    in = strict_centre_inside(pc, sv4);
                }
                if (in) { w = 1.0; ++count; }
            } else {
                const double a = ex * r.c - ey * r.s, b = ex * r.s + ey * r.c;
                double d = 0.0;
                bool edgy = false, edgy2 = false;
                const int cls = classify_pair<true>(r, a, b, d, edgy);
                if (cls != PAIR_OUTSIDE) {
                    if (cls == PAIR_INSIDE) w = 1.0;
                    else if (cls == PAIR_GENERAL) w = wedge_pair_area<true>(r, px - (X - 0.5), py - (Y - 0.5), a < 0.0, b < 0.0, r.policy, edgy2);
                    else w = single_cut_area<true>(r, d, cls == PAIR_CUT_LR, r.policy, edgy2);
                    if (edgy || edgy2) {
                        if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                        w = strict_pair_area(sv4, X, Y, r.policy);
                    }
                }
            }
            if (w != 0.0) {
                sumA += w;
                const T *q = img + virt_offset(r, X, Y, sv.rowStride, chan);
                if (!MULTI) acc[0] += w * (double)q[0];
                else {
                    float v[kMaxChan];
                    load_pixel(q, chan, v);
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (c < chan) acc[c] += w * (double)v[c];
                }
            }
            X += kFixGroup;
            while (X > x1 && Y <= y1) { X -= nW; ++Y; }
        }
    }
    // the group's sums (every lane of the wave takes part; groups of invalid entries add zeros)
#pragma unroll
    for (int off = kFixGroup / 2; off > 0; off >>= 1) {
        sumA += __shfl_xor(sumA, off, kFixGroup);
        count += __shfl_xor(count, off, kFixGroup);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] += __shfl_xor(acc[c], off, kFixGroup);
    }
    if (valid && sub == 0) {
        float *out = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride + (int64_t)dx * chan;
        const bool any = MODE == AAI_MODE_FAST ? count > 0 : DBL_EPSILON < fabs(sumA);           // Source.cpp:905 / 577
        const double denom = MODE == AAI_MODE_FAST ? (double)count : sumA;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < chan) out[c] = any ? (float)(acc[c] / denom) : 0.f;
    }
}

template <typename T>
static void fixup_typed(const RotLaunch &r, int batch, const T *src, ImageView sv, float *dst, ImageView dv,
                        const uint2 *waveFlags, unsigned nList, hipStream_t stream)
{
    if (waveFlags) {
        // the plan's list: 16 lanes per listed pixel
        const dim3 groups((nList + kRotBlock / kFixGroup - 1) / (kRotBlock / kFixGroup), 1, batch);
        if (r.chan > 1) {
            if (r.mode == AAI_MODE_FAST) hipLaunchKernelGGL((aai_fixup_group_kernel<AAI_MODE_FAST, T, true>), groups, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
            else hipLaunchKernelGGL((aai_fixup_group_kernel<AAI_MODE_AREA, T, true>), groups, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
        } else {
            if (r.mode == AAI_MODE_FAST) hipLaunchKernelGGL((aai_fixup_group_kernel<AAI_MODE_FAST, T, false>), groups, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
            else hipLaunchKernelGGL((aai_fixup_group_kernel<AAI_MODE_AREA, T, false>), groups, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
        }
        return;
    }
    // no list: so many pixels are flagged that the whole image takes the strict pass, one lane per dst pixel
    const dim3 grid((r.dW + 15) / 16, (r.dyEnd - r.dyBase + 15) / 16, batch);
    if (r.chan > 1) {      // interleaved channels
        if (r.mode == AAI_MODE_FAST)
            hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_FAST, true, T, true>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
        else
            hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_AREA, true, T, true>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
        return;
    }
    if (r.mode == AAI_MODE_FAST)
        hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_FAST, true, T>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
    else
        hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_AREA, true, T>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags, nList);
}

void launch_rotated_fixup(const RotLaunch &r, int batch, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                          const uint2 *waveFlags, unsigned nList, hipStream_t stream)
{
    if ((waveFlags && !nList) || batch <= 0 || r.dW <= 0 || r.dyEnd <= r.dyBase) return;
    switch (srcType) {
    case SRC_U8: fixup_typed(r, batch, static_cast<const unsigned char *>(src), sv, dst, dv, waveFlags, nList, stream); break;
    case SRC_U16: fixup_typed(r, batch, static_cast<const unsigned short *>(src), sv, dst, dv, waveFlags, nList, stream); break;
    default: fixup_typed(r, batch, static_cast<const float *>(src), sv, dst, dv, waveFlags, nList, stream); break;
    }
}

}  // namespace aai
