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

template <typename T>
static void fixup_typed(const RotLaunch &r, int batch, const T *src, ImageView sv, float *dst, ImageView dv,
                        const uint2 *waveFlags, unsigned nList, hipStream_t stream)
{
    const dim3 grid = waveFlags ? dim3((nList + kRotBlock - 1) / kRotBlock, 1, batch)
                                : dim3((r.dW + 15) / 16, (r.dyEnd - r.dyBase + 15) / 16, batch);
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
