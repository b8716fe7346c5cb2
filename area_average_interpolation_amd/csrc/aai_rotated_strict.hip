// aai_rotated_strict.hip -- the knife-edge fix-up pass of the rotated-lattice kernels.
//
// Compiled with -ffp-contract=off (see the Makefile): the strict replay (aai_strict.hpp) reproduces the
// reference's DBL_EPSILON decisions only if every product and sum rounds exactly like the reference build's.
// It runs over the plan's list of flagged dst pixels (see aai_rotated_kernel.hpp), one lane per list entry, or over
// the whole image when the list would be longer than the plan keeps.
#include "aai_rotated_kernel.hpp"

namespace aai {

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
