// aai_rotated_strict.hip -- the knife-edge fix-up pass of the rotated-lattice kernels.
//
// Compiled with -ffp-contract=off (see the Makefile): the strict replay (aai_strict.hpp) reproduces the
// reference's DBL_EPSILON decisions only if every product and sum rounds exactly like the reference build's.
// Waves whose flag the production pass left clear return immediately, so in generic geometry this launch
// costs a few microseconds.
#include "aai_rotated_kernel.hpp"

namespace aai {

template <typename T>
static void fixup_typed(const RotLaunch &r, dim3 grid, const T *src, ImageView sv, float *dst, ImageView dv,
                        const unsigned *waveFlags, hipStream_t stream)
{
    if (r.chan > 1) {      // interleaved channels
        if (r.mode == AAI_MODE_FAST)
            hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_FAST, true, T, true>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags);
        else
            hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_AREA, true, T, true>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags);
        return;
    }
    if (r.mode == AAI_MODE_FAST)
        hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_FAST, true, T>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags);
    else
        hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_AREA, true, T>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, waveFlags);
}

void launch_rotated_fixup(const RotLaunch &r, dim3 grid, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                          const unsigned *waveFlags, hipStream_t stream)
{
    switch (srcType) {
    case SRC_U8: fixup_typed(r, grid, static_cast<const unsigned char *>(src), sv, dst, dv, waveFlags, stream); break;
    case SRC_U16: fixup_typed(r, grid, static_cast<const unsigned short *>(src), sv, dst, dv, waveFlags, stream); break;
    default: fixup_typed(r, grid, static_cast<const float *>(src), sv, dst, dv, waveFlags, stream); break;
    }
}

}  // namespace aai
