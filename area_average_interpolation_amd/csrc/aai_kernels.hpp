// aai_kernels.hpp -- launcher declarations shared between the C ABI (aai_capi.cpp) and the HIP
// translation units.  Every launcher only enqueues work on `stream`.
#pragma once

#include <hip/hip_runtime.h>

#include "aai_plan.hpp"
#include "aai_rot_math.hpp"

namespace aai {

// Image addressing shared by every kernel: element (x,y) of image b is base[b*imageStride + y*rowStride + x].
// Source element types (AAI_DTYPE_* in include/aai.h).  Outputs are always fp32.
enum SrcType { SRC_F32 = 0, SRC_U8 = 1, SRC_U16 = 2 };

struct ImageView {
    int64_t rowStride;
    int64_t imageStride;
};

// ---- K1: axis-aligned separable kernel ----------------------------------------------------------------
struct AxisLaunch {
    const AxisEntry *laneTab;   // device, nA entries (ascending source x)
    const AxisEntry *rowTab;    // device, nB entries (ascending source y)
    const AxisStrip *strips;    // device, nStrips entries
    int nA, nB, nStrips;
    int srcW, srcH;
    int64_t outBase, outStrideA, outStrideB;   // dst element = outBase + ka*outStrideA + kb*outStrideB
    int wide;
    int maxRowSpan;             // largest number of source rows any output row needs
    int rowsShared;             // consecutive output rows share a source row
    int maxOutputsPerStrip;
    // interleaved channels (1 = none): lane entries are (pixel, channel) pairs over the source row's ELEMENTS; the taps of
    // one entry are tapStep elements apart; dst element = outBase + (ka / outChan)*outStrideA + ka % outChan + kb*outStrideB
    // (outChan = 1 when the lane order is already the dst element order: not transposed, not flipped)
    int tapStep, outChan;
    int transposed;             // the lane axis runs along dst y (quadrants 1 and 3)
};
hipError_t launch_axis(const AxisLaunch &a, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, hipStream_t stream, const char **kernelName);

// ---- K2/K3/K4/K5: per-output-pixel kernels on the rotated lattice --------------------------------------
// Knife-edge flags (one 32-bit word per wave of the 16x16-pixel tiling, geometry only): launch_knife_scan fills
// them and counts the flagged waves in counter[0]; launch_rotated runs the fix-up pass iff waveFlags != NULL.
size_t rotated_flag_words(const RotLaunch &r);
hipError_t launch_knife_scan(const RotLaunch &r, unsigned *waveFlags, unsigned *counter, hipStream_t stream);
hipError_t launch_rotated(const RotLaunch &r, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                          int batch, const unsigned *waveFlags, hipStream_t stream, const char **kernelName);

// ---- utilities -----------------------------------------------------------------------------------------
hipError_t launch_synth(float *dst, int W, int H, int64_t stride, uint64_t seed, hipStream_t stream);
hipError_t launch_f64_to_f32(const double *src, float *dst, size_t n, hipStream_t stream);
hipError_t launch_f32_to_f64(const float *src, double *dst, size_t n, hipStream_t stream);

}  // namespace aai
