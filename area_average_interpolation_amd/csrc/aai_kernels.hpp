// aai_kernels.hpp -- launcher declarations shared between the host engine (aai_engine.cpp, aai_capi.cpp) and the HIP
// translation units.  Every launcher only enqueues work on `stream`.
#pragma once

#include <hip/hip_runtime.h>

#include "aai_plan.hpp"
#include "aai_rot_math.hpp"

#include <cstdlib>

namespace aai {

// Launch heuristics can be overridden from the environment for experiments (tools/*_ab.sh) -- only in a build made with
// `make EXTRA=-DAAI_EXPERIMENTS`: the shipping library never reads these variables.
inline const char *experiment_env(const char *name)
{
#if defined(AAI_EXPERIMENTS)
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

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
    // launch shape measured by the plan for this (geometry, device): output rows per workgroup (0 = built-in default),
    // nontemporal source loads, grid order (see aai_axis_kernel)
    int tuneRows, tuneNt, tuneSwap;
};
void set_axis_tune(const char *spec);      // experiments only (tools/tune_axis.py)
hipError_t launch_axis(const AxisLaunch &a, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, hipStream_t stream, const char **kernelName);

// ---- K2/K3/K4/K5: per-output-pixel kernels on the rotated lattice --------------------------------------
// Flagged dst pixels (geometry only), found once per plan: one 64-bit lane mask per wave of the 16x16-pixel tiling.
// launch_knife_scan marks the pixels with a knife edge of the reference's classifier, launch_quad_scan adds the
// pixels whose decisions the fp32 quad kernel must not take; both count the pixels they newly flag in counter[0].
// launch_flag_list turns the masks into the list of (dx, dy) the double-precision fix-up pass runs over.
struct RotFlags {
    const void *list = nullptr;        // device array of uint2 (dx, dy), `count` entries
    unsigned count = 0;
    bool dense = false;                // so many that the whole image takes the double-precision pass instead
    // With the lane masks kept, the quad kernel leaves the flagged pixels alone, so the fix-up pass no longer has to
    // FOLLOW it: it runs beside it on the plan's side stream (fork / join through two events), where its few,
    // latency-bound waves cost nothing instead of ~40 us behind the production pass.
    const unsigned long long *masks = nullptr;
    const unsigned *tileFlags = nullptr;   // one bit per 16 x 16 tile that holds a flagged pixel (launch_tile_flags), tileFlagWords words per tile row
    int tileFlagWords = 0;
    const int *live = nullptr;         // per 16-row tile row of the canvas: first and last 16-column tile that can hold a non-zero pixel (rotated_live_spans)
    int form = 0;                      // which fp32 formulation's scan produced the flags: ROT_FORM_QUAD or ROT_FORM_CELL (it serves the launch)
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
enum RotForm { ROT_FORM_QUAD = 0, ROT_FORM_CELL = 1 };
size_t rotated_flag_words(const RotLaunch &r);      // waves of the tiling = 64-bit words of the mask array
hipError_t launch_knife_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream);
// the double-precision fix-up pass over a list of dst pixels (defined in aai_rotated_strict.hip);
// pixelList == NULL: the whole image (grid as for the production pass, at most 65535 tile rows)
void launch_rotated_fixup(const RotLaunch &r, int batch, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                          const uint2 *pixelList, unsigned nList, hipStream_t stream);
// axis-aligned geometries (K1): dst pixels whose weights the separable model gets wrong (aai_axis_verify.hpp), same layout
hipError_t launch_axis_verify(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream);
hipError_t launch_quad_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream);
// one bit per 16 x 16 tile whose lane masks are not all zero: out[tileRow * rowWords + (tileX >> 5)] bit (tileX & 31); out zeroed by the caller
inline unsigned tile_flag_row_words(unsigned tilesX) { return (tilesX + 31u) / 32u + 1u; }
hipError_t launch_tile_flags(const unsigned long long *laneMasks, unsigned tilesX, unsigned tilesY, unsigned *out, hipStream_t stream);
hipError_t launch_flag_list(const unsigned long long *laneMasks, size_t waves, unsigned tilesX, void *list, unsigned *cursor, unsigned capacity,
                            hipStream_t stream);
hipError_t launch_rotated(const RotLaunch &r, const QuadMap &m, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                          int batch, const RotFlags &flags, hipStream_t stream, const char **kernelName);
bool quad_can_address(const RotLaunch &r, int srcType, ImageView sv);
// (which form of the kernel family a launch_quad call took, where it has several: a static string, or NULL; per thread)
void set_quad_kernel_note(const char *note);
const char *quad_kernel_note();
hipError_t launch_quad(const RotLaunch &r, const QuadMap &m, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, const unsigned long long *skipMasks, hipStream_t stream, const int *live = nullptr);

int quad_anchor_rows(const RotLaunch &r);      // images of 4 GiB and more: how many source rows apart two lanes of a wave can read

// wide footprints (aai_rotated_wide.hip): the quad formulation over a window split into parts, one lane per part
bool wide_can_serve(const RotLaunch &r, int srcType, ImageView sv);      // r.chan set; RotLaunch::wide, plain images, area mode
hipError_t launch_wide_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream);
hipError_t launch_wide(const RotLaunch &r, const QuadMap &m, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, const unsigned long long *skipMasks, hipStream_t stream);

// the cell formulation (aai_rotated_cell.hip): one lane per cell of the dst grid, every (dst, src) pair evaluated once
bool cell_can_serve(const RotLaunch &r, int srcType, ImageView sv);      // r.chan set; plain images below 4 GiB, area mode
hipError_t launch_cell_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream);
hipError_t launch_cell(const RotLaunch &r, const QuadMap &m, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                       int batch, const unsigned long long *skipMasks, hipStream_t stream);

// ---- utilities -----------------------------------------------------------------------------------------
hipError_t launch_synth(float *dst, int W, int H, int64_t stride, uint64_t seed, hipStream_t stream);
hipError_t launch_synth_rows(float *dst, int W, int H, int row0, int row1, int64_t stride, uint64_t seed, hipStream_t stream);
hipError_t launch_f64_to_f32(const double *src, float *dst, size_t n, hipStream_t stream);
hipError_t launch_f32_to_f64(const float *src, double *dst, size_t n, hipStream_t stream);

}  // namespace aai
