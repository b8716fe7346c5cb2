// aai_plan.hpp -- host-side geometry ("plan") for one resampling request, and the argument blocks the
// device kernels consume.  Host code is plain C++17 (no HIP types) so that validation and layout
// queries work on machines without a GPU.
//
// Geometry follows SURVEY.md Appendix A, i.e. Source.cpp:112-305 of the reference, evaluated in double
// precision.  Nothing the reference materialises per pixel (modSrc, dstPos, the edge-line tables) is
// stored: they are affine in the pixel indices and are recomputed on the fly.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/aai.h"
#include "aai_rot_math.hpp"

namespace aai {

// ---- geometry --------------------------------------------------------------------------------------
struct Geometry {
    int W = 0, H = 0;            // input image
    int scale = 1;               // Source.cpp:139
    int quadrant = 0;            // Source.cpp:140-146
    double angle = 0;            // reduced angle, degrees, [0,90)
    double sn = 0, cs = 1;       // Source.cpp:147-148
    int mW = 0, mH = 0;          // virtual (scaled, pre-rotated) source size, Source.cpp:150-156
    double isoX = 0, isoY = 0;   // isocenter on the virtual lattice, Source.cpp:173-174
    double ratio = 1, side = 1;  // expansionRatio, dstSideLength, Source.cpp:177-178
    int dW = 0, dH = 0;          // Source.cpp:179-180
    double dIsoX = 0, dIsoY = 0; // Source.cpp:185-186
    double fracX = 0, fracY = 0; // Source.cpp:183-184
    double offX = 0, offY = 0;   // Source.cpp:187-200
    bool lt45 = true;            // Source.cpp:230
    double tsn = 0, tcs = 1, ttn = 0;   // Source.cpp:229-240 (ttn snapped to 0 below DBL_EPSILON)
    bool axisAligned = true;     // ttn == 0 with lt45: dst pixel edges parallel to the source axes
};

// Validates like Source.cpp:112-132 (+ finiteness and size limits) and fills g.  Returns AAI_OK or an
// AAI_ERR_* code with the message in `msg`.
int make_geometry(const aai_request &rq, Geometry &g, std::string &msg);

// Centre of dst pixel (dx,dy) on the virtual lattice, Source.cpp:212-219.
inline void dst_centre(const Geometry &g, double dx, double dy, double &px, double &py)
{
    const double u = (dx + g.fracX) * g.side - g.isoX + g.offX;
    const double v = (dy + g.fracY) * g.side - g.isoY + g.offY;
    px = u * g.cs + v * g.sn + g.isoX;
    py = -u * g.sn + v * g.cs + g.isoY;
}

// Interior width (virtual pixels, L - cos - sin) from which the area kernel walks rows as runs.
constexpr double kRunsMinInterior = 3.25;

// Fills the uniform block of the per-output-pixel kernels (K2-K5) from the geometry.
RotLaunch make_rot_launch(const Geometry &g, int mode, int policy);

// Virtual pixel (X, Y) -> element of the source image for the quad kernel (Source.cpp:164-167: the pre-rotation only
// flips / swaps axes, and (m - 1 - X) div scale = m / scale - 1 - X div scale because m is a multiple of scale):
//   element = base + uX * strideX + uY * strideY,  uX = flipX ? nX - 1 - X div scale : X div scale  (uY alike)
// with non-negative strides, so that a lane addresses its pixels with unsigned 32-bit byte offsets from one uniform
// base.  rowStride in elements; srcRow0 = the source row the image pointer addresses (row bands).
struct QuadMap {
    int64_t base;                // elements; negative for a band (base = -srcRow0 * rowStride)
    int64_t strideX, strideY;    // elements per step of uX / uY: 1 or rowStride
    int nX, nY;                  // source pixels along virtual X / Y (mW / scale, mH / scale)
    int flipX, flipY;
    int scale;
    float invScale;
    double invScaleD;
    // interleaved 8- / 16-bit pixels are fetched with one 4- / 8-byte load each, which may reach past the pixel: loads start
    // no later than lastLoad4 / lastLoad8 (byte offsets of the last 4 / 8 bytes of the image) and shift the rest away
    uint32_t lastLoad4, lastLoad8;
    int anchorRows;              // set by the launcher for plain images of 4 GiB and more: offsets are relative to an anchor row per wave (QuadSrc::issue)    // set by the launcher when the kernel skips the pixels the plan's scans flagged: one bit per 16 x 16 dst tile, set where the tile holds
    // a flagged pixel (tileFlagWords 32-bit words per tile row, one more than the tiles need: a wave reads two words at once), so that
    // a wave of the cell kernel asks for the per-pixel masks only where there is something to find; NULL = no summary
    const unsigned *tileFlags;
    int tileFlagWords;
    // Plain fp32 images without replication, a window wholly inside the lattice (QuadSrc::issue's common case): the byte offset of
    // the first element of the window's first line IN MEMORY ORDER is fastC0 + X fastSX + Y fastSY - (WIN - 1) fastRev4 (mod 2^32,
    // (X, Y) = virtual pixel of window position (0, 0)), its lines follow fastLine bytes apart.  fastOk: the fields are valid (lattice
    // below 2^23 pixels a side, so that a 24-bit multiply serves the contiguous axis).
    // (last in the struct: kernels that never read them do not pull them into scalar registers with their neighbours)
    uint32_t fastC0, fastSX, fastSY, fastLine, fastRev4;
    int fastAlongX, fastOk;      // fastAlongX: the contiguous axis is virtual X (quadrants 0 / 2)
    // every index x byte-stride product of the map fits a 24-bit multiply (v_mul_i32_i24, full rate, where the 32-bit multiply runs at a
    // quarter): fewer than 2^23 source pixels a side and a row pitch below 8 MiB.  fastOk implies it.
    int mul24Ok;
    // set by the cell kernel's launcher for plain images of 4 GiB and more: every wave moves its base pointer to the first source row its
    // cells can touch and takes that row's byte offset (mod 2^32) off its 32-bit lane offsets (QuadSrc::rebase, aai_rotated_cell.hip)
    int rebaseWaves;
};
QuadMap make_quad_map(const Geometry &g, int64_t rowStride, int srcRow0, int channels = 1, int elementBytes = 4);      // channels: elements per pixel (interleaved)

// ---- K1: separable axis-aligned tables ---------------------------------------------------------------
// One entry per output index along one axis: the source window [s0,s1] along the matching SOURCE axis
// (original-image indices, ascending) and the three distinct weights a box footprint can produce.
// Weights are already divided by the window's total weight, so the kernel never normalises.
struct alignas(16) AxisEntry {
    int32_t s0, s1;      // inclusive source index range; s0 > s1 never happens (empty => weights 0)
    float wFirst;        // weight of s0
    float wMid;          // weight of every index strictly between s0 and s1
    float wLast;         // weight of s1 (unused when s0 == s1)
    int32_t pad[3];
};
static_assert(sizeof(AxisEntry) == 32, "AxisEntry layout");

// A wave strip: output indices [k0,k1) along the lane axis whose windows all lie in [x0, x0+STRIP_COLS).
struct alignas(16) AxisStrip { int32_t k0, k1, x0, pad; };

constexpr int STRIP_COLS = 256;   // 64 lanes x float4

struct AxisTables {
    std::vector<AxisEntry> lane;     // along source x (the coalesced, lane-mapped axis): nA entries
    std::vector<AxisEntry> row;      // along source y: nB entries
    std::vector<AxisStrip> strips;   // partition of the lane axis
    int nA = 0, nB = 0;
    // output element for (ka,kb) = base + ka*strideA + kb*strideB   (in elements of one dst image, with
    // the dst row stride folded in by the launcher)
    bool transposed = false;         // lane axis runs along dst y (quadrants 1 and 3)
    bool flipA = false, flipB = false;
    bool wide = false;               // some window is wider than a strip -> per-pixel fallback kernel
    int maxRowSpan = 0;              // largest s1-s0+1 over the row table
    bool rowsShared = false;         // consecutive output rows read a common source row (windows not pixel-aligned)
    int maxOutputsPerStrip = 0;
    int channels = 1;                // interleaved channels: the lane table has nA = pixels * channels entries, taps `channels` apart
};

// mode: AAI_MODE_AREA (overlap lengths) or AAI_MODE_FAST (centre counts).  Only for g.axisAligned.
void build_axis_tables(const Geometry &g, int mode, AxisTables &t, int channels = 1);
void restrict_axis_tables_to_band(const Geometry &g, AxisTables &t, int row0, int row1, int &srcRow0, int &srcRow1, int extraRows = 0);

// K1's separable model checked on the host, one representative per (column class, row class), where the geometry's
// arithmetic is exact (aai_plan.cpp); false = does not qualify, run the device scan.  flagged: (dx, dy) of the dst pixels
// the fix-up pass must recompute; dense: more than maxListed of them.
bool axis_verify_by_class(const RotLaunch &r, std::vector<std::pair<int, int>> &flagged, bool &dense, unsigned maxListed);

// Source rows [srcRow0, srcRow1) that dst rows [row0,row1) of a rotated-lattice request can touch (conservative).
void rotated_band_source_rows(const Geometry &g, int row0, int row1, bool sampler, int &srcRow0, int &srcRow1);
// Live span of every 16-row tile row of the dst canvas, in 16-column tiles: spans[2 t] ... spans[2 t + 1] (first > last: none).
// Tiles outside it hold only pixels that are 0 in every mode (rot_live_cols); empty when the geometry has no dead tiles to speak of.
void rotated_live_spans(const RotLaunch &r, bool sampler, std::vector<int> &spans);

}  // namespace aai
