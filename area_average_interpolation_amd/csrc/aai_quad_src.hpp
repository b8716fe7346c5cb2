// aai_quad_src.hpp -- the staged source window of the fp32 per-pixel kernels (device code only): shared by
// aai_rotated_quad.hip (one lane per dst pixel) and aai_rotated_cell.hip (one lane per cell of the dst grid).
#pragma once

#include "aai_kernels.hpp"
#include "aai_rot_quad.hpp"

namespace aai {

constexpr int kQuadBlock = 256;      // lanes per workgroup: the LDS window is [slot][kQuadBlock], one column per lane

// N consecutive fp32 elements from an element-aligned address in as few load instructions as possible
template <int N>
__device__ __forceinline__ void load_line(const float *p, float (&seg)[N])
{
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    int at = 0;
    if (N >= 4) { const f4u q = *reinterpret_cast<const f4u *>(p); seg[0] = q.x; seg[1] = q.y; seg[2] = q.z; seg[3] = q.w; at = 4; }
    if (N - at == 4) { const f4u q = *reinterpret_cast<const f4u *>(p + at); seg[at] = q.x; seg[at + 1] = q.y; seg[at + 2] = q.z; seg[at + 3] = q.w; }
    else if (N - at == 3) { const f3u q = *reinterpret_cast<const f3u *>(p + at); seg[at] = q.x; seg[at + 1] = q.y; seg[at + 2] = q.z; }
    else if (N - at == 2) { const f2u q = *reinterpret_cast<const f2u *>(p + at); seg[at] = q.x; seg[at + 1] = q.y; }
    else if (N - at == 1) seg[at] = p[at];
}

// does any of the 16 x 16 tiles [tx0, tx0 + n) (n <= 32) of tile row `tileRow` hold a flagged pixel?  Wave-uniform arguments: scalar loads
__device__ __forceinline__ bool tiles_flagged(const unsigned *__restrict__ tileFlags, int rowWords, int tileRow, int tx0, int n)
{
    const unsigned *p = tileFlags + (size_t)tileRow * rowWords + (tx0 >> 5);
    const unsigned long long both = ((unsigned long long)p[1] << 32) | p[0];
    return ((both >> (tx0 & 31)) & ((1ull << n) - 1ull)) != 0;
}

// The staged window of one lane.  Offsets are unsigned bytes from the image's first element (QuadMap: non-negative
// strides), one table per axis; positions outside the lattice are clamped onto it -- their values are fetched but
// never read (quad_pixel only reads slots whose valid bit is set).
// STAGED: the caller reads values through commit() / at() only (not reg()): vector-loaded lines may stay in memory order
template <typename T, int WIN, bool SCALED, bool STAGED = false>
struct QuadSrc {
    const char *img;                 // first element of this image (band offset included)
    const QuadMap *m;
    int mW, mH;
    float (*lds)[kQuadBlock];        // [WIN * WIN][kQuadBlock]
    int tid;
    T v[WIN * WIN];
    int arranged;                    // STAGED: 0 = v[] is in slot order; else 1 + arrangement of the vector-loaded lines (commit)

    // allInside: the caller has established (by a wave vote) that every lane's window lies inside the lattice
    __device__ __forceinline__ void issue(int xg0, int yg0, unsigned long long, bool allInside = false)
    {
        unsigned colOff[WIN], rowOff[WIN];            // source indices along virtual X / Y first, byte offsets below
        const unsigned sxb = (unsigned)m->strideX * (unsigned)sizeof(T), syb = (unsigned)m->strideY * (unsigned)sizeof(T);
        arranged = 0;
        if (!SCALED && STAGED && allInside && sizeof(T) == 4 && m->anchorRows == 0 && m->fastOk) {
            // The common case of the staged kernels, spelt out: every window of the wave inside the lattice, so no clamps, and the
            // WIN lines (the window axis that is contiguous in memory: virtual X in quadrants 0 / 2, virtual Y in 1 / 3) start a
            // wave-uniform number of bytes apart -- ONE per-lane byte offset from the host-composed coefficients of the map
            // (QuadMap::fastC0 ...: a multiply along the strided axis, a 24-bit multiply-add along the contiguous one) and one add per
            // further line, instead of 2 WIN clamped indices, flips, multiplies and adds per lane
            const uint32_t across = (uint32_t)(m->fastAlongX ? yg0 : xg0) * (m->fastAlongX ? m->fastSY : m->fastSX);
            const int alongStep = (int)(m->fastAlongX ? m->fastSX : m->fastSY);                            // +4 or -4
            uint32_t line = (uint32_t)(__mul24(m->fastAlongX ? xg0 : yg0, alongStep) + (int)(across + (m->fastC0 - (uint32_t)(WIN - 1) * m->fastRev4)));
            arranged = 1 + (m->fastAlongX ? 0 : 2) + (m->fastRev4 ? 1 : 0);
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                float seg[WIN];
                load_line<WIN>(reinterpret_cast<const float *>(img + line), seg);
                line += m->fastLine;
#pragma unroll
                for (int e = 0; e < WIN; ++e) v[k * WIN + e] = (T)seg[e];
            }
            return;
        }
        if (SCALED && allInside && WIN - 1 <= m->scale && m->anchorRows == 0) {
            // Replicated pixels, window no wider than a source pixel plus one: it spans at most two source columns and two
            // source rows -- four loads, and every position selects its value by which side of the split it lies on
            // (x >= 0 here, so (x + 0.5) / scale is at least 0.5 / scale away from an integer: the floor is exact -- in fp32 too,
            // because the planner sends replicated lattices of 2^22 pixels a side and more to the double-precision kernels
            // (make_rot_launch): the quotient's rounding error, 2^-23 x / scale, stays below 0.5 / scale)
            const int scale = m->scale;
            const int qx0 = (int)(((float)xg0 + 0.5f) * m->invScale), qy0 = (int)(((float)yg0 + 0.5f) * m->invScale);
            const int splitX = scale - (xg0 - __mul24(qx0, scale)), splitY = scale - (yg0 - __mul24(qy0, scale));      // in [1, scale]: first column / row of the second source pixel
            const int ux0 = m->flipX ? m->nX - 1 - qx0 : qx0, uy0 = m->flipY ? m->nY - 1 - qy0 : qy0;
            const unsigned c0 = (unsigned)ux0 * sxb, r0 = (unsigned)uy0 * syb;
            // the second source pixel, where the window reaches it, is one step along the (possibly flipped) axis (never fetched from
            // outside the image)
            const unsigned c1 = splitX < WIN ? (m->flipX ? c0 - sxb : c0 + sxb) : c0, r1 = splitY < WIN ? (m->flipY ? r0 - syb : r0 + syb) : r0;
            const T v00 = *reinterpret_cast<const T *>(img + (c0 + r0)), v10 = *reinterpret_cast<const T *>(img + (c1 + r0));
            const T v01 = *reinterpret_cast<const T *>(img + (c0 + r1)), v11 = *reinterpret_cast<const T *>(img + (c1 + r1));
            T top[WIN], bottom[WIN];
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                top[i] = i >= splitX ? v10 : v00;
                bottom[i] = i >= splitX ? v11 : v01;
            }
#pragma unroll
            for (int j = 0; j < WIN; ++j)
#pragma unroll
                for (int i = 0; i < WIN; ++i) v[j * WIN + i] = j >= splitY ? bottom[i] : top[i];
            return;
        }
        if (!SCALED && allInside) {
            const int bx = m->flipX ? m->nX - 1 - xg0 : xg0, by = m->flipY ? m->nY - 1 - yg0 : yg0;
            const int sx = m->flipX ? -1 : 1, sy = m->flipY ? -1 : 1;
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                colOff[i] = (unsigned)(bx + sx * i);
                rowOff[i] = (unsigned)(by + sy * i);
            }
        } else if (!SCALED) {
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int X = min(max(xg0 + i, 0), mW - 1), Y = min(max(yg0 + i, 0), mH - 1);
                colOff[i] = (unsigned)(m->flipX ? m->nX - 1 - X : X);
                rowOff[i] = (unsigned)(m->flipY ? m->nY - 1 - Y : Y);
            }
        } else {
            // floor division of coordinates that are >= -8 (the window meets the lattice): exact, (n + 0.5) / scale is
            // at least 0.5 / scale away from an integer; inside the window (rem + i + 0.5) / scale with rem + i <
            // scale + 8 is far from every integer compared with fp32 rounding
            const int scale = m->scale;
            const int tx = xg0 + 8 * scale, ty = yg0 + 8 * scale;
            const int qx0 = (int)(((double)tx + 0.5) * m->invScaleD), qy0 = (int)(((double)ty + 0.5) * m->invScaleD);
            const float remX = (float)(tx - qx0 * scale) + 0.5f, remY = (float)(ty - qy0 * scale) + 0.5f;
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int ix = min(max(xg0 + i, 0), mW - 1) - xg0, iy = min(max(yg0 + i, 0), mH - 1) - yg0;
                const int qx = qx0 - 8 + (int)((remX + (float)ix) * m->invScale), qy = qy0 - 8 + (int)((remY + (float)iy) * m->invScale);
                colOff[i] = (unsigned)(m->flipX ? m->nX - 1 - qx : qx);
                rowOff[i] = (unsigned)(m->flipY ? m->nY - 1 - qy : qy);
            }
        }
        // Images of 4 GiB and more: offsets are taken from an anchor row of this WAVE instead of the image's first row --
        // every lane of a 16 x 4 dst tile reads within anchorRows source rows of any other lane (QuadMap::anchorRows, 0
        // for smaller images), so the anchor is one lane's first row less that bound and the base pointer moves with it
        if (m->anchorRows) {
            const bool rowsAlongX = sxb > syb;           // wave-uniform: which window axis walks the source rows
            const int first = __builtin_amdgcn_readfirstlane((int)(rowsAlongX ? colOff[0] : rowOff[0]));
            const unsigned anchor = (unsigned)max(first - m->anchorRows, 0);
            img += (int64_t)anchor * (int64_t)(rowsAlongX ? sxb : syb);
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                if (rowsAlongX) colOff[i] -= anchor; else rowOff[i] -= anchor;
            }
        }
#pragma unroll
        for (int i = 0; i < WIN; ++i) { colOff[i] *= sxb; rowOff[i] *= syb; }
        // Without replication one axis of the window is contiguous in memory (virtual X along source x in quadrants 0 / 2,
        // virtual Y in 1 / 3): fetch each of its WIN lines with one or two vector loads instead of WIN scalar ones -- lanes
        // are L source pixels apart, so every load instruction touches a dozen cache lines and their NUMBER is what the
        // texture path charges for.  Only where no lane of the wave has a clamped (off-image) column or row.
        if (!SCALED && sizeof(T) == 4 && (allInside || __all(xg0 >= 0 && xg0 + WIN <= mW && yg0 >= 0 && yg0 + WIN <= mH))) {
            // (every condition below is wave-uniform: the map is a kernel argument)
            const bool alongX = sxb == (unsigned)sizeof(T);
            const bool rev = alongX ? m->flipX != 0 : m->flipY != 0;      // the window axis runs against memory
            const unsigned first0 = alongX ? (rev ? colOff[WIN - 1] : colOff[0]) : (rev ? rowOff[WIN - 1] : rowOff[0]);
            if (STAGED) {
                // keep the lines as they lie in memory; commit() parks every element in its slot (static LDS offsets per
                // arrangement: no per-element selects, nothing to merge back into registers)
                arranged = 1 + (alongX ? 0 : 2) + (rev ? 1 : 0);
#pragma unroll
                for (int k = 0; k < WIN; ++k) {
                    float seg[WIN];
                    load_line<WIN>(reinterpret_cast<const float *>(img + ((alongX ? rowOff[k] : colOff[k]) + first0)), seg);
#pragma unroll
                    for (int e = 0; e < WIN; ++e) v[k * WIN + e] = (T)seg[e];
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                // line k: fixed row (alongX) or fixed column, its WIN elements ascending in memory from `first`; the values
                // stay in the registers they arrive in (reg()), so the arrangement is a per-element select here
                const unsigned first = alongX ? rowOff[k] + min(colOff[0], colOff[WIN - 1]) : colOff[k] + min(rowOff[0], rowOff[WIN - 1]);
                const bool back = alongX ? colOff[0] > colOff[WIN - 1] : rowOff[0] > rowOff[WIN - 1];
                float seg[WIN];
                load_line<WIN>(reinterpret_cast<const float *>(img + first), seg);
#pragma unroll
                for (int e = 0; e < WIN; ++e) {
                    const float val = back ? seg[WIN - 1 - e] : seg[e];
                    if (alongX) v[k * WIN + e] = (T)val; else v[e * WIN + k] = (T)val;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < WIN; ++j)
#pragma unroll
            for (int i = 0; i < WIN; ++i)
                v[j * WIN + i] = *reinterpret_cast<const T *>(img + (colOff[i] + rowOff[j]));
    }
    __device__ __forceinline__ void commit()
    {
        if (STAGED && arranged) {
            const int a = __builtin_amdgcn_readfirstlane(arranged) - 1;      // bit 1: lines are columns, bit 0: reversed
            if (a == 0) {
#pragma unroll
                for (int k = 0; k < WIN; ++k)
#pragma unroll
                    for (int e = 0; e < WIN; ++e) lds[k * WIN + e][tid] = (float)v[k * WIN + e];
            } else if (a == 1) {
#pragma unroll
                for (int k = 0; k < WIN; ++k)
#pragma unroll
                    for (int e = 0; e < WIN; ++e) lds[k * WIN + (WIN - 1 - e)][tid] = (float)v[k * WIN + e];
            } else if (a == 2) {
#pragma unroll
                for (int k = 0; k < WIN; ++k)
#pragma unroll
                    for (int e = 0; e < WIN; ++e) lds[e * WIN + k][tid] = (float)v[k * WIN + e];
            } else {
#pragma unroll
                for (int k = 0; k < WIN; ++k)
#pragma unroll
                    for (int e = 0; e < WIN; ++e) lds[(WIN - 1 - e) * WIN + k][tid] = (float)v[k * WIN + e];
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < WIN * WIN; ++k) lds[k][tid] = (float)v[k];
    }
    __device__ __forceinline__ void at(int slot, float (&vals)[1]) const { vals[0] = lds[slot][tid]; }
    __device__ __forceinline__ float reg(int slot) const { return (float)v[slot]; }
};

struct NoSrc {
    __device__ __forceinline__ float reg(int) const { return 1.f; }
    __device__ __forceinline__ void issue(int, int, unsigned long long, bool = false) {}
    __device__ __forceinline__ void commit() {}
    __device__ __forceinline__ void at(int, float (&vals)[1]) const { vals[0] = 1.f; }
};

}  // namespace aai
