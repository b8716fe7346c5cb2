// aai_quad_src.hpp -- the staged source window of the fp32 per-pixel kernels (device code only): shared by
// aai_rotated_quad.hip (one lane per dst pixel) and aai_rotated_cell.hip (one lane per cell of the dst grid).
#pragma once

#include "aai_kernels.hpp"
#include "aai_rot_quad.hpp"

namespace aai {

constexpr int kQuadBlock = 256;      // lanes per workgroup: the LDS window is [slot][kQuadBlock], one column per lane

// ---- XCD-aware workgroup order ---------------------------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (workgroup b of a launch runs on XCD b % 8: MI355X_MICROARCH.md), each with an L2 of
// its own.  In launch order horizontally adjacent tiles of a rotated request -- whose slanted footprints share 128-byte lines -- sit
// behind different L2s and every shared line crosses the fabric once per L2 that wants it: config 3's fast mode asked the fabric for
// 1.75 x its source and was bound by exactly that traffic (profiles/r04_fast_xcd.txt).  With `band` > 0 the linear workgroup index is
// re-read so that XCD x owns the tile rows [band (8 n + x), band (8 n + x + 1)) of every group of 8 band tile rows and walks them column
// by column: its neighbours in x (and, inside a band, in y) are its own predecessors, and the 8 XCDs advance in step over rows of
// (nearly) equal length, so none runs out of work early (the contiguous bands and cyclic super-tiles of rounds 2-3 lost to that).
// gridDim.y must be a multiple of 8 band (xcd_grid_rows); rows beyond the real ones are the caller's to skip.
__device__ __forceinline__ void xcd_tile(int band, int &tx, int &ty)
{
    if (band <= 0) return;
    const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, group = gridDim.x * 8u * (unsigned)band;
    const unsigned super = b / group, within = b - super * group, j = within >> 3;
    tx = (int)(j / (unsigned)band);
    ty = (int)((super * 8u + (within & 7u)) * (unsigned)band + j % (unsigned)band);
}
// the band height a launcher uses: its default, or AAI_XCD_ROWS in the experiments build
inline int xcd_band(int dflt)
{
    static const int env = [] { const char *e = experiment_env("AAI_XCD_ROWS"); return e ? atoi(e) : -1; }();
    return env >= 0 ? env : dflt;
}
// gridDim.y for `rows` tile rows under xcd_tile; 0: the order cannot be used (more than 65535 rows)
inline int xcd_grid_rows(int rows, int band)
{
    if (band <= 0) return rows;
    const int64_t padded = ((int64_t)rows + 8 * band - 1) / (8 * band) * (8 * band);
    return padded <= 65535 ? (int)padded : 0;
}

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
// ARR (kernels that read through reg(), plain fp32 images without replication, QuadMap::fastOk): 0 ... 3 = the arrangement of the
// window in memory -- bit 1: its lines are window COLUMNS (the contiguous axis is virtual Y), bit 0: the lines run against memory -- as a
// compile-time constant: v[] then stays in MEMORY order (line k, element e at v[k WIN + e]) and reg(slot) is a fixed register, where a
// run-time arrangement sent every element of the window through a chain of selects (~70 of the ~440 vector instructions of config
// 3's fast mode).  The kernel picks the instantiation with one wave-uniform switch (quad_arrangement).  -1: slot order, any map.
template <typename T, int WIN, bool SCALED, bool STAGED = false, int ARR = -1>
struct QuadSrc {
    const char *img;                 // first element of this image (band offset included)
    const QuadMap *m;
    int mW, mH;
    float (*lds)[kQuadBlock];        // [WIN * WIN][kQuadBlock]
    int tid;
    T v[WIN * WIN];
    int arranged;                    // STAGED: 0 = v[] is in slot order; else 1 + arrangement of the vector-loaded lines (commit)
    // Images of 4 GiB and more under the cell kernel: `img` points at a source row of the wave's own choosing and `rebase` is that row's
    // byte offset modulo 2^32 -- the lane offsets below are computed modulo 2^32 as for a small image and `rebase` taken off: what
    // remains is the true offset from `img` as long as the wave's rows span less than 4 GiB (cell_can_serve)
    uint32_t rebase = 0;

    // index of window position (i, j) in v[] under arrangement ARR
    static __device__ __forceinline__ constexpr int mem_index(int j, int i)
    {
        return ARR == 0 ? j * WIN + i : (ARR == 1 ? j * WIN + (WIN - 1 - i) : (ARR == 2 ? i * WIN + j : (ARR == 3 ? i * WIN + (WIN - 1 - j) : j * WIN + i)));
    }
    // allInside: the caller has established (by a wave vote) that every lane's window lies inside the lattice
    __device__ __forceinline__ void issue(int xg0, int yg0, unsigned long long, bool allInside = false)
    {
        unsigned colOff[WIN], rowOff[WIN];            // source indices along virtual X / Y first, byte offsets below
        const unsigned sxb = (unsigned)m->strideX * (unsigned)sizeof(T), syb = (unsigned)m->strideY * (unsigned)sizeof(T);
        arranged = 0;
        if (ARR >= 0) {
            // (the caller guarantees: no replication, fp32, QuadMap::fastOk, below 4 GiB)
            if (allInside) {
                // (two 24-bit multiplies, QuadMap::mul24Ok, and one three-operand add)
                uint32_t line = (uint32_t)__mul24(xg0, (int)m->fastSX) + (uint32_t)__mul24(yg0, (int)m->fastSY) + (m->fastC0 - (uint32_t)(WIN - 1) * m->fastRev4);
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
            // at the lattice's border: every position by itself, clamped onto the lattice, into its place in memory order
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int X = min(max(xg0 + i, 0), mW - 1), Y = min(max(yg0 + i, 0), mH - 1);
                colOff[i] = (unsigned)(m->flipX ? m->nX - 1 - X : X) * sxb;
                rowOff[i] = (unsigned)(m->flipY ? m->nY - 1 - Y : Y) * syb;
            }
#pragma unroll
            for (int k = 0; k < WIN; ++k)
#pragma unroll
                for (int e = 0; e < WIN; ++e) {
                    // (line k, element e in memory order) is window position (i, j):
                    const int i = ARR == 0 ? e : (ARR == 1 ? WIN - 1 - e : k), j = ARR == 0 || ARR == 1 ? k : (ARR == 2 ? e : WIN - 1 - e);
                    v[k * WIN + e] = *reinterpret_cast<const T *>(img + (colOff[i] + rowOff[j]));
                }
            return;
        }
        if (!SCALED && STAGED && allInside && sizeof(T) == 4 && m->anchorRows == 0 && m->fastOk) {
            // The common case of the staged kernels, spelt out: every window of the wave inside the lattice, so no clamps, and the
            // WIN lines (the window axis that is contiguous in memory: virtual X in quadrants 0 / 2, virtual Y in 1 / 3) start a
            // wave-uniform number of bytes apart -- ONE per-lane byte offset from the host-composed coefficients of the map
            // (QuadMap::fastC0 ...: a 24-bit multiply per axis and a three-operand add) and one add per further line, instead of 2 WIN
            // clamped indices, flips, multiplies and adds per lane
            uint32_t line = (uint32_t)__mul24(xg0, (int)m->fastSX) + (uint32_t)__mul24(yg0, (int)m->fastSY) + (m->fastC0 - (uint32_t)(WIN - 1) * m->fastRev4 - rebase);
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
        if (SCALED && allInside && WIN - 1 <= m->scale && m->anchorRows == 0 && m->mul24Ok) {
            // Replicated pixels, window no wider than a source pixel plus one: it spans at most two source columns and two
            // source rows -- four loads, and every position selects its value by which side of the split it lies on
            // (x >= 0 here, so (x + 0.5) / scale is at least 0.5 / scale away from an integer: the floor is exact -- in fp32 too,
            // because the planner sends replicated lattices of 2^22 pixels a side and more to the double-precision kernels
            // (make_rot_launch): the quotient's rounding error, 2^-23 x / scale, stays below 0.5 / scale)
            const int scale = m->scale;
            const int qx0 = (int)(((float)xg0 + 0.5f) * m->invScale), qy0 = (int)(((float)yg0 + 0.5f) * m->invScale);
            const int splitX = scale - (xg0 - __mul24(qx0, scale)), splitY = scale - (yg0 - __mul24(qy0, scale));      // in [1, scale]: first column / row of the second source pixel
            const int ux0 = m->flipX ? m->nX - 1 - qx0 : qx0, uy0 = m->flipY ? m->nY - 1 - qy0 : qy0;
            const unsigned c0 = (unsigned)__mul24(ux0, (int)sxb), r0 = (unsigned)__mul24(uy0, (int)syb) - rebase;
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
            // replicated pixels, any window (also one that misses the lattice altogether): replicated_indices, aai_rot_quad.hpp
            int qx[WIN], qy[WIN];
            replicated_indices<WIN>(xg0, mW, m->scale, m->invScaleD, m->invScale, qx);
            replicated_indices<WIN>(yg0, mH, m->scale, m->invScaleD, m->invScale, qy);
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                colOff[i] = (unsigned)(m->flipX ? m->nX - 1 - qx[i] : qx[i]);
                rowOff[i] = (unsigned)(m->flipY ? m->nY - 1 - qy[i] : qy[i]);
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
        for (int i = 0; i < WIN; ++i) { colOff[i] *= sxb; rowOff[i] = rowOff[i] * syb - rebase; }
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
    __device__ __forceinline__ float reg(int slot) const { return (float)v[ARR >= 0 ? mem_index(slot / WIN, slot % WIN) : slot]; }
};

// the arrangement of a plain fp32 window in memory (QuadSrc's ARR), or -1 where the register-window kernels keep slot order
template <typename T, bool SCALED>
__device__ __forceinline__ int quad_arrangement(const QuadMap &m)
{
    if (SCALED || sizeof(T) != 4 || !m.fastOk || m.anchorRows != 0) return -1;
    return (m.fastAlongX ? 0 : 2) + (m.fastRev4 ? 1 : 0);
}

// One dst pixel (or one part of a wide window) of fast mode through the register window: the window's arrangement in memory picks
// the instantiation (one wave-uniform switch), so that every window value is read from a fixed register (QuadSrc's ARR).
template <typename T, int WIN, bool SCALED, int ARR>
__device__ __forceinline__ void fast_window_sum_arr(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const char *img, int Xc, int Yc,
                                                    double dfx, double dfy, int tid, int partI, int partJ, float &sum, int &count)
{
    QuadSrc<T, WIN, SCALED, false, ARR> s;
    s.img = img; s.m = &m; s.mW = r.mW; s.mH = r.mH; s.lds = nullptr; s.tid = tid;
    quad_fast_pixel<float, WIN, false>(q, Xc, Yc, dfx, dfy, r.mW, r.mH, s, sum, count, partI, partJ);
}
template <typename T, int WIN, bool SCALED>
__device__ __forceinline__ void fast_window_sum(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const char *img, int Xc, int Yc,
                                                double dfx, double dfy, int tid, int partI, int partJ, float &sum, int &count)
{
    if constexpr (!SCALED && sizeof(T) == 4) {
        switch (quad_arrangement<T, SCALED>(m)) {                     // (wave-uniform: a scalar branch)
        case 0: return fast_window_sum_arr<T, WIN, SCALED, 0>(r, q, m, img, Xc, Yc, dfx, dfy, tid, partI, partJ, sum, count);
        case 1: return fast_window_sum_arr<T, WIN, SCALED, 1>(r, q, m, img, Xc, Yc, dfx, dfy, tid, partI, partJ, sum, count);
        case 2: return fast_window_sum_arr<T, WIN, SCALED, 2>(r, q, m, img, Xc, Yc, dfx, dfy, tid, partI, partJ, sum, count);
        case 3: return fast_window_sum_arr<T, WIN, SCALED, 3>(r, q, m, img, Xc, Yc, dfx, dfy, tid, partI, partJ, sum, count);
        default: break;
        }
    }
    fast_window_sum_arr<T, WIN, SCALED, -1>(r, q, m, img, Xc, Yc, dfx, dfy, tid, partI, partJ, sum, count);
}
template <typename T, int WIN, bool SCALED>
__device__ __forceinline__ float fast_window_value(const RotLaunch &r, const QuadConsts<float> &q, const QuadMap &m, const char *img, int Xc, int Yc,
                                                   double dfx, double dfy, int tid)
{
    float sum;
    int count;
    fast_window_sum<T, WIN, SCALED>(r, q, m, img, Xc, Yc, dfx, dfy, tid, 0, 0, sum, count);
    return count > 0 ? sum / (float)count : 0.f;                      // Source.cpp:905
}

// (kQuadBlock = 256 lanes: 16 x 16 dst pixels, the tiling of the scans -- aai_quad_src.hpp)
constexpr int kQuadMaxChan = 4;

// the `chan` (2..4) interleaved fp32 channels of one pixel in ONE load instruction (element-aligned)
__device__ __forceinline__ void load_channels(const float *p, int chan, float (&v)[kQuadMaxChan])
{
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    v[2] = 0.f; v[3] = 0.f;
    if (chan == 3) { const f3u q = *reinterpret_cast<const f3u *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; }
    else if (chan == 4) { const f4u q = *reinterpret_cast<const f4u *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else { const f2u q = *reinterpret_cast<const f2u *>(p); v[0] = q.x; v[1] = q.y; }
}

// Interleaved channels (2..4 per pixel): the same window, every slot holding all channels of its pixel -- as WORDS raw
// words, so that 8-bit RGB(A) costs one LDS word per slot like a plain image (16-bit: one or two, fp32: one per
// channel).  Like the plain window it is fetched up front into registers and parked in LDS after the classification.
template <typename T, int WIN, bool SCALED, int WORDS>
struct QuadSrcMulti {
    const char *img;
    const QuadMap *m;
    int mW, mH, chan;
    unsigned *lds;                   // [WIN * WIN * WORDS][kQuadBlock]
    int tid;
    unsigned v[WIN * WIN * WORDS];

    __device__ __forceinline__ void issue(int xg0, int yg0, unsigned long long, bool = false)
    {
        unsigned colOff[WIN], rowOff[WIN];
        const unsigned sxb = (unsigned)m->strideX * (unsigned)sizeof(T), syb = (unsigned)m->strideY * (unsigned)sizeof(T);
        // every position clamped onto the lattice (its value is fetched but never read when its valid bit is clear); under replication
        // divided by the scale in a form that is exact for any window origin (replicated_indices)
        int qx[WIN], qy[WIN];
        if (SCALED) {
            replicated_indices<WIN>(xg0, mW, m->scale, m->invScaleD, m->invScale, qx);
            replicated_indices<WIN>(yg0, mH, m->scale, m->invScaleD, m->invScale, qy);
        } else {
#pragma unroll
            for (int i = 0; i < WIN; ++i) { qx[i] = min(max(xg0 + i, 0), mW - 1); qy[i] = min(max(yg0 + i, 0), mH - 1); }
        }
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            colOff[i] = (unsigned)(m->flipX ? m->nX - 1 - qx[i] : qx[i]) * sxb;
            rowOff[i] = (unsigned)(m->flipY ? m->nY - 1 - qy[i] : qy[i]) * syb;
        }
#pragma unroll
        for (int j = 0; j < WIN; ++j)
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const T *p = reinterpret_cast<const T *>(img + (colOff[i] + rowOff[j]));
                unsigned *w = v + (j * WIN + i) * WORDS;
                if (sizeof(T) == 4) {
                    float f[kQuadMaxChan];
                    load_channels(reinterpret_cast<const float *>(p), chan, f);
#pragma unroll
                    for (int c = 0; c < WORDS; ++c) w[c] = __float_as_uint(f[c]);
                } else if (sizeof(T) == 2) {
                    // 2..4 sixteen-bit channels = 4..8 bytes: ONE load (the number of load instructions, each touching ~64
                    // cache lines, is what bounds these kernels), started early enough to stay inside the image
                    typedef unsigned u1w __attribute__((aligned(2)));
                    typedef unsigned u2w __attribute__((ext_vector_type(2), aligned(2)));
                    const unsigned want = colOff[i] + rowOff[j];
                    if (WORDS == 1) w[0] = *reinterpret_cast<const u1w *>(img + want);
                    else {
                        const unsigned from = min(want, m->lastLoad8);
                        const u2w q2 = *reinterpret_cast<const u2w *>(img + from);
                        const unsigned long long both = (((unsigned long long)q2.y << 32) | q2.x) >> (8u * (want - from));
                        w[0] = (unsigned)both;
                        w[WORDS - 1] = chan > 3 ? (unsigned)(both >> 32) : (unsigned)(both >> 32) & 65535u;
                    }
                } else {
                    typedef unsigned u1b __attribute__((aligned(1)));
                    typedef unsigned short u1s __attribute__((aligned(1)));
                    const unsigned want = colOff[i] + rowOff[j];
                    if (chan == 2) w[0] = *reinterpret_cast<const u1s *>(img + want);
                    else {
                        const unsigned from = min(want, m->lastLoad4);
                        const unsigned q1 = *reinterpret_cast<const u1b *>(img + from) >> (8u * (want - from));
                        w[0] = chan > 3 ? q1 : q1 & 0xffffffu;
                    }
                }
            }
    }
    __device__ __forceinline__ void commit()
    {
#pragma unroll
        for (int k = 0; k < WIN * WIN * WORDS; ++k) lds[(size_t)k * kQuadBlock + tid] = v[k];
    }
    __device__ __forceinline__ void at(int slot, float (&vals)[kQuadMaxChan]) const
    {
        const unsigned *p = lds + (size_t)(slot * WORDS) * kQuadBlock + tid;
        if (sizeof(T) == 4) {
#pragma unroll
            for (int c = 0; c < kQuadMaxChan; ++c) vals[c] = c < WORDS ? __uint_as_float(p[(size_t)c * kQuadBlock]) : 0.f;
        } else if (sizeof(T) == 2) {
            const unsigned w0 = p[0], w1 = WORDS > 1 ? p[kQuadBlock] : 0u;
            vals[0] = (float)(w0 & 65535u); vals[1] = (float)(w0 >> 16); vals[2] = (float)(w1 & 65535u); vals[3] = (float)(w1 >> 16);
        } else {
            const unsigned w0 = p[0];
            vals[0] = (float)(w0 & 255u); vals[1] = (float)((w0 >> 8) & 255u); vals[2] = (float)((w0 >> 16) & 255u); vals[3] = (float)(w0 >> 24);
        }
    }
};

struct NoSrc {
    __device__ __forceinline__ float reg(int) const { return 1.f; }
    __device__ __forceinline__ void issue(int, int, unsigned long long, bool = false) {}
    __device__ __forceinline__ void commit() {}
    __device__ __forceinline__ void at(int, float (&vals)[1]) const { vals[0] = 1.f; }
};

}  // namespace aai
