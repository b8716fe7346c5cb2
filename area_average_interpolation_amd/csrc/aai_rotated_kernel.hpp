// aai_rotated_kernel.hpp -- the rotated-lattice area / fast kernel template (K2/K3), shared by two
// translation units: aai_rotated.hip instantiates the production pass (STRICT = false, fused multiply-adds
// allowed) and aai_rotated_strict.hip the knife-edge fix-up pass (STRICT = true, compiled with
// -ffp-contract=off because it replays the reference's arithmetic operation by operation).
#pragma once

#include "aai_kernels.hpp"
#include "aai_rot_math.hpp"
#include "aai_strict.hpp"

namespace aai {

constexpr int kRotListCap = 24;    // queued vertex-region pairs per lane (overflow is processed in line)
constexpr int kRotBlock = 256;     // 16 x 16 dst pixels; a wave covers 16 x 4

// four consecutive source elements as fp32; p only needs element alignment
__device__ __forceinline__ void load4(const float *p, float v[4])
{
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const f4u q = *reinterpret_cast<const f4u *>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
template <typename T>
__device__ __forceinline__ void load4(const T *p, float v[4])
{
    v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2]; v[3] = (float)p[3];
}

// the `chan` (2..4) interleaved channels of one pixel as fp32: ONE load instruction for fp32 images (neighbouring lanes
// are L source pixels apart, so the number of load instructions, each touching ~64 cache lines, is what the
// multi-channel kernels are bound by: three 4-byte loads per pixel made RGB 2.4x slower than one channel)
__device__ __forceinline__ void load_pixel(const float *p, int chan, float v[4])
{
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    v[2] = 0.f; v[3] = 0.f;
    if (chan == 3) { const f3u q = *reinterpret_cast<const f3u *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; }
    else if (chan == 4) { const f4u q = *reinterpret_cast<const f4u *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else { const f2u q = *reinterpret_cast<const f2u *>(p); v[0] = q.x; v[1] = q.y; }
}
template <typename T>
__device__ __forceinline__ void load_pixel(const T *p, int chan, float v[4])
{
    v[0] = (float)p[0]; v[1] = (float)p[1];
    v[2] = chan > 2 ? (float)p[2] : 0.f;
    v[3] = chan > 3 ? (float)p[3] : 0.f;
}

// STRICT = false: the production pass.  Every pair is answered by the fast path.
// STRICT = true: the double-precision fix-up pass over the dst pixels the plan's scans flagged (once per geometry:
//   aai_knife_scan_kernel for the reference's knife edges, aai_quad_scan_kernel for decisions the fp32
//   production pass must not take).  Each listed pixel is recomputed in double precision, replaying
//   the reference's own arithmetic (aai_strict.hpp) for the knife-edge pairs.
// MULTI: interleaved channels (RotLaunch::chan = 2..4).  The geometry -- classification and overlap areas -- is computed
// once per (dst, src) pair and applied to every channel; false compiles the single-channel kernel unchanged.
constexpr int kMaxChan = 4;

template <int MODE, bool STRICT, typename T, bool MULTI = false>
__global__ __launch_bounds__(kRotBlock) void aai_rotated_kernel(RotLaunch r, const T *__restrict__ src, ImageView sv,
                                                              float *__restrict__ dst, ImageView dv,
                                                              const uint2 *__restrict__ pixelList, unsigned nList)
{
    __shared__ unsigned short pending[kRotListCap][kRotBlock];

    const int tid = threadIdx.x;
    int dx, dy;
    bool valid;
    if (STRICT && pixelList) {
        // The fix-up pass runs over the plan's list of flagged dst pixels (one lane each).  The list depends on the
        // geometry alone, so one list serves every image of a batch and every row band.
        const unsigned e = blockIdx.x * kRotBlock + (unsigned)tid;
        valid = e < nList;
        const uint2 p = valid ? pixelList[e] : make_uint2(0u, 0u);
        dx = (int)p.x; dy = (int)p.y;
        valid = valid && dy >= r.dyBase && dy < r.dyEnd;
    } else {
        // (STRICT without a list: so many pixels are flagged that the whole image takes this pass)
        dx = blockIdx.x * 16 + (tid & 15);
        dy = r.dyBase + blockIdx.y * 16 + (tid >> 4);
        valid = dx < r.dW && dy < r.dyEnd;
    }
    const T *img = src + (int64_t)blockIdx.z * sv.imageStride;

    // (Staging the tile's source footprint in LDS with coalesced loads was measured and rejected: cfg3 1452 ->
    // 1744 us, cfg3 fast 435 -> 560 us, cfg5 17.3 -> 21.4 ms.  The per-lane 4-byte loads hit L1/L2 and are not the
    // limiter; the extra pass, the barrier and the lost occupancy cost more.)
    const int chan = MULTI ? r.chan : 1;
    // acc[c] += w * value of channel c at virtual pixel (X, Y)
    auto add_pixel = [&](double (&acc)[MULTI ? kMaxChan : 1], double w, int X, int Y) {
        const T *p = img + virt_offset(r, X, Y, sv.rowStride, chan);
        if (!MULTI) acc[0] += w * (double)p[0];
        else {
            float v[kMaxChan];
            load_pixel(p, chan, v);
#pragma unroll
            for (int c = 0; c < kMaxChan; ++c)
                if (c < chan) acc[c] += w * (double)v[c];
        }
    };
    // write acc[c] * scale (or 0 when the pixel got no weight) to the dst pixel's channels
    auto store_pixel = [&](float *out, const double (&acc)[MULTI ? kMaxChan : 1], bool any, double denom) {
        out[0] = any ? (float)(acc[0] / denom) : 0.f;
        if (MULTI) {
#pragma unroll
            for (int c = 1; c < kMaxChan; ++c)
                if (c < chan) out[c] = any ? (float)(acc[c] / denom) : 0.f;
        }
    };

    if (valid) {
        float *out = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride + (int64_t)dx * chan;

        double px, py;
        pixel_centre(r, dx, dy, px, py);

        // tight bounding box of the square; nothing outside it can overlap (the reference searches a wider
        // window, Source.cpp:426-429, whose extra pixels all classify as "not included")
        int x0, x1, y0, y1;
        rot_window(r, px, py, x0, x1, y0, y1);

        SVec sv4[4];                                 // the reference's vertices, fetched lazily (STRICT only)
        bool haveVertices = false;

        if (MODE == AAI_MODE_FAST && !STRICT) {
            // Production pass of the fast mode.  The centres inside the closed square form one interval per
            // source row (two pairs of parallel edges = two interval constraints on X), so there is no
            // per-pixel test: bounds from four multiplications per row, then loads and adds.  Centres within
            // 2e-9 of an edge would make the integer bounds ambiguous -- exactly the pixels the knife-edge scan
            // flags and the fix-up pass (the per-pixel loop below) redoes.
            int count = 0;
            double acc[MULTI ? kMaxChan : 1] = {};
            if (MULTI && r.scale == 1) {
                // lines = source rows (see below); one vector load per pixel brings all its channels
                const bool cols = virt_lines_are_columns(r);
                const int u0 = cols ? x0 : y0, u1 = cols ? x1 : y1;
                const int nIn = cols ? r.mH : r.mW;
                const double pIn = cols ? py : px, pOut = cols ? px : py;
                for (int u = u0; u <= u1; ++u) {
                    double lo, hi;
                    centre_interval(r, cols, u - pOut, lo, hi);
                    const double da = fmax(ceil(pIn + lo), 0.0), db = fmin(floor(pIn + hi), (double)(nIn - 1));
                    if (!(da <= db)) continue;
                    const int wa = (int)da, wb = (int)db;
                    for (int w = wa; w <= wb; ++w) add_pixel(acc, 1.0, cols ? u : w, cols ? w : u);
                    count += wb - wa + 1;
                }
            } else if (!MULTI && r.scale == 1 && min(r.mW, r.mH) >= 4) {
                // Without replication a "line" of virtual pixels is a source row (virt_line): walk the lines that way
                // round -- rows in quadrants 0/2, columns in 1/3 -- and fetch each interval four source columns at a
                // time (one dword-aligned 16-byte load per lane): neighbouring lanes are L source pixels apart, so
                // it is the number of load instructions, each touching ~64 cache lines, that bounds large footprints.
                const bool cols = virt_lines_are_columns(r);
                const int u0 = cols ? x0 : y0, u1 = cols ? x1 : y1;
                const int nIn = cols ? r.mH : r.mW;
                const double pIn = cols ? py : px, pOut = cols ? px : py;
                double acc1 = 0.0;
                for (int u = u0; u <= u1; ++u) {
                    double lo, hi;
                    centre_interval(r, cols, u - pOut, lo, hi);
                    const double da = fmax(ceil(pIn + lo), 0.0), db = fmin(floor(pIn + hi), (double)(nIn - 1));
                    if (!(da <= db)) continue;
                    const int wa = (int)da, wb = (int)db;
                    bool rev;
                    const T *srow = img + virt_line(r, u, sv.rowStride, rev);
                    const int sa = rev ? nIn - 1 - wb : wa, sb = rev ? nIn - 1 - wa : wb;      // source columns, ascending
                    for (int c0 = sa; c0 <= sb; c0 += 4) {
                        const int cc = min(c0, nIn - 4);       // keep the vector inside the row; elements left of c0 were done
                        float v[4];
                        load4(srow + cc, v);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (cc + j >= c0 && cc + j <= sb) { if (j & 1) acc1 += (double)v[j]; else acc[0] += (double)v[j]; }
                    }
                    count += wb - wa + 1;
                }
                acc[0] += acc1;
            } else
            for (int Y = y0; Y <= y1; ++Y) {
                const double ey = Y - py;
                const double es = ey * r.s, ec = ey * r.c;
                const double lo = fmax((es - r.h) * r.rc, (-r.h - ec) * r.rs);
                const double hi = fmin((es + r.h) * r.rc, (r.h - ec) * r.rs);
                // clamp in double before converting: near-axis rotations make the unconstrained bounds astronomically large
                const double da = fmax(ceil(px + lo), 0.0), db = fmin(floor(px + hi), (double)(r.mW - 1));
                if (!(da <= db)) continue;
                const int xa = (int)da, xb = (int)db;
                for (int X = xa; X <= xb; ++X) add_pixel(acc, 1.0, X, Y);
                count += xb - xa + 1;
            }
            store_pixel(out, acc, count > 0, (double)count);      // Source.cpp:905
        } else if (MODE == AAI_MODE_FAST) {
            // closed-square membership of the pixel centre with the reference's parameter slack
            const double lim = r.h + DBL_EPSILON * r.side;
            int count = 0;
            double acc[MULTI ? kMaxChan : 1] = {};
            for (int Y = y0; Y <= y1; ++Y)
                for (int X = x0; X <= x1; ++X) {
                    const double ex = X - px, ey = Y - py;
                    const double a = fabs(ex * r.c - ey * r.s), b = fabs(ex * r.s + ey * r.c);
                    bool in = a <= lim && b <= lim;
                    // a centre on (or within the guard of) an edge: the reference's ray cast decides
                    const bool edgy = STRICT && ((fabs(a - r.h) < AAI_KNIFE_GUARD && b <= r.h + AAI_KNIFE_GUARD) ||
                                                 (fabs(b - r.h) < AAI_KNIFE_GUARD && a <= r.h + AAI_KNIFE_GUARD));
                    if (edgy) {
                        if (STRICT) {
                            if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                            SVec pc; pc.x = X; pc.y = Y;
                            in = strict_centre_inside(pc, sv4);
                        }
                    }
                    if (in) { ++count; add_pixel(acc, 1.0, X, Y); }
                }
            store_pixel(out, acc, count > 0, (double)count);      // Source.cpp:905
        } else {
            // Pass 1 over the window: pairs that are outside, inside, or cut by a single edge line are settled
            // on the spot (two dot products + a closed form); the rest -- pixels near a dst vertex -- are queued.
            double sumA = 0.0;
            double sumVA[MULTI ? kMaxChan : 1] = {};
            int nPend = 0;
            const bool packable = (x1 - x0) < 256 && (y1 - y0) < 128;
            for (int Y = y0; Y <= y1; ++Y) {
                for (int X = x0; X <= x1; ++X) {
                    const double ex = X - px, ey = Y - py;
                    const double a = ex * r.c - ey * r.s, b = ex * r.s + ey * r.c;
                    double d = 0.0;
                    bool edgy = false, edgy2 = false;
                    const int cls = classify_pair<STRICT>(r, a, b, d, edgy);
                    if (cls == PAIR_OUTSIDE) continue;                              // type 0
                    double area;
                    if (cls == PAIR_INSIDE) area = 1.0;                             // type 1
                    else if (cls == PAIR_GENERAL) {
                        if (packable && nPend < kRotListCap) {
                            pending[nPend++][tid] = (unsigned short)((edgy ? 0x8000 : 0) | ((Y - y0) << 8) | (X - x0));
                            continue;
                        }
                        area = wedge_pair_area<STRICT>(r, px - (X - 0.5), py - (Y - 0.5), a < 0.0, b < 0.0, r.policy, edgy2);
                    } else area = single_cut_area<STRICT>(r, d, cls == PAIR_CUT_LR, r.policy, edgy2);   // types 2, 3, 4
                    if (STRICT && (edgy || edgy2)) {
                        {
                            if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                            area = strict_pair_area(sv4, X, Y, r.policy);
                        }
                    }
                    if (area != 0.0) {
                        sumA += area;
                        add_pixel(sumVA, area, X, Y);
                    }
                }
            }
            // Pass 2: the queued pairs, dense across the lanes of a wave.
            for (int i = 0; i < nPend; ++i) {
                const unsigned short code = pending[i][tid];
                const int X = x0 + (code & 255), Y = y0 + ((code >> 8) & 127);
                bool edgy = false;
                const double ex = X - px, ey = Y - py;
                const bool nearLeft = ex * r.c - ey * r.s < 0.0, nearTop = ex * r.s + ey * r.c < 0.0;
                double area = wedge_pair_area<STRICT>(r, px - (X - 0.5), py - (Y - 0.5), nearLeft, nearTop, r.policy, edgy);
                if (STRICT && (edgy || (code & 0x8000))) {
                    {
                        if (!haveVertices) { strict_vertices(r, dx, dy, sv4); haveVertices = true; }
                        area = strict_pair_area(sv4, X, Y, r.policy);
                    }
                }
                if (area != 0.0) {
                    sumA += area;
                    add_pixel(sumVA, area, X, Y);
                }
            }
            store_pixel(out, sumVA, DBL_EPSILON < fabs(sumA), sumA);   // Source.cpp:577
        }
    }
}


}  // namespace aai
