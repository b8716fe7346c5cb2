// aai_rotated.hip -- K2 (area average at a general rotation), K3 (fast / centre-inclusion mode) and the
// K4/K5 bilinear / bicubic comparison samplers for gfx950.
//
// K2 replaces Source.cpp:413-579 + 962-1431 of the reference: for every dst pixel -- a square of side L
// (> sqrt 2) virtual-source pixels, rotated by the reduced angle about its centre -- accumulate
// sum(area*value)/sum(area) over the virtual source pixels it overlaps.  The reference classifies each
// (dst,src) pair with 32 segment tests and a 19-pattern area table; here
//   * pairs that are wholly inside / outside the square are settled by two dot products;
//   * the remaining "boundary" pairs get their EXACT overlap area from a closed-form scan-line integral of
//     the square's left/right boundary clamped to the pixel (no polygon lists, no divisions);
//   * AAI_POLICY_REFERENCE then substitutes the reference's value for the one family of pairs where its
//     table departs from the exact area: a lone left/right dst edge cutting exactly one corner of the
//     source pixel (top-right or bottom-left), where Source.cpp:1055-1062 multiplies the complementary
//     legs.
// All geometry is fp64 (the substituted area is discontinuous, so classification must agree with the
// reference's double-precision decisions); pixel values are fp32, accumulated in fp64.
//
// K3 replaces Source.cpp:868-907 + 837-864: mean of the virtual pixels whose centres lie in the closed
// dst square.  K4/K5 are build-defined (the reference only names them in README.md:8): point samples of
// the original image at the dst pixel centre mapped through the same affine map, clamp-to-edge taps,
// 0 outside the image, Keys a = -0.5 for the cubic.
//
// One lane per dst pixel, a wave covers a 16x4 dst tile; pairs cut by a single edge line get a
// closed form on the spot; pairs near a dst vertex are compacted into a per-lane LDS list so that the
// general clip runs ~6 times per pixel, dense across the wave, instead of once per window position.
#include "aai_rotated_kernel.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace aai {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGridY = 65535;

// ---- knife-edge scan (once per geometry) ------------------------------------------------------------------
// Same tiling as aai_rotated_kernel: one 64-bit word per wave has the bits of its dst pixels with a vertex on
// a pixel-boundary line or an edge through a lattice point (aai_rot_math.hpp: pixel_on_knife_edge);
// counter[0] counts the flagged pixels.  Depends on the geometry only, so the plan runs it once and keeps
// the list of flagged pixels; generic geometries (every BASELINE configuration) have no knife edge.
__global__ __launch_bounds__(kRotBlock) void aai_knife_scan_kernel(RotLaunch r, unsigned long long *__restrict__ laneMasks, unsigned *__restrict__ counter, int tileRow0)
{
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dx = blockIdx.x * 16 + (tid & 15);
    const int dy = (tileRow0 + blockIdx.y) * 16 + (tid >> 4);
    bool knife = false;
    if (dx < r.dW && dy < r.dH) {
        double px, py;
        pixel_centre(r, dx, dy, px, py);
        knife = pixel_on_knife_edge(r, px, py, r.mode != AAI_MODE_FAST);
    }
    const unsigned long long any = __ballot(knife);
    if ((tid & 63) == 0) {
        laneMasks[((size_t)(tileRow0 + blockIdx.y) * gridDim.x + blockIdx.x) * (kRotBlock / 64) + wave] = any;
        if (any != 0ull) atomicAdd(counter, (unsigned)__popcll(any));
    }
}

// ---- K2 for large footprints: rows as runs ------------------------------------------------------------------
// Production pass of the area mode when the dst square has a wide interior (RotLaunch::runs, heavy down-sampling at
// an angle).  Same lane-per-dst-pixel tiling and the same per-pair arithmetic as aai_rotated_kernel, but each source
// row of the window is split by line_runs() into [boundary | interior | boundary]: interior pixels have area exactly
// 1 and are just summed (independent loads, nothing to classify), pixels outside the touched interval are never
// visited, and only the ~6 L boundary pixels go through classify_pair.  aai_rotated_kernel visits all
// (1.26 L + 1)^2 window positions and waits for one dependent load per overlapping position, which at L >= 6 leaves
// it latency-bound (profiles/r01_rotated_envelope.txt).  Sums are reassociated (interior first), a difference of
// ~1e-16 relative.  The knife-edge fix-up pass is unchanged: flagged waves are redone by the strict kernel.
// MULTI: interleaved channels -- the same runs, every pixel's channels fetched by one vector load (load_pixel).
template <typename T, bool MULTI>
__global__ __launch_bounds__(kRotBlock) void aai_rotated_runs_kernel(RotLaunch r, const T *__restrict__ src, ImageView sv,
                                                                   float *__restrict__ dst, ImageView dv)
{
    constexpr int NC = MULTI ? kMaxChan : 1;
    __shared__ unsigned short pending[kRotListCap][kRotBlock];

    const int tid = threadIdx.x;
    const int dx = blockIdx.x * 16 + (tid & 15);
    const int dy = r.dyBase + blockIdx.y * 16 + (tid >> 4);
    if (!(dx < r.dW && dy < r.dyEnd)) return;          // no barrier below
    const T *img = src + (int64_t)blockIdx.z * sv.imageStride;
    const int chan = MULTI ? r.chan : 1;
    float *out = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy - r.dyBase) * dv.rowStride + (int64_t)dx * chan;

    double px, py;
    pixel_centre(r, dx, dy, px, py);
    int x0, x1, y0, y1;
    rot_window(r, px, py, x0, x1, y0, y1);

    double sumA = 0.0;
    double sumVA[NC] = {};
    int nPend = 0;
    const bool packable = (x1 - x0) < 256 && (y1 - y0) < 128;
    // lines = virtual rows (quadrants 0, 2) or virtual columns (quadrants 1, 3): whichever is a SOURCE row, so that
    // a lane walks contiguous memory
    const bool cols = virt_lines_are_columns(r);
    const int u0 = cols ? x0 : y0, u1 = cols ? x1 : y1, w0 = cols ? y0 : x0, w1 = cols ? y1 : x1;
    const int nIn = cols ? r.mH : r.mW;
    const double pIn = cols ? py : px, pOut = cols ? px : py;
    for (int u = u0; u <= u1; ++u) {
        int t0, t1, i0, i1;
        line_runs(r, cols, pIn, u - pOut, w0, w1, t0, t1, i0, i1);
        if (t0 > t1) continue;
        bool rev;
        const T *srow = img + virt_line(r, u, sv.rowStride, rev);
        // one boundary pixel: the body of aai_rotated_kernel's first pass
        auto boundary = [&](int w, const float (&v)[NC]) {
            const int X = cols ? u : w, Y = cols ? w : u;
            const double ex = X - px, ey = Y - py;
            const double a = ex * r.c - ey * r.s, b = ex * r.s + ey * r.c;
            double d = 0.0;
            bool edgy = false, edgy2 = false;
            const int cls = classify_pair<false>(r, a, b, d, edgy);
            if (cls == PAIR_OUTSIDE) return;
            double area;
            if (cls == PAIR_INSIDE) area = 1.0;
            else if (cls == PAIR_GENERAL) {
                if (packable && nPend < kRotListCap) {
                    pending[nPend++][tid] = (unsigned short)(((Y - y0) << 8) | (X - x0));
                    return;
                }
                area = wedge_pair_area<false>(r, px - (X - 0.5), py - (Y - 0.5), a < 0.0, b < 0.0, r.policy, edgy2);
            } else area = single_cut_area<false>(r, d, cls == PAIR_CUT_LR, r.policy, edgy2);
            if (area != 0.0) {
                sumA += area;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (c < chan) sumVA[c] += area * (double)v[c];
            }
        };
        double s0 = 0.0, s1 = 0.0;
        if (MULTI) {
            double sc[NC] = {};
            for (int w = t0; w <= t1; ++w) {
                float v[kMaxChan];
                load_pixel(srow + (int64_t)(rev ? nIn - 1 - w : w) * chan, chan, v);
                float vv[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) vv[c] = v[c];
                if (w >= i0 && w <= i1) {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (c < chan) sc[c] += (double)vv[c];
                } else boundary(w, vv);
            }
            if (i0 <= i1) {
#pragma unroll
                for (int c = 0; c < NC; ++c) sumVA[c] += sc[c];
                sumA += (double)(i1 - i0 + 1);
            }
            continue;
        }
        if (nIn >= 4) {
            // Fetch the touched segment four source columns at a time (one dword-aligned 16-byte load per lane instead
            // of four 4-byte loads: neighbouring lanes are L source pixels apart, so every load instruction touches
            // ~64 cache lines and their number is what bounds this kernel), then deal with the four positions from
            // registers.
            const int sa = rev ? nIn - 1 - t1 : t0, sb = rev ? nIn - 1 - t0 : t1;        // source columns, ascending
            for (int c0 = sa; c0 <= sb; c0 += 4) {
                const int cc = min(c0, nIn - 4);               // keep the vector inside the row; elements left of c0 were done
                float v[4];
                load4(srow + cc, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sx = cc + j;
                    if (sx < c0 || sx > sb) continue;
                    const int w = rev ? nIn - 1 - sx : sx;
                    if (w >= i0 && w <= i1) { if (j & 1) s1 += (double)v[j]; else s0 += (double)v[j]; }
                    else { const float one[NC] = {v[j]}; boundary(w, one); }
                }
            }
        } else {
            for (int w = t0; w <= t1; ++w) {
                const float one[NC] = {(float)srow[rev ? nIn - 1 - w : w]};
                if (w >= i0 && w <= i1) s0 += (double)one[0];
                else boundary(w, one);
            }
        }
        if (i0 <= i1) { sumVA[0] += s0 + s1; sumA += (double)(i1 - i0 + 1); }
    }
    for (int i = 0; i < nPend; ++i) {
        const unsigned short code = pending[i][tid];
        const int X = x0 + (code & 255), Y = y0 + (code >> 8);
        bool edgy = false;
        const double ex = X - px, ey = Y - py;
        const bool nearLeft = ex * r.c - ey * r.s < 0.0, nearTop = ex * r.s + ey * r.c < 0.0;
        const double area = wedge_pair_area<false>(r, px - (X - 0.5), py - (Y - 0.5), nearLeft, nearTop, r.policy, edgy);
        if (area != 0.0) {
            sumA += area;
            const T *p = img + virt_offset(r, X, Y, sv.rowStride, chan);
            if (!MULTI) sumVA[0] += area * (double)p[0];
            else {
                float v[kMaxChan];
                load_pixel(p, chan, v);
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (c < chan) sumVA[c] += area * (double)v[c];
            }
        }
    }
    const bool any = DBL_EPSILON < fabs(sumA);                        // Source.cpp:577
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (c < chan) out[c] = any ? (float)(sumVA[c] / sumA) : 0.f;
}

// ---- K4/K5 -------------------------------------------------------------------------------------------
// N (2 or 4) consecutive fp32 taps from an element-aligned address in one load
template <int N>
__device__ __forceinline__ void load_taps(const float *p, float (&t)[N])
{
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    if (N == 2) { const f2u q = *reinterpret_cast<const f2u *>(p); t[0] = q.x; t[1] = q.y; }
    else { const f4u q = *reinterpret_cast<const f4u *>(p); t[0] = q.x; t[1] = q.y; t[N > 2 ? 2 : 0] = q.z; t[N > 3 ? 3 : 0] = q.w; }
}

__device__ __forceinline__ void keys(float t, float w[4])
{
    // Every multiply-add spelled out: the planar fast path and the general path (typed sources, interleaved channels) must
    // round these weights alike, whatever the compiler would fuse in either place.
    const float a = -0.5f, t2 = t * t, t3 = t2 * t;
    w[0] = a * (fmaf(-2.f, t2, t3) + t);                                  // a (t^3 - 2 t^2 + t)
    w[1] = fmaf(a + 2.f, t3, fmaf(-(a + 3.f), t2, 1.f));                  // (a + 2) t^3 - (a + 3) t^2 + 1
    w[2] = fmaf(-(a + 2.f), t3, fmaf(2.f * a + 3.f, t2, -(a * t)));       // -(a + 2) t^3 + (2 a + 3) t^2 - a t
    w[3] = a * (t2 - t3);                                                 // a (t^2 - t^3)
}

// one sample point: integer tap origin, fractions, and whether it lies outside the image extent
struct SamplePoint { int ix, iy; float tx, ty; bool outside; };

// the sample point of dst pixel (dx, dy) in continuous original-image coordinates: affine in (dx, dy), coefficients from
// the host (the same expression for every caller, so that row bands reproduce the full image bit for bit)
__device__ __forceinline__ SamplePoint sample_point(const RotLaunch &r, int dx, int dy)
{
    const double ddx = (double)dx, ddy = (double)dy;
    const double sx = fma(ddx, r.sAx, fma(ddy, r.sBx, r.sCx)), sy = fma(ddx, r.sAy, fma(ddy, r.sBy, r.sCy));
    // (a guard of 1e-9 pixels on the extent: points exactly on its edge are inside however the map was rounded)
    const double guard = 1e-9;
    SamplePoint p;
    p.outside = sx < -0.5 - guard || sx > r.W - 0.5 + guard || sy < -0.5 - guard || sy > r.H - 0.5 + guard;
    const double fx = floor(sx), fy = floor(sy);
    p.ix = p.outside ? 0 : (int)fx; p.iy = p.outside ? 0 : (int)fy;
    p.tx = (float)(sx - fx); p.ty = (float)(sy - fy);
    return p;
}

// dst pixels are written once and never read back: around the caches (with whole lines per wave, below: config 5's
// bilinear leg 1.01 -> 0.70 ms; either step alone: 0.89 / 1.03)
__device__ __forceinline__ void sample_store(float *p, float v) { __builtin_nontemporal_store(v, p); }

// rows of dst pixels per wave (a wave = 64 consecutive dx): with one row per wave the kernel was bound by the latency of a
// wave's life -- coordinates, two dependent loads, one store -- not by instructions or bytes
template <int MODE> struct SampleRows { static constexpr int value = MODE == AAI_MODE_BILINEAR ? 4 : 2; };      // (8 rows: 1.84 ms where 4 take 1.11)

template <int MODE, typename T>
__global__ __launch_bounds__(kBlock) void aai_sample_kernel(RotLaunch r, const T *__restrict__ src, ImageView sv,
                                                             float *__restrict__ dst, ImageView dv, const int *__restrict__ live)
{
    constexpr int R = SampleRows<MODE>::value;
    constexpr int N = MODE == AAI_MODE_BILINEAR ? 2 : 4, FIRST = MODE == AAI_MODE_BILINEAR ? 0 : -1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave-uniform: row addresses are scalar
    const int dy0 = r.dyBase + (blockIdx.y * 4 + wave) * R;
    if (dy0 >= r.dyEnd) return;
    const int nRows = min(R, r.dyEnd - dy0);
    const T *img = src + (int64_t)blockIdx.z * sv.imageStride;
    const int chan = r.chan > 1 ? r.chan : 1;          // interleaved channels share the taps' positions and weights
    float *outRow0 = dst + (int64_t)blockIdx.z * dv.imageStride + (int64_t)(dy0 - r.dyBase) * dv.rowStride;
    // A wave stores 64 consecutive pixels of a dst row.  Rows of a large output start at any multiple of 4 bytes, so
    // columns [64 b, 64 b + 64) straddle three 128-byte lines of which two are shared with the neighbouring workgroups --
    // which run on other XCDs, behind other L2s: 2.15 GB of config 5 went out at 2.3 TB/s whatever the kernel computed
    // (no loads, fp32 coordinates: the same 0.94 ms).  Each row therefore shifts its columns left to the 256-byte boundary
    // below them: every store of a wave is two whole lines (the grid has one more column of workgroups).
    int dxs[R];
    bool inRow[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int shift = chan == 1 ? (int)((reinterpret_cast<uintptr_t>(outRow0 + (int64_t)j * dv.rowStride) >> 2) & 63) : 0;
        dxs[j] = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63) - shift;
        inRow[j] = j < nRows && dxs[j] >= 0 && dxs[j] < r.dW;
    }
    if (live && chan == 1) {
        // the columns of this wave lie in the 16-column tiles 4 b - 4 ... 4 b + 3 of one tile row: all in a corner of the
        // rotated canvas -> zeros
        const int first = live[2 * (dy0 >> 4)], last = live[2 * (dy0 >> 4) + 1];
        if ((int)blockIdx.x * 4 + 3 < first || (int)blockIdx.x * 4 - 4 > last) {
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (inRow[j]) sample_store(outRow0 + (int64_t)j * dv.rowStride + dxs[j], 0.f);
            return;
        }
    }

    SamplePoint pt[R];
    bool whole = true;          // no tap column of this lane is clamped (or the point is outside / not in the row: no taps at all)
#pragma unroll
    for (int j = 0; j < R; ++j) {
        pt[j] = sample_point(r, inRow[j] ? dxs[j] : 0, dy0 + (j < nRows ? j : 0));
        if (!inRow[j]) pt[j].outside = true;
        whole = whole && (pt[j].outside || (pt[j].ix + FIRST >= 0 && pt[j].ix + FIRST + N <= r.W));
    }
    // The N taps of a tap row are neighbours in memory: one vector load per tap row instead of N scalar ones (neighbouring
    // lanes sample a fraction of a pixel apart along a slanted line, so every load instruction touches ~10 cache lines
    // and their number is what the texture path charges for: 16 -> 4 per bicubic sample), all R x N of them in flight
    // together.  Plain fp32 images below 4 GiB (a scalar base plus an unsigned 32-bit byte offset per lane), and only while
    // no lane of the wave has a clamped column.
    const bool below4G = (int64_t)r.H * sv.rowStride * (int64_t)sizeof(T) < ((int64_t)1 << 32);
    if (sizeof(T) == 4 && chan == 1 && below4G && __all(whole)) {
        const char *base = reinterpret_cast<const char *>(img) - (int64_t)r.srcRow0 * sv.rowStride * 4;
        const unsigned rowBytes = (unsigned)sv.rowStride * 4u;
        float t[R][N][N];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const unsigned colBytes = (unsigned)(pt[j].ix + FIRST) * 4u;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const unsigned row = (unsigned)min(max(pt[j].iy + FIRST + k, 0), r.H - 1);
                if (!pt[j].outside) load_taps<N>(reinterpret_cast<const float *>(base + (row * rowBytes + colBytes)), t[j][k]);
                else {
#pragma unroll
                    for (int i = 0; i < N; ++i) t[j][k][i] = 0.f;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            float v;
            if (MODE == AAI_MODE_BILINEAR) {
                const float top = fmaf(t[j][0][1] - t[j][0][0], pt[j].tx, t[j][0][0]), bot = fmaf(t[j][1][1] - t[j][1][0], pt[j].tx, t[j][1][0]);
                v = fmaf(bot - top, pt[j].ty, top);
            } else {
                float wx[4], wy[4];
                keys(pt[j].tx, wx); keys(pt[j].ty, wy);
                v = 0.f;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    float row = 0.f;
#pragma unroll
                    for (int i = 0; i < N; ++i) row = fmaf(wx[i], t[j][k][i], row);
                    v = fmaf(wy[k], row, v);
                }
            }
            if (inRow[j]) sample_store(outRow0 + (int64_t)j * dv.rowStride + dxs[j], pt[j].outside ? 0.f : v);
        }
        return;
    }
    // the general path, row by row: clamp-to-edge taps, typed sources, interleaved channels
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (!inRow[j]) continue;
        const SamplePoint p = pt[j];
        float *out = outRow0 + (int64_t)j * dv.rowStride + (int64_t)dxs[j] * chan;
        if (p.outside) {
            for (int c = 0; c < chan; ++c) sample_store(out + c, 0.f);
            continue;
        }
        int xo[N];
        int64_t yo[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            xo[i] = min(max(p.ix + FIRST + i, 0), r.W - 1) * chan;
            yo[i] = (int64_t)(min(max(p.iy + FIRST + i, 0), r.H - 1) - r.srcRow0) * sv.rowStride;
        }
        float wx[4], wy[4];
        if (MODE != AAI_MODE_BILINEAR) { keys(p.tx, wx); keys(p.ty, wy); }
        for (int c = 0; c < chan; ++c) {
            const T *ch = img + c;
            float v;
            if (MODE == AAI_MODE_BILINEAR) {
                const float v00 = (float)ch[yo[0] + xo[0]], v10 = (float)ch[yo[0] + xo[1]];
                const float v01 = (float)ch[yo[N - 1] + xo[0]], v11 = (float)ch[yo[N - 1] + xo[1]];
                // explicit fused multiply-adds: the same rounding whatever the compiler does with the channel loop
                const float top = fmaf(v10 - v00, p.tx, v00), bot = fmaf(v11 - v01, p.tx, v01);
                v = fmaf(bot - top, p.ty, top);
            } else {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    float row = 0.f;
#pragma unroll
                    for (int i = 0; i < N; ++i) row = fmaf(wx[i], (float)ch[yo[k] + xo[i]], row);
                    acc = fmaf(wy[k], row, acc);
                }
                v = acc;
            }
            sample_store(out + c, v);
        }
    }
}

}  // namespace

size_t rotated_flag_words(const RotLaunch &r)
{
    if (r.mode == AAI_MODE_BILINEAR || r.mode == AAI_MODE_BICUBIC) return 0;
    return (size_t)((r.dW + 15) / 16) * (size_t)((r.dH + 15) / 16) * (kRotBlock / 64);
}

hipError_t launch_knife_scan(const RotLaunch &r, unsigned long long *laneMasks, unsigned *counter, hipStream_t stream)
{
    if (r.dW <= 0 || r.dH <= 0) return hipSuccess;
    const int tileRows = (r.dH + 15) / 16;
    for (int t0 = 0; t0 < tileRows; t0 += kMaxGridY) {      // grid.y carries at most 65535 tiles
        dim3 grid((r.dW + 15) / 16, std::min(tileRows - t0, kMaxGridY), 1);
        hipLaunchKernelGGL(aai_knife_scan_kernel, grid, dim3(kRotBlock), 0, stream, r, laneMasks, counter, t0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// experiments build only (make exp): AAI_ROT_TUNE="runs=0|1 quad=0|1", read once per process
struct RotTune { int runs = -1, quad = -1; };
static const RotTune &rot_tune()
{
    static const RotTune t = [] {
        RotTune v;
        if (const char *env = experiment_env("AAI_ROT_TUNE")) {
            if (const char *p = strstr(env, "runs=")) v.runs = atoi(p + 5) != 0;
            if (const char *p = strstr(env, "quad=")) v.quad = atoi(p + 5) != 0;
        }
        return v;
    }();
    return t;
}

// the fp32 quad kernels serve this launch (RotLaunch::quad; area mode: plain and interleaved images; fast mode: plain images)
static bool quad_serves(const RotLaunch &r, int srcType, ImageView sv)
{
    return r.quad && rot_tune().quad != 0 && (r.mode == AAI_MODE_AREA || (r.mode == AAI_MODE_FAST && r.chan == 1)) && quad_can_address(r, srcType, sv);
}

// ... or the cell formulation does: the plan's scan was the cell scan (RotFlags::form) and the launch is one it takes
static bool cell_serves(const RotLaunch &r, int srcType, ImageView sv, const RotFlags &flags)
{
    return flags.form == ROT_FORM_CELL && rot_tune().quad != 0 && cell_can_serve(r, srcType, sv);
}

// ... or, for footprints wider than one 8 x 8 window, the quad formulation split into parts (aai_rotated_wide.hip)
static bool wide_serves(const RotLaunch &r, int srcType, ImageView sv)
{
    return rot_tune().quad != 0 && wide_can_serve(r, srcType, sv);
}

// one launch of at most 65535 tile rows (16-row tiles; the bicubic sampler: 8-row tiles)
template <typename T>
static hipError_t launch_rotated_band(const RotLaunch &r, const QuadMap &m, const T *src, int srcType, ImageView sv, float *dst, ImageView dv,
                                      int batch, const RotFlags &flags, hipStream_t stream, const char **kernelName)
{
    if (r.mode == AAI_MODE_BILINEAR || r.mode == AAI_MODE_BICUBIC) {
        const int tileRows = 4 * (r.mode == AAI_MODE_BILINEAR ? SampleRows<AAI_MODE_BILINEAR>::value : SampleRows<AAI_MODE_BICUBIC>::value);
        dim3 grid((r.dW + 63) / 64 + (r.chan > 1 ? 0 : 1), (r.dyEnd - r.dyBase + tileRows - 1) / tileRows, batch);    // (+1: rows shift their columns left to a 256-byte boundary)
        if (r.mode == AAI_MODE_BILINEAR) {
            if (kernelName) *kernelName = "aai_sample_kernel<bilinear>";
            hipLaunchKernelGGL((aai_sample_kernel<AAI_MODE_BILINEAR, T>), grid, dim3(kBlock), 0, stream, r, src, sv, dst, dv, flags.live);
        } else {
            if (kernelName) *kernelName = "aai_sample_kernel<bicubic>";
            hipLaunchKernelGGL((aai_sample_kernel<AAI_MODE_BICUBIC, T>), grid, dim3(kBlock), 0, stream, r, src, sv, dst, dv, flags.live);
        }
        return hipGetLastError();
    }
    dim3 grid((r.dW + 15) / 16, (r.dyEnd - r.dyBase + 15) / 16, batch);
    const RotTune &tune = rot_tune();
    const bool quad = quad_serves(r, srcType, sv);
    if (cell_serves(r, srcType, sv, flags)) {
        // the cell formulation: one lane per cell of the dst grid, every (dst, src) pair evaluated once
        if (kernelName) *kernelName = r.chan > 1 ? "aai_cell_multi_kernel<area, channels>" : "aai_cell_kernel<area>";
        return launch_cell(r, m, src, srcType, sv, dst, dv, batch, flags.count ? flags.masks : nullptr, stream);
    }
    if (wide_serves(r, srcType, sv)) {
        if (kernelName) *kernelName = r.mode == AAI_MODE_FAST ? "aai_wide_fast_kernel" : "aai_wide_kernel<area>";
        return launch_wide(r, m, src, srcType, sv, dst, dv, batch, flags.count ? flags.masks : nullptr, stream);
    }
    if (r.chan > 1 && quad) {
        // interleaved channels through the fp32 quad formulation: areas once per pair, applied to every channel
        if (kernelName) *kernelName = "aai_quad_multi_kernel<area, channels>";
        return launch_quad(r, m, src, srcType, sv, dst, dv, batch, flags.count ? flags.masks : nullptr, stream);
    } else if (r.chan > 1) {
        // interleaved channels: the same kernels with the areas shared between the channels
        if (r.mode == AAI_MODE_AREA && r.runs) {
            if (kernelName) *kernelName = "aai_rotated_runs_kernel<area, channels>";
            hipLaunchKernelGGL((aai_rotated_runs_kernel<T, true>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv);
        } else if (r.mode == AAI_MODE_FAST) {
            if (kernelName) *kernelName = "aai_rotated_kernel<fast, channels>";
            hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_FAST, false, T, true>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, nullptr, 0u);
        } else {
            if (kernelName) *kernelName = "aai_rotated_kernel<area, channels>";
            hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_AREA, false, T, true>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, nullptr, 0u);
        }
    } else if (r.mode == AAI_MODE_FAST && quad) {
        // centres in the dst square, fp32 in the dst frame; flagged pixels belong to the fix-up pass as in area mode
        if (kernelName) *kernelName = "aai_quad_fast_kernel";
        const hipError_t e = launch_quad(r, m, src, srcType, sv, dst, dv, batch, flags.count ? flags.masks : nullptr, stream, flags.live);
        if (kernelName && quad_kernel_note()) *kernelName = quad_kernel_note();      // (the LDS-staged form)
        return e;
    } else if (r.mode == AAI_MODE_FAST) {
        if (kernelName) *kernelName = "aai_rotated_kernel<fast>";
        hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_FAST, false, T>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, nullptr, 0u);
    } else if (quad) {
        // the fp32 quad formulation; the pixels flagged by the plan's scans are recomputed by the fix-up pass
        if (kernelName) *kernelName = "aai_quad_kernel<area>";
        return launch_quad(r, m, src, srcType, sv, dst, dv, batch, flags.count ? flags.masks : nullptr, stream, flags.live);
    } else {
        const int runs = tune.runs >= 0 ? (tune.runs && r.scale == 1) : r.runs;
        if (runs) {
            if (kernelName) *kernelName = "aai_rotated_runs_kernel<area>";
            hipLaunchKernelGGL((aai_rotated_runs_kernel<T, false>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv);
        } else {
            if (kernelName) *kernelName = "aai_rotated_kernel<area>";
            hipLaunchKernelGGL((aai_rotated_kernel<AAI_MODE_AREA, false, T>), grid, dim3(kRotBlock), 0, stream, r, src, sv, dst, dv, nullptr, 0u);
        }
    }
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_rotated_typed(const RotLaunch &r, const QuadMap &m, const T *src, int srcType, ImageView sv, float *dst, ImageView dv,
                                       int batch, const RotFlags &flags, hipStream_t stream, const char **kernelName)
{
    if (r.dW <= 0 || r.dyEnd <= r.dyBase || batch <= 0) return hipSuccess;
    // grid.y carries at most 65535 tiles: taller outputs (more than ~1M rows, 524k for the bicubic sampler) go in several launches
    const bool sampler = r.mode == AAI_MODE_BILINEAR || r.mode == AAI_MODE_BICUBIC;
    if (!sampler && flags.dense) {
        // (nearly) every pixel sits on a knife edge: the double-precision pass computes the whole image
        if (kernelName) *kernelName = r.mode == AAI_MODE_FAST ? "aai_rotated_kernel<fast, strict>" : "aai_rotated_kernel<area, strict>";
        for (int y0 = r.dyBase; y0 < r.dyEnd; y0 += kMaxGridY * 16) {
            RotLaunch rb = r;
            rb.dyBase = y0;
            rb.dyEnd = r.dyEnd - y0 > kMaxGridY * 16 ? y0 + kMaxGridY * 16 : r.dyEnd;
            launch_rotated_fixup(rb, batch, src, srcType, sv, dst + (int64_t)(y0 - r.dyBase) * dv.rowStride, dv, nullptr, 0u, stream);
        }
        return hipGetLastError();
    }
    const int maxRows = kMaxGridY * (r.mode == AAI_MODE_BICUBIC ? 8 : 16);
    // The double-precision pass over the pixels the plan's scans flagged (none for the samplers).  Behind the production
    // pass on the same stream -- or, when the production kernel is the quad kernel and skips those pixels, beside it on
    // the plan's side stream: fork before, join after.
    const bool fixup = !sampler && flags.count != 0;
    bool beside = fixup && flags.masks && flags.side && (quad_serves(r, srcType, sv) || cell_serves(r, srcType, sv, flags) || wide_serves(r, srcType, sv));
    if (beside) {
        // a caller's stream that is being captured into a graph must not pull the plan's shared side stream into the
        // capture (another thread may use it meanwhile): the fix-up pass then follows the production pass in-stream
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) beside = false;
    }
    hipError_t e = hipSuccess;
    if (beside) {
        e = hipEventRecord(flags.fork, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(flags.side, flags.fork, 0);
        if (e != hipSuccess) return e;
        if (!r.noFixup) launch_rotated_fixup(r, batch, src, srcType, sv, dst, dv, static_cast<const uint2 *>(flags.list), flags.count, flags.side);      // (AAI_POLICY_DIAG_NO_FIXUP)
        e = hipEventRecord(flags.join, flags.side);
        if (e != hipSuccess) return e;           // (nothing was enqueued on the side stream after the fork that the caller could race with)
    }
    for (int y0 = r.dyBase; y0 < r.dyEnd && e == hipSuccess; y0 += maxRows) {
        RotLaunch rb = r;
        rb.dyBase = y0;
        rb.dyEnd = r.dyEnd - y0 > maxRows ? y0 + maxRows : r.dyEnd;
        RotFlags fb = flags;
        if (!beside) fb.masks = nullptr;
        QuadMap mb = m;
        mb.tileFlags = fb.masks ? flags.tileFlags : nullptr;
        mb.tileFlagWords = flags.tileFlagWords;
        e = launch_rotated_band(rb, mb, src, srcType, sv, dst + (int64_t)(y0 - r.dyBase) * dv.rowStride, dv, batch, fb, stream, kernelName);
    }
    if (beside) {
        // whatever happened above, the side stream's writes to dst are ordered before anything the caller enqueues next
        const hipError_t j = hipStreamWaitEvent(stream, flags.join, 0);
        return e != hipSuccess ? e : j;
    }
    if (e != hipSuccess) return e;
    if (fixup && !r.noFixup) launch_rotated_fixup(r, batch, src, srcType, sv, dst, dv, static_cast<const uint2 *>(flags.list), flags.count, stream);
    return hipGetLastError();
}

hipError_t launch_rotated(const RotLaunch &r, const QuadMap &m, const void *src, int srcType, ImageView sv, float *dst, ImageView dv,
                          int batch, const RotFlags &flags, hipStream_t stream, const char **kernelName)
{
    switch (srcType) {
    case SRC_U8: return launch_rotated_typed(r, m, static_cast<const unsigned char *>(src), srcType, sv, dst, dv, batch, flags, stream, kernelName);
    case SRC_U16: return launch_rotated_typed(r, m, static_cast<const unsigned short *>(src), srcType, sv, dst, dv, batch, flags, stream, kernelName);
    default: return launch_rotated_typed(r, m, static_cast<const float *>(src), srcType, sv, dst, dv, batch, flags, stream, kernelName);
    }
}

}  // namespace aai
