// aai_capi.cpp -- the C ABI declared in include/aai.h: request validation, plan cache (device-side
// weight tables for the axis-aligned kernel), kernel dispatch, and the host-buffer convenience calls.
//
// There is deliberately no CPU implementation behind this ABI: without a HIP device every compute entry
// point fails with AAI_ERR_NO_DEVICE.  The CPU oracle under oracle/ is test infrastructure and is never
// linked or loaded here.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <list>
#include <mutex>
#include <string>
#include <vector>

#include "aai_kernels.hpp"

namespace {

thread_local std::string g_lastError;
thread_local std::string g_lastKernel;

int fail(int code, const std::string &msg)
{
    g_lastError = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    // a missing / unusable device is reported as such so callers can tell it from a kernel fault
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver || e == hipErrorNotInitialized)
        return fail(AAI_ERR_NO_DEVICE, m);
    return fail(AAI_ERR_HIP, m);
}

#define AAI_HIP(call)                                        \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return hip_fail(e__, #call);  \
    } while (0)

// ---- plan cache ------------------------------------------------------------------------------------
struct Plan {
    aai_request key{};
    int band0 = -1, band1 = -1;      // dst row band this plan serves (-1: the whole image)
    int channels = 1;                // interleaved channels the K1 tables were built for
    int srcRow0 = 0, srcRow1 = 0;     // source rows the band reads; the source pointer addresses row srcRow0
    int device = -1;
    aai::Geometry g;
    int kernel = 0;
    // K1
    aai::AxisTables tabs;
    aai::AxisEntry *dLane = nullptr, *dRow = nullptr;
    aai::AxisStrip *dStrips = nullptr;
    int tuneRows = 0, tuneNt = 0, tuneSwap = 0;      // K1 launch shape measured on this device (0 rows = built-in default)
    // K2/K3: the dst pixels flagged by the one-off scans (knife edges of the reference's classifier; decisions the fp32
    // quad kernel leaves to double precision) as a list of (dx, dy) the fix-up pass runs over; `dense` when there are
    // so many that the whole image takes that pass instead
    void *dList = nullptr;
    unsigned flaggedPixels = 0;
    bool dense = false;
    // quad kernel: the lane masks of the flagged pixels (it skips them) and the side stream the fix-up pass runs on
    unsigned long long *dMasks = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    ~Plan()
    {
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        if (dMasks) (void)hipFree(dMasks);
        if (dList) (void)hipFree(dList);
        if (dLane) (void)hipFree(dLane);
        if (dRow) (void)hipFree(dRow);
        if (dStrips) (void)hipFree(dStrips);
    }
};

std::mutex g_planMutex;
std::list<Plan> g_plans;              // most recently used first
constexpr size_t kMaxPlans = 32;
constexpr unsigned kMaxListedPixels = 1u << 24;      // beyond 16 M flagged pixels the whole image takes the double-precision pass

bool same_request(const aai_request &a, const aai_request &b)
{
    return a.mode == b.mode && a.policy == b.policy && a.src_width == b.src_width && a.src_height == b.src_height &&
           a.src_res_x == b.src_res_x && a.src_res_y == b.src_res_y && a.dst_res_x == b.dst_res_x &&
           a.dst_res_y == b.dst_res_y && a.src_iso_x == b.src_iso_x && a.src_iso_y == b.src_iso_y &&
           a.rotation_deg == b.rotation_deg;
}

int check_request(const aai_request *rq)
{
    if (!rq) return fail(AAI_ERR_BAD_ARGUMENT, "Null request.");
    if (rq->mode < AAI_MODE_AREA || rq->mode > AAI_MODE_BICUBIC) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown interpolation mode.");
    const int rule = rq->policy & ~AAI_POLICY_DOUBLE_PRECISION;
    if (rule != AAI_POLICY_REFERENCE && rule != AAI_POLICY_EXACT) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown weight policy.");
    return AAI_OK;
}

// K1 row bands keep one more source row on either side where the fix-up pass behind K1 may run (get_plan: verifyAxis):
// its window includes rows that only touch the dst pixel
int axis_band_margin(const aai_request &rq)
{
    return rq.mode == AAI_MODE_AREA || rq.mode == AAI_MODE_FAST ? 1 : 0;
}

int pick_kernel(const aai_request &rq, const aai::Geometry &g)
{
    if (rq.mode == AAI_MODE_BILINEAR || rq.mode == AAI_MODE_BICUBIC) return AAI_KERNEL_SAMPLE;
    if (g.axisAligned) return AAI_KERNEL_AXIS;
    return rq.mode == AAI_MODE_FAST ? AAI_KERNEL_FAST : AAI_KERNEL_ROTATED;
}

int resolved_kernel(const aai_request &rq, const aai::Geometry &g)
{
    int kernel = pick_kernel(rq, g);
    if (kernel == AAI_KERNEL_AXIS) {
        aai::AxisTables t;
        aai::build_axis_tables(g, rq.mode, t);
        if (t.wide) kernel = AAI_KERNEL_AXIS_WIDE;
    }
    return kernel;
}

void fill_layout(const aai::Geometry &g, int kernel, aai_layout *out)
{
    aai_layout l{};
    l.dst_width = g.dW; l.dst_height = g.dH;
    l.dst_iso_x = g.dIsoX; l.dst_iso_y = g.dIsoY;
    l.scale = g.scale; l.quadrant = g.quadrant;
    l.reduced_angle_deg = g.angle; l.side = g.side;
    l.kernel = kernel;
    *out = l;
}

// the argument block of K1 for a plan and a dst row stride (elements)
aai::AxisLaunch make_axis_launch(const Plan &p, int channels, int64_t dstStride)
{
    const aai::AxisTables &t = p.tabs;
    const aai::Geometry &g = p.g;
    aai::AxisLaunch a{};
    a.laneTab = p.dLane; a.rowTab = p.dRow; a.strips = p.dStrips;
    a.nA = t.nA; a.nB = t.nB; a.nStrips = (int)t.strips.size();
    a.srcW = g.W * channels; a.srcH = g.H;      // elements of a source row
    a.wide = t.wide ? 1 : 0;
    a.maxRowSpan = t.maxRowSpan;
    a.rowsShared = t.rowsShared ? 1 : 0;
    a.maxOutputsPerStrip = t.maxOutputsPerStrip;
    // (ka,kb) -> dst element: the lane axis is dst x unless the quadrant transposes; flips run an axis
    // backwards
    // (with interleaved channels a dst pixel is `channels` elements wide and lane entry ka = pixel * channels + channel)
    const int nApix = t.nA / channels;
    const int64_t sa = t.transposed ? dstStride : channels, sb = t.transposed ? channels : dstStride;
    a.outStrideA = t.flipA ? -sa : sa;
    a.outStrideB = t.flipB ? -sb : sb;
    a.outBase = (t.flipA ? (int64_t)(nApix - 1) * sa : 0) + (t.flipB ? (int64_t)(t.nB - 1) * sb : 0);
    a.transposed = t.transposed ? 1 : 0;
    a.tapStep = channels; a.outChan = channels;
    if (channels > 1 && !t.transposed && !t.flipA) { a.outStrideA = 1; a.outChan = 1; }     // lane order = dst element order
    a.tuneRows = p.tuneRows; a.tuneNt = p.tuneNt; a.tuneSwap = p.tuneSwap;
    return a;
}

// K1 is HBM-bound and its best launch shape moves with the box by a few per cent (the same binary measured 5.8 to 6.9
// TB/s across boxes of one pool: profiles/r01_axis_tune_sweep2.txt, profiles/r02_axis_autotune.txt), so a plan for a
// large streaming geometry times the shapes that ever win -- output rows per workgroup, nontemporal or cached loads,
// grid order -- once on this device, on scratch images larger than the Infinity Cache, and keeps the fastest.  Only the
// common family (plain fp32 images, footprints of >= 4 source rows, un-transposed quadrants, whole image).
void tune_axis_plan(Plan &p, int channels, int band0)
{
    static const bool enabled = [] { const char *e = getenv("AAI_AXIS_AUTOTUNE"); return !(e && atoi(e) == 0); }();
    const aai::AxisTables &t = p.tabs;
    const aai::Geometry &g = p.g;
    const size_t srcBytes = sizeof(float) * (size_t)g.W * g.H, dstBytes = sizeof(float) * (size_t)g.dW * g.dH;
    if (!enabled || channels != 1 || band0 >= 0 || t.wide || t.transposed || t.maxRowSpan < 4 || t.maxOutputsPerStrip > 64 ||
        srcBytes < ((size_t)64 << 20) || !dstBytes)
        return;
    const int images = (int)std::min<size_t>(8, std::max<size_t>(2, (((size_t)1 << 30) + srcBytes - 1) / srcBytes));
    float *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMalloc((void **)&src, srcBytes * images) == hipSuccess && hipMalloc((void **)&dst, dstBytes * images) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    for (int b = 0; b < images && ok; ++b)         // realistic data: the memory system's speed depends on what it moves
        ok = aai::launch_synth(src + (size_t)b * g.W * g.H, g.W, g.H, g.W, (uint64_t)b + 1, nullptr) == hipSuccess;
    struct Shape { int rows, nt, swap; float ms; };
    Shape shapes[] = {{1, 1, 0, 0.f}, {2, 1, 0, 0.f}, {1, 0, 0, 0.f}, {2, 0, 0, 0.f}, {4, 1, 0, 0.f}};
    const aai::ImageView sv{g.W, (int64_t)g.W * g.H}, dv{g.dW, (int64_t)g.dW * g.dH};
    // the shapes take turns, three rounds, two launches per turn; each keeps its fastest turn
    for (int round = 0; round < 4 && ok; ++round)
        for (Shape &sh : shapes) {
            p.tuneRows = sh.rows; p.tuneNt = sh.nt; p.tuneSwap = sh.swap;
            const aai::AxisLaunch a = make_axis_launch(p, 1, g.dW);
            float ms = 0.f;
            ok = ok && hipEventRecord(e0, nullptr) == hipSuccess &&
                 aai::launch_axis(a, src, aai::SRC_F32, sv, dst, dv, images, nullptr, nullptr) == hipSuccess &&
                 aai::launch_axis(a, src, aai::SRC_F32, sv, dst, dv, images, nullptr, nullptr) == hipSuccess &&
                 hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (ok && round > 0 && (sh.ms == 0.f || ms < sh.ms)) sh.ms = ms;       // round 0 warms up
        }
    p.tuneRows = 0; p.tuneNt = 0; p.tuneSwap = 0;
    if (ok) {
        const Shape *best = &shapes[0];
        for (const Shape &sh : shapes)
            if (sh.ms < best->ms * 0.99f) best = &sh;          // a later shape must win by more than the timing noise
        p.tuneRows = best->rows; p.tuneNt = best->nt; p.tuneSwap = best->swap;
    }
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
}

// Finds or builds the plan for (request, current device).  Returns a pointer valid until evicted; callers
// hold g_planMutex for the duration of the launch (launches only enqueue, so this is short).
int get_plan(const aai_request &rq, int band0, int band1, int channels, Plan **out)
{
    int dev = -1;
    AAI_HIP(hipGetDevice(&dev));
    {
        // Only K1's tables depend on the row band.  The per-pixel kernels take the band as launch parameters and their
        // one-off scans cover the whole image, so every band and every channel count of a rotated request shares one plan.
        aai::Geometry g0;
        std::string msg0;
        int rc0 = aai::make_geometry(rq, g0, msg0);
        if (rc0 != AAI_OK) return fail(rc0, msg0);
        if (pick_kernel(rq, g0) != AAI_KERNEL_AXIS) { band0 = band1 = -1; channels = 1; }
    }
    for (auto it = g_plans.begin(); it != g_plans.end(); ++it) {
        if (it->device == dev && it->band0 == band0 && it->band1 == band1 && it->channels == channels && same_request(it->key, rq)) {
            g_plans.splice(g_plans.begin(), g_plans, it);
            *out = &g_plans.front();
            return AAI_OK;
        }
    }
    aai::Geometry g;
    std::string msg;
    int rc = aai::make_geometry(rq, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);

    g_plans.emplace_front();
    Plan &p = g_plans.front();
    p.key = rq; p.band0 = band0; p.band1 = band1; p.channels = channels; p.device = dev; p.g = g; p.kernel = pick_kernel(rq, g);
    p.srcRow0 = 0; p.srcRow1 = g.H;
    if (p.kernel == AAI_KERNEL_AXIS) {
        aai::build_axis_tables(g, rq.mode, p.tabs, channels);
        if (band0 >= 0) aai::restrict_axis_tables_to_band(g, p.tabs, band0, band1, p.srcRow0, p.srcRow1, axis_band_margin(rq));
        if (p.tabs.wide) p.kernel = AAI_KERNEL_AXIS_WIDE;
        auto upload = [&](const void *h, size_t bytes, void **d) -> hipError_t {
            if (!bytes) { *d = nullptr; return hipSuccess; }
            hipError_t e = hipMalloc(d, bytes);
            if (e != hipSuccess) return e;
            return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        };
        hipError_t e = upload(p.tabs.lane.data(), p.tabs.lane.size() * sizeof(aai::AxisEntry), (void **)&p.dLane);
        if (e == hipSuccess) e = upload(p.tabs.row.data(), p.tabs.row.size() * sizeof(aai::AxisEntry), (void **)&p.dRow);
        if (e == hipSuccess) e = upload(p.tabs.strips.data(), p.tabs.strips.size() * sizeof(aai::AxisStrip), (void **)&p.dStrips);
        if (e != hipSuccess) { g_plans.pop_front(); return hip_fail(e, "uploading axis tables"); }
        if (p.kernel == AAI_KERNEL_AXIS) tune_axis_plan(p, channels, band0);
    }
    const bool axisKernel = p.kernel == AAI_KERNEL_AXIS || p.kernel == AAI_KERNEL_AXIS_WIDE;
    // K1's separable model against the reference's classifier (aai_axis_verify.hpp).  Both policies: they differ in the
    // corner-triangle rule of a slanted left/right edge only, which does not exist at multiples of 90 degrees.
    const bool verifyAxis = axisKernel && axis_band_margin(rq) != 0;
    if (p.kernel == AAI_KERNEL_ROTATED || p.kernel == AAI_KERNEL_FAST || verifyAxis) {
        // one-off scans of this geometry (rotated: aai_knife_scan_kernel, and aai_quad_scan_kernel where the fp32 quad
        // kernels serve it; axis-aligned: aai_axis_verify_kernel); keeps the list of flagged pixels only if there are any
        const aai::RotLaunch r = aai::make_rot_launch(g, rq.mode, rq.policy);
        const size_t waves = aai::rotated_flag_words(r);
        unsigned long long *dMasks = nullptr;
        unsigned *dCount = nullptr;
        unsigned count = 0;
        hipError_t e = hipSuccess;
        if (waves) {
            e = hipMalloc((void **)&dMasks, waves * sizeof(unsigned long long));
            if (e == hipSuccess) e = hipMalloc((void **)&dCount, sizeof(unsigned));
            if (e == hipSuccess) e = hipMemset(dCount, 0, sizeof(unsigned));
            if (e == hipSuccess) e = verifyAxis ? aai::launch_axis_verify(r, dMasks, dCount, nullptr) : aai::launch_knife_scan(r, dMasks, dCount, nullptr);
            if (e == hipSuccess && r.quad) e = aai::launch_quad_scan(r, dMasks, dCount, nullptr);
            if (e == hipSuccess) e = hipMemcpy(&count, dCount, sizeof(unsigned), hipMemcpyDeviceToHost);
            // (AAI_MAX_LISTED_PIXELS: test hook, lowers the threshold so that small geometries exercise the dense form)
            static const unsigned maxListed = [] { const char *v = getenv("AAI_MAX_LISTED_PIXELS"); return v ? (unsigned)strtoul(v, nullptr, 10) : kMaxListedPixels; }();
            if (e == hipSuccess && count > maxListed) { p.dense = true; count = 0; }
            if (e == hipSuccess && count) {
                e = hipMalloc(&p.dList, (size_t)count * 2 * sizeof(unsigned));
                if (e == hipSuccess) e = hipMemset(dCount, 0, sizeof(unsigned));
                if (e == hipSuccess) e = aai::launch_flag_list(dMasks, waves, (unsigned)((g.dW + 15) / 16), p.dList, dCount, count, nullptr);
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
            if (dCount) (void)hipFree(dCount);
            if (e == hipSuccess && count && r.quad) {
                // keep the masks: the quad kernel skips the flagged pixels and the fix-up pass runs beside it
                p.dMasks = dMasks;
                dMasks = nullptr;
                e = hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&p.fork, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&p.join, hipEventDisableTiming);
            }
            if (dMasks) (void)hipFree(dMasks);
        }
        if (e != hipSuccess) { g_plans.pop_front(); return hip_fail(e, verifyAxis ? "axis model scan" : "knife-edge scan"); }
        p.flaggedPixels = count;
    }
    while (g_plans.size() > kMaxPlans) g_plans.pop_back();
    *out = &p;
    return AAI_OK;
}

constexpr int kMaxGridZ = 65535;

// element offset into a typed source buffer
const void *src_at(const void *base, int srcType, int64_t elements)
{
    const int64_t esz = srcType == aai::SRC_U8 ? 1 : srcType == aai::SRC_U16 ? 2 : 4;
    return static_cast<const char *>(base) + elements * esz;
}

int enqueue(const aai_request &rq, int batch, const void *dSrc, int srcType, int64_t srcStride, int64_t srcImageStride,
            float *dDst, int64_t dstStride, int64_t dstImageStride, hipStream_t stream, int band0 = -1, int band1 = -1,
            int channels = 1)
{
    std::lock_guard<std::mutex> lock(g_planMutex);
    Plan *p = nullptr;
    int rc = get_plan(rq, band0, band1, channels, &p);
    if (rc != AAI_OK) return rc;
    const aai::Geometry &g = p->g;
    // strides are in elements; an interleaved pixel takes `channels` of them
    if (srcStride < (int64_t)g.W * channels) return fail(AAI_ERR_BAD_ARGUMENT, "Source stride smaller than the image width.");
    if (dstStride < (int64_t)g.dW * channels) return fail(AAI_ERR_BAD_ARGUMENT, "Destination stride smaller than the output width.");
    if (g.dW == 0 || g.dH == 0 || batch == 0) return AAI_OK;

    aai::ImageView sv{srcStride, srcImageStride}, dv{dstStride, dstImageStride};
    const char *name = "";
    hipError_t e;
    const bool axisKernel = p->kernel == AAI_KERNEL_AXIS || p->kernel == AAI_KERNEL_AXIS_WIDE;
    // per-pixel launch description: the rotated kernels, and the fix-up pass behind K1
    aai::RotLaunch r = aai::make_rot_launch(g, rq.mode, rq.policy);
    if (band0 >= 0) {
        int srcRow0 = p->srcRow0, srcRow1 = p->srcRow1;
        if (!axisKernel) aai::rotated_band_source_rows(g, band0, band1, p->kernel == AAI_KERNEL_SAMPLE, srcRow0, srcRow1);
        r.dyBase = band0; r.dyEnd = band1; r.srcRow0 = srcRow0;
    }
    r.chan = channels;
    if (axisKernel && !p->dense) {
        const aai::AxisLaunch a = make_axis_launch(*p, channels, dstStride);
        e = hipSuccess;
        for (int b0 = 0; b0 < batch && e == hipSuccess; b0 += kMaxGridZ) {      // grid.z carries the batch
            const void *s0 = src_at(dSrc, srcType, (int64_t)b0 * srcImageStride);
            float *d0 = dDst + (int64_t)b0 * dstImageStride;
            const int nb = std::min(batch - b0, kMaxGridZ);
            e = aai::launch_axis(a, s0, srcType, sv, d0, dv, nb, stream, &name);
            // dst pixels where the reference's classifier departs from the separable model (aai_axis_verify.hpp)
            if (e == hipSuccess && p->flaggedPixels) {
                aai::launch_rotated_fixup(r, nb, s0, srcType, sv, d0, dv, static_cast<const uint2 *>(p->dList), p->flaggedPixels, stream);
                e = hipGetLastError();
            }
        }
    } else {
        // (an axis-aligned geometry lands here when the separable model fails for most of its pixels: `dense`)
        const aai::QuadMap qm = aai::make_quad_map(g, srcStride, r.srcRow0, channels, srcType == aai::SRC_U8 ? 1 : srcType == aai::SRC_U16 ? 2 : 4);
        aai::RotFlags flags;
        flags.list = p->dList; flags.count = p->flaggedPixels; flags.dense = p->dense;
        flags.masks = p->dMasks; flags.side = p->side; flags.fork = p->fork; flags.join = p->join;
        e = hipSuccess;
        for (int b0 = 0; b0 < batch && e == hipSuccess; b0 += kMaxGridZ)
            e = aai::launch_rotated(r, qm, src_at(dSrc, srcType, (int64_t)b0 * srcImageStride), srcType, sv, dDst + (int64_t)b0 * dstImageStride, dv,
                                    std::min(batch - b0, kMaxGridZ), flags, stream, &name);
    }
    g_lastKernel = name;
    if (e != hipSuccess) return hip_fail(e, name);
    return AAI_OK;
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(AAI_ERR_NO_DEVICE, "No HIP device available: libaai_hip has no CPU fallback.");
    }
    return AAI_OK;
}

}  // namespace

// Host-buffer convenience path: H2D, one launch, D2H.  T is float or double (converted on the device).
template <typename T>
int resample_host(const aai_request *req, const T *src, int64_t srcStride, T *dst, int64_t dstStride, aai_layout *layout)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    aai::Geometry g;
    std::string msg;
    rc = aai::make_geometry(*req, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    if (!src || !dst) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    if (srcStride < g.W) return fail(AAI_ERR_BAD_ARGUMENT, "Source stride smaller than the image width.");
    if (dstStride < g.dW) return fail(AAI_ERR_BAD_ARGUMENT, "Destination stride smaller than the output width.");
    rc = require_device();
    if (rc != AAI_OK) return rc;

    const size_t nSrc = (size_t)g.W * g.H, nDst = (size_t)g.dW * g.dH;
    T *dSrcT = nullptr, *dDstT = nullptr;
    float *dSrc = nullptr, *dDst = nullptr;
    hipStream_t stream = nullptr;
    auto cleanup = [&]() {
        if (dSrcT) (void)hipFree(dSrcT);
        if (dDstT && (void *)dDstT != (void *)dDst) (void)hipFree(dDstT);
        if (dSrc && (void *)dSrc != (void *)dSrcT) (void)hipFree(dSrc);
        if (dDst) (void)hipFree(dDst);
    };
#define AAI_HIP_C(call)                                                        \
    do {                                                                       \
        hipError_t e__ = (call);                                               \
        if (e__ != hipSuccess) { cleanup(); return hip_fail(e__, #call); }     \
    } while (0)

    AAI_HIP_C(hipMalloc((void **)&dSrcT, sizeof(T) * nSrc));
    AAI_HIP_C(hipMemcpy2D(dSrcT, sizeof(T) * g.W, src, sizeof(T) * srcStride, sizeof(T) * g.W, g.H, hipMemcpyHostToDevice));
    if (nDst) AAI_HIP_C(hipMalloc((void **)&dDst, sizeof(float) * nDst));
    if (sizeof(T) == sizeof(float)) {
        dSrc = reinterpret_cast<float *>(dSrcT);
        dDstT = reinterpret_cast<T *>(dDst);
    } else {
        AAI_HIP_C(hipMalloc((void **)&dSrc, sizeof(float) * nSrc));
        AAI_HIP_C(aai::launch_f64_to_f32(reinterpret_cast<const double *>(dSrcT), dSrc, nSrc, stream));
        if (nDst) AAI_HIP_C(hipMalloc((void **)&dDstT, sizeof(T) * nDst));
    }
    if (nDst) {
        rc = enqueue(*req, 1, dSrc, aai::SRC_F32, g.W, 0, dDst, g.dW, 0, stream);
        if (rc != AAI_OK) { cleanup(); return rc; }
        if (sizeof(T) != sizeof(float))
            AAI_HIP_C(aai::launch_f32_to_f64(dDst, reinterpret_cast<double *>(dDstT), nDst, stream));
        AAI_HIP_C(hipStreamSynchronize(stream));
        AAI_HIP_C(hipMemcpy2D(dst, sizeof(T) * dstStride, dDstT, sizeof(T) * g.dW, sizeof(T) * g.dW, g.dH, hipMemcpyDeviceToHost));
    }
    cleanup();
#undef AAI_HIP_C
    if (layout) fill_layout(g, resolved_kernel(*req, g), layout);
    g_lastError.clear();
    return AAI_OK;
}

extern "C" {

int aai_version(void) { return AAI_VERSION_MAJOR * 1000 + AAI_VERSION_MINOR; }

const char *aai_last_error(void) { return g_lastError.c_str(); }

const char *aai_last_kernel(void) { return g_lastKernel.c_str(); }

const char *aai_error_string(int code)
{
    switch (code) {
    case AAI_OK: return "";
    case AAI_ERR_RESOLUTION_MISMATCH: return "Assumed X & Y resolution are same.";
    case AAI_ERR_RESOLUTION_NONPOSITIVE: return "0 or negative resolution is not acceptable.";
    case AAI_ERR_NO_ROWS: return "There is no data in src array.";
    case AAI_ERR_NO_COLUMNS: return "There is no data in the second dimension of src array.";
    case AAI_ERR_NONFINITE: return "Non-finite argument.";
    case AAI_ERR_BAD_ARGUMENT: return "Bad argument.";
    case AAI_ERR_TOO_LARGE: return "Image too large.";
    case AAI_ERR_NO_DEVICE: return "No HIP device available.";
    case AAI_ERR_HIP: return "HIP runtime error.";
    case AAI_ERR_EMPTY_OUTPUT: return "Output image would be empty.";
    default: return "Unknown error.";
    }
}

int aai_query(const aai_request *req, aai_layout *out)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (!out) return fail(AAI_ERR_BAD_ARGUMENT, "Null layout.");
    aai::Geometry g;
    std::string msg;
    rc = aai::make_geometry(*req, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    fill_layout(g, resolved_kernel(*req, g), out);
    g_lastError.clear();
    return AAI_OK;
}

int aai_device_count(int *count)
{
    if (!count) return fail(AAI_ERR_BAD_ARGUMENT, "Null count.");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return AAI_OK;
}

int aai_set_device(int ordinal)
{
    int rc = require_device();
    if (rc != AAI_OK) return rc;
    AAI_HIP(hipSetDevice(ordinal));
    return AAI_OK;
}

int aai_device_synchronize(void)
{
    int rc = require_device();
    if (rc != AAI_OK) return rc;
    AAI_HIP(hipDeviceSynchronize());
    return AAI_OK;
}

static int resample_batch_device_typed(const aai_request *req, int32_t batch, const void *d_src, int32_t src_dtype,
                                       int64_t src_stride, int64_t src_image_stride,
                                       float *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *stream)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (src_dtype != AAI_DTYPE_F32 && src_dtype != AAI_DTYPE_U8 && src_dtype != AAI_DTYPE_U16) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown source element type.");
    if (batch < 0) return fail(AAI_ERR_BAD_ARGUMENT, "Negative batch.");      // any size: enqueue() splits batches beyond the grid.z limit
    // argument errors are reported before the device is touched, like the reference reports them first
    {
        aai::Geometry g;
        std::string msg;
        rc = aai::make_geometry(*req, g, msg);
        if (rc != AAI_OK) return fail(rc, msg);
    }
    if (!d_src || !d_dst) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    rc = require_device();
    if (rc != AAI_OK) return rc;
    rc = enqueue(*req, batch, d_src, src_dtype, src_stride, src_image_stride, d_dst, dst_stride, dst_image_stride, (hipStream_t)stream);
    if (rc == AAI_OK) g_lastError.clear();
    return rc;
}

int aai_resample_batch_device_f32(const aai_request *req, int32_t batch,
                                  const float *d_src, int64_t src_stride, int64_t src_image_stride,
                                  float *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *stream)
{
    return resample_batch_device_typed(req, batch, d_src, AAI_DTYPE_F32, src_stride, src_image_stride, d_dst, dst_stride, dst_image_stride, stream);
}

int aai_resample_batch_device(const aai_request *req, int32_t batch, const void *d_src, int32_t src_dtype,
                              int64_t src_stride, int64_t src_image_stride,
                              float *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *stream)
{
    return resample_batch_device_typed(req, batch, d_src, src_dtype, src_stride, src_image_stride, d_dst, dst_stride, dst_image_stride, stream);
}

int aai_resample_batch_multi_device_f32(const aai_request *req, int32_t n_shards, const int32_t *devices, const int32_t *counts,
                                        const float *const *d_src, int64_t src_stride, int64_t src_image_stride,
                                        float *const *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *const *streams)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (n_shards < 0 || (n_shards > 0 && (!devices || !counts || !d_src || !d_dst))) return fail(AAI_ERR_BAD_ARGUMENT, "Bad shard description.");
    {
        aai::Geometry g;
        std::string msg;
        rc = aai::make_geometry(*req, g, msg);
        if (rc != AAI_OK) return fail(rc, msg);
    }
    for (int i = 0; i < n_shards; ++i) {
        if (counts[i] < 0) return fail(AAI_ERR_BAD_ARGUMENT, "Negative batch.");
        if (counts[i] > 0 && (!d_src[i] || !d_dst[i])) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    }
    rc = require_device();
    if (rc != AAI_OK) return rc;
    int home = 0;
    AAI_HIP(hipGetDevice(&home));
    for (int i = 0; i < n_shards && rc == AAI_OK; ++i) {
        if (counts[i] == 0) continue;
        const hipError_t e = hipSetDevice(devices[i]);
        if (e != hipSuccess) { rc = hip_fail(e, "hipSetDevice"); break; }
        rc = enqueue(*req, counts[i], d_src[i], aai::SRC_F32, src_stride, src_image_stride, d_dst[i], dst_stride, dst_image_stride,
                     streams ? (hipStream_t)streams[i] : nullptr);
    }
    (void)hipSetDevice(home);
    if (rc == AAI_OK) g_lastError.clear();
    return rc;
}

int aai_resample_device_f32(const aai_request *req, const float *d_src, int64_t src_stride,
                            float *d_dst, int64_t dst_stride, void *stream)
{
    return aai_resample_batch_device_f32(req, 1, d_src, src_stride, 0, d_dst, dst_stride, 0, stream);
}

int aai_band_source_rows(const aai_request *req, int32_t dst_row0, int32_t dst_row1, int32_t *src_row0, int32_t *src_row1)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (!src_row0 || !src_row1) return fail(AAI_ERR_BAD_ARGUMENT, "Null output pointer.");
    aai::Geometry g;
    std::string msg;
    rc = aai::make_geometry(*req, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    if (dst_row0 < 0 || dst_row1 > g.dH || dst_row0 >= dst_row1) return fail(AAI_ERR_BAD_ARGUMENT, "Band rows out of range.");
    const int kernel = pick_kernel(*req, g);
    int a = 0, b = g.H;
    if (kernel == AAI_KERNEL_AXIS) {
        aai::AxisTables t;
        aai::build_axis_tables(g, req->mode, t);
        aai::restrict_axis_tables_to_band(g, t, dst_row0, dst_row1, a, b, axis_band_margin(*req));
    } else {
        if (dst_row0 % 16 != 0) return fail(AAI_ERR_BAD_ARGUMENT, "Band start must be a multiple of 16 rows for rotated requests.");
        aai::rotated_band_source_rows(g, dst_row0, dst_row1, kernel == AAI_KERNEL_SAMPLE, a, b);
    }
    *src_row0 = a; *src_row1 = b;
    g_lastError.clear();
    return AAI_OK;
}

int aai_resample_band_device_f32(const aai_request *req, int32_t dst_row0, int32_t dst_row1,
                                 const float *d_src_rows, int64_t src_stride, float *d_dst_rows, int64_t dst_stride, void *stream)
{
    int32_t a, b;
    int rc = aai_band_source_rows(req, dst_row0, dst_row1, &a, &b);      // validates request and band
    if (rc != AAI_OK) return rc;
    if (!d_src_rows || !d_dst_rows) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    rc = require_device();
    if (rc != AAI_OK) return rc;
    rc = enqueue(*req, 1, d_src_rows, aai::SRC_F32, src_stride, 0, d_dst_rows, dst_stride, 0, (hipStream_t)stream, dst_row0, dst_row1);
    if (rc == AAI_OK) g_lastError.clear();
    return rc;
}

int aai_prepare(const aai_request *req, int32_t channels)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (channels < 1 || channels > 4) return fail(AAI_ERR_BAD_ARGUMENT, "Channels must be 1..4.");
    {
        aai::Geometry g;
        std::string msg;
        rc = aai::make_geometry(*req, g, msg);
        if (rc != AAI_OK) return fail(rc, msg);
    }
    rc = require_device();
    if (rc != AAI_OK) return rc;
    std::lock_guard<std::mutex> lock(g_planMutex);
    Plan *p = nullptr;
    rc = get_plan(*req, -1, -1, channels, &p);
    if (rc == AAI_OK) g_lastError.clear();
    return rc;
}

/* experiments only (tools/tune_axis.py); not declared in include/aai.h */
void aai_debug_axis_tune(const char *spec) { aai::set_axis_tune(spec); }

const char *aai_debug_plan_shape(const aai_request *req)
{
    // "kernel=K rows=R nt=N swap=S flagged=F dense=D" of the cached whole-image plan of this request on the current device
    // ("" when there is none)
    static thread_local std::string text;
    text.clear();
    int dev = -1;
    if (!req || hipGetDevice(&dev) != hipSuccess) return text.c_str();
    std::lock_guard<std::mutex> lock(g_planMutex);
    for (const Plan &p : g_plans)
        if (p.device == dev && p.band0 < 0 && p.channels == 1 && same_request(p.key, *req))
            text = "kernel=" + std::to_string(p.kernel) + " rows=" + std::to_string(p.tuneRows) + " nt=" + std::to_string(p.tuneNt) + " swap=" + std::to_string(p.tuneSwap) +
                   " flagged=" + std::to_string(p.flaggedPixels) + " dense=" + std::to_string(p.dense ? 1 : 0);
    return text.c_str();
}

int aai_synth_rows_device_f32(float *d_dst, int32_t width, int32_t height, int32_t row0, int32_t row1, int64_t stride, uint64_t seed, void *stream)
{
    if (!d_dst || width < 0 || height < 0 || row0 < 0 || row1 < row0 || row1 > height || stride < width)
        return fail(AAI_ERR_BAD_ARGUMENT, "Bad synthetic image arguments.");
    int rc = require_device();
    if (rc != AAI_OK) return rc;
    AAI_HIP(aai::launch_synth_rows(d_dst, width, height, row0, row1, stride, seed, (hipStream_t)stream));
    return AAI_OK;
}

int aai_synth_device_f32(float *d_dst, int32_t width, int32_t height, int64_t stride, uint64_t seed, void *stream)
{
    if (!d_dst || width < 0 || height < 0 || stride < width) return fail(AAI_ERR_BAD_ARGUMENT, "Bad synthetic image arguments.");
    int rc = require_device();
    if (rc != AAI_OK) return rc;
    AAI_HIP(aai::launch_synth(d_dst, width, height, stride, seed, (hipStream_t)stream));
    return AAI_OK;
}

int aai_resample_host(const aai_request *req, const void *src, int32_t src_dtype, int64_t src_stride,
                      float *dst, int64_t dst_stride, aai_layout *layout)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    const size_t esz = src_dtype == AAI_DTYPE_F32 ? 4 : src_dtype == AAI_DTYPE_U8 ? 1 : src_dtype == AAI_DTYPE_U16 ? 2 : 0;
    if (!esz) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown source element type.");
    aai::Geometry g;
    std::string msg;
    rc = aai::make_geometry(*req, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    if (!src || !dst) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    if (src_stride < g.W) return fail(AAI_ERR_BAD_ARGUMENT, "Source stride smaller than the image width.");
    if (dst_stride < g.dW) return fail(AAI_ERR_BAD_ARGUMENT, "Destination stride smaller than the output width.");
    rc = require_device();
    if (rc != AAI_OK) return rc;
    void *dSrc = nullptr;
    float *dDst = nullptr;
    const size_t nDst = (size_t)g.dW * g.dH;
    auto cleanup = [&]() { if (dSrc) (void)hipFree(dSrc); if (dDst) (void)hipFree(dDst); };
#define AAI_HIP_C(call)                                                        \
    do {                                                                       \
        hipError_t e__ = (call);                                               \
        if (e__ != hipSuccess) { cleanup(); return hip_fail(e__, #call); }     \
    } while (0)
    AAI_HIP_C(hipMalloc(&dSrc, esz * (size_t)g.W * g.H));      // no padding: every kernel clamps its vector loads into the image
    AAI_HIP_C(hipMemcpy2D(dSrc, esz * g.W, src, esz * src_stride, esz * g.W, g.H, hipMemcpyHostToDevice));
    if (nDst) {
        AAI_HIP_C(hipMalloc((void **)&dDst, sizeof(float) * nDst));
        rc = enqueue(*req, 1, dSrc, src_dtype, g.W, 0, dDst, g.dW, 0, nullptr);
        if (rc != AAI_OK) { cleanup(); return rc; }
        AAI_HIP_C(hipStreamSynchronize(nullptr));
        AAI_HIP_C(hipMemcpy2D(dst, sizeof(float) * dst_stride, dDst, sizeof(float) * g.dW, sizeof(float) * g.dW, g.dH, hipMemcpyDeviceToHost));
    }
    cleanup();
#undef AAI_HIP_C
    if (layout) fill_layout(g, resolved_kernel(*req, g), layout);
    g_lastError.clear();
    return AAI_OK;
}

int aai_resample_interleaved_device(const aai_request *req, int32_t batch, int32_t channels,
                                    const void *d_src, int32_t src_dtype, int64_t src_stride, int64_t src_image_stride,
                                    float *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *stream)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (channels < 1 || channels > 4) return fail(AAI_ERR_BAD_ARGUMENT, "Channels must be 1..4.");
    if (batch < 0) return fail(AAI_ERR_BAD_ARGUMENT, "Negative batch.");
    if (src_dtype != AAI_DTYPE_F32 && src_dtype != AAI_DTYPE_U8 && src_dtype != AAI_DTYPE_U16) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown source element type.");
    aai::Geometry g;
    std::string msg;
    rc = aai::make_geometry(*req, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    if ((int64_t)g.W * channels > INT32_MAX / 2 || (int64_t)g.dW * channels > INT32_MAX / 2) return fail(AAI_ERR_TOO_LARGE, "Image too large.");
    if (batch > 0 && g.dW > 0 && g.dH > 0 && (!d_src || !d_dst)) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    rc = require_device();
    if (rc != AAI_OK) return rc;
    rc = enqueue(*req, batch, d_src, src_dtype, src_stride, src_image_stride, d_dst, dst_stride, dst_image_stride, (hipStream_t)stream, -1, -1, channels);
    if (rc == AAI_OK) g_lastError.clear();
    return rc;
}

int aai_resample_interleaved_host(const aai_request *req, int32_t channels, const void *src, int32_t src_dtype, int64_t src_stride,
                                  float *dst, int64_t dst_stride, aai_layout *layout)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (channels < 1 || channels > 4) return fail(AAI_ERR_BAD_ARGUMENT, "Channels must be 1..4.");
    const size_t esz = src_dtype == AAI_DTYPE_F32 ? 4 : src_dtype == AAI_DTYPE_U8 ? 1 : src_dtype == AAI_DTYPE_U16 ? 2 : 0;
    if (!esz) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown source element type.");
    aai::Geometry g;
    std::string msg;
    rc = aai::make_geometry(*req, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    if (!src || !dst) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    const int64_t rowIn = (int64_t)g.W * channels, rowOut = (int64_t)g.dW * channels;      // elements per dense row
    if (rowIn > INT32_MAX / 2 || rowOut > INT32_MAX / 2) return fail(AAI_ERR_TOO_LARGE, "Image too large.");
    if (src_stride < rowIn) return fail(AAI_ERR_BAD_ARGUMENT, "Source stride smaller than the image width.");
    if (dst_stride < rowOut) return fail(AAI_ERR_BAD_ARGUMENT, "Destination stride smaller than the output width.");
    rc = require_device();
    if (rc != AAI_OK) return rc;
    void *dSrc = nullptr;
    float *dDst = nullptr;
    const size_t nDst = (size_t)rowOut * g.dH;
    auto cleanup = [&]() { if (dSrc) (void)hipFree(dSrc); if (dDst) (void)hipFree(dDst); };
#define AAI_HIP_C(call)                                                        \
    do {                                                                       \
        hipError_t e__ = (call);                                               \
        if (e__ != hipSuccess) { cleanup(); return hip_fail(e__, #call); }     \
    } while (0)
    AAI_HIP_C(hipMalloc(&dSrc, esz * (size_t)rowIn * g.H));
    AAI_HIP_C(hipMemcpy2D(dSrc, esz * rowIn, src, esz * src_stride, esz * rowIn, g.H, hipMemcpyHostToDevice));
    if (nDst) {
        AAI_HIP_C(hipMalloc((void **)&dDst, sizeof(float) * nDst));
        rc = enqueue(*req, 1, dSrc, src_dtype, rowIn, 0, dDst, rowOut, 0, nullptr, -1, -1, channels);
        if (rc != AAI_OK) { cleanup(); return rc; }
        AAI_HIP_C(hipStreamSynchronize(nullptr));
        AAI_HIP_C(hipMemcpy2D(dst, sizeof(float) * dst_stride, dDst, sizeof(float) * rowOut, sizeof(float) * rowOut, g.dH, hipMemcpyDeviceToHost));
    }
    cleanup();
#undef AAI_HIP_C
    if (layout) fill_layout(g, resolved_kernel(*req, g), layout);
    g_lastError.clear();
    return AAI_OK;
}

int aai_host_alloc(void **ptr, uint64_t bytes)
{
    if (!ptr) return fail(AAI_ERR_BAD_ARGUMENT, "Null pointer.");
    int rc = require_device();
    if (rc != AAI_OK) return rc;
    AAI_HIP(hipHostMalloc(ptr, bytes ? (size_t)bytes : 1, hipHostMallocDefault));
    g_lastError.clear();
    return AAI_OK;
}

int aai_host_free(void *ptr)
{
    if (!ptr) return AAI_OK;
    AAI_HIP(hipHostFree(ptr));
    return AAI_OK;
}

namespace {

// Device slots of the pipelined host-batch entry, kept between calls (allocating and freeing ~100 MB buffers costs
// about as much as moving one 8-bit image over PCIe).  One pool per process; calls are serialised on its mutex.
constexpr int kSlots = 3;
struct SlotPool {
    int device = -1;
    size_t srcBytes = 0, dstBytes = 0;
    hipStream_t streams[kSlots] = {nullptr, nullptr, nullptr};
    void *dSrc[kSlots] = {nullptr, nullptr, nullptr};
    float *dDst[kSlots] = {nullptr, nullptr, nullptr};
    void release()
    {
        for (int s = 0; s < kSlots; ++s) {
            if (streams[s]) { (void)hipStreamSynchronize(streams[s]); (void)hipStreamDestroy(streams[s]); streams[s] = nullptr; }
            if (dSrc[s]) { (void)hipFree(dSrc[s]); dSrc[s] = nullptr; }
            if (dDst[s]) { (void)hipFree(dDst[s]); dDst[s] = nullptr; }
        }
        srcBytes = dstBytes = 0; device = -1;
    }
    hipError_t reserve(size_t needSrc, size_t needDst)
    {
        int dev = -1;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev == device && needSrc <= srcBytes && needDst <= dstBytes) return hipSuccess;
        release();
        for (int s = 0; s < kSlots && e == hipSuccess; ++s) {
            e = hipStreamCreateWithFlags(&streams[s], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipMalloc(&dSrc[s], needSrc);
            if (e == hipSuccess) e = hipMalloc((void **)&dDst[s], needDst);
        }
        if (e != hipSuccess) { release(); return e; }
        device = dev; srcBytes = needSrc; dstBytes = needDst;
        return hipSuccess;
    }
};
std::mutex g_slotMutex;
SlotPool g_slots;

// page-locked (hipHostMalloc / hipHostRegister) memory copies asynchronously; anything else is staged by the runtime
bool is_page_locked(const void *p)
{
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

}  // namespace

int aai_resample_batch_host(const aai_request *req, int32_t batch, const void *src, int32_t src_dtype,
                            int64_t src_stride, int64_t src_image_stride,
                            float *dst, int64_t dst_stride, int64_t dst_image_stride, aai_layout *layout)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    const size_t esz = src_dtype == AAI_DTYPE_F32 ? 4 : src_dtype == AAI_DTYPE_U8 ? 1 : src_dtype == AAI_DTYPE_U16 ? 2 : 0;
    if (!esz) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown source element type.");
    if (batch < 0) return fail(AAI_ERR_BAD_ARGUMENT, "Negative batch.");
    aai::Geometry g;
    std::string msg;
    rc = aai::make_geometry(*req, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    if (batch > 0 && (!src || !dst)) return fail(AAI_ERR_BAD_ARGUMENT, "Null image pointer.");
    if (src_stride < g.W) return fail(AAI_ERR_BAD_ARGUMENT, "Source stride smaller than the image width.");
    if (dst_stride < g.dW) return fail(AAI_ERR_BAD_ARGUMENT, "Destination stride smaller than the output width.");
    rc = require_device();
    if (rc != AAI_OK) return rc;

    const size_t nDst = (size_t)g.dW * g.dH;
    if (batch > 0 && nDst) {
        std::lock_guard<std::mutex> lock(g_slotMutex);
        SlotPool &p = g_slots;
        AAI_HIP(p.reserve(esz * (size_t)g.W * g.H, sizeof(float) * nDst));
        // Pageable buffers: the runtime's blocking copy (pinned bounce buffers, double-buffered) is its fastest path
        // and the asynchronous one much slower, so only page-locked buffers are copied asynchronously.
        const bool asyncUp = is_page_locked(src), asyncDown = is_page_locked(dst);
        const char *srcBytes = static_cast<const char *>(src);
        hipError_t e = hipSuccess;
        for (int b = 0; b < batch && e == hipSuccess; ++b) {
            const int s = b % kSlots;
            hipStream_t st = p.streams[s];
            // stream order protects the slot: this upload waits for the download of image b - kSlots.  Dense images
            // go as one linear copy (the 2-D path copies row by row and is several times slower).
            const char *hSrc = srcBytes + esz * (size_t)b * src_image_stride;
            float *hDst = dst + (size_t)b * dst_image_stride;
            if (!asyncUp) e = hipStreamSynchronize(st);          // a blocking copy does not wait for the slot's stream
            if (e != hipSuccess) break;
            if (src_stride == g.W) {
                e = asyncUp ? hipMemcpyAsync(p.dSrc[s], hSrc, esz * (size_t)g.W * g.H, hipMemcpyHostToDevice, st)
                            : hipMemcpy(p.dSrc[s], hSrc, esz * (size_t)g.W * g.H, hipMemcpyHostToDevice);
            } else {
                e = asyncUp ? hipMemcpy2DAsync(p.dSrc[s], esz * g.W, hSrc, esz * src_stride, esz * g.W, g.H, hipMemcpyHostToDevice, st)
                            : hipMemcpy2D(p.dSrc[s], esz * g.W, hSrc, esz * src_stride, esz * g.W, g.H, hipMemcpyHostToDevice);
            }
            if (e != hipSuccess) break;
            rc = enqueue(*req, 1, p.dSrc[s], src_dtype, g.W, 0, p.dDst[s], g.dW, 0, st);
            if (rc != AAI_OK) break;
            if (!asyncDown) e = hipStreamSynchronize(st);
            if (e != hipSuccess) break;
            if (dst_stride == g.dW) {
                e = asyncDown ? hipMemcpyAsync(hDst, p.dDst[s], sizeof(float) * nDst, hipMemcpyDeviceToHost, st)
                              : hipMemcpy(hDst, p.dDst[s], sizeof(float) * nDst, hipMemcpyDeviceToHost);
            } else {
                e = asyncDown ? hipMemcpy2DAsync(hDst, sizeof(float) * dst_stride, p.dDst[s], sizeof(float) * g.dW, sizeof(float) * g.dW, g.dH, hipMemcpyDeviceToHost, st)
                              : hipMemcpy2D(hDst, sizeof(float) * dst_stride, p.dDst[s], sizeof(float) * g.dW, sizeof(float) * g.dW, g.dH, hipMemcpyDeviceToHost);
            }
        }
        for (int s = 0; s < kSlots; ++s) {
            const hipError_t e2 = hipStreamSynchronize(p.streams[s]);
            if (e == hipSuccess) e = e2;
        }
        if (rc != AAI_OK) return rc;
        if (e != hipSuccess) return hip_fail(e, "aai_resample_batch_host");
    }
    if (layout) fill_layout(g, resolved_kernel(*req, g), layout);
    g_lastError.clear();
    return AAI_OK;
}

int aai_resample_f32(const aai_request *req, const float *src, int64_t src_stride, float *dst, int64_t dst_stride, aai_layout *layout)
{
    return resample_host<float>(req, src, src_stride, dst, dst_stride, layout);
}

int aai_resample_f64(const aai_request *req, const double *src, int64_t src_stride, double *dst, int64_t dst_stride, aai_layout *layout)
{
    return resample_host<double>(req, src, src_stride, dst, dst_stride, layout);
}

}  // extern "C"
