// aai_engine.cpp -- the host-side engine behind the C ABI: per-thread error state, request checks, the plan cache
// (K1 tables and launch-shape measurement, the one-off scans of a geometry and their fix-up lists) and the dispatch of a
// request onto the kernels.  aai_capi.cpp holds the extern "C" entry points and the host-buffer paths built on this.
//
// There is deliberately no CPU implementation here: without a HIP device every compute entry point fails with
// AAI_ERR_NO_DEVICE.  The CPU oracle under oracle/ is test infrastructure and is never linked or loaded.
#include "aai_engine.hpp"

#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <map>
#include <tuple>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace aai {
namespace engine {

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

// ---- plan cache ------------------------------------------------------------------------------------
std::mutex g_planMutex;
std::list<PlanRef> &plan_cache()
{
    static std::list<PlanRef> *cache = new std::list<PlanRef>();      // never destroyed: no HIP calls from static destructors
    return *cache;
}
static std::mutex g_poolMutex;
static std::map<int, DevicePool *> &pools()
{
    static std::map<int, DevicePool *> *m = new std::map<int, DevicePool *>();
    return *m;
}
DevicePool &device_pool(int device)
{
    std::lock_guard<std::mutex> lock(g_poolMutex);
    DevicePool *&p = pools()[device];
    if (!p) p = new DevicePool();
    return *p;
}
// the pool's build stream, created on first use (callers hold no lock)
static hipError_t pool_build_stream(DevicePool &pool, hipStream_t *out)
{
    std::lock_guard<std::mutex> lock(pool.m);
    if (!pool.build) {
        const hipError_t e = hipStreamCreateWithFlags(&pool.build, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = pool.build;
    return hipSuccess;
}
// the side streams and their events; under pool.m.  A failure leaves the pool without them (passes then run in-stream).
static void pool_create_sides(DevicePool &pool)
{
    hipError_t e = hipSuccess;
    for (int k = 0; k < DevicePool::kSideSlots && e == hipSuccess; ++k) {
        // (Default priority.  Created with the highest priority -- so that the pass's few hundred short workgroups would go first
        // -- the mere existence of such streams cost EVERY kernel of the process a third to a half of its rate: config 3 162 ->
        // 270 us, its fast mode 98 -> 217 us, 8:1 171 -> 329 us: profiles/r04_fixup_beside.txt.)
        e = hipStreamCreateWithFlags(&pool.side[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pool.fork[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pool.join[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) pool.sideReady = true;
    else { (void)hipGetLastError(); pool.sideFailed = true; }
}

void drop_plans()
{
    std::list<PlanRef> gone;
    {
        std::lock_guard<std::mutex> lock(g_planMutex);
        gone.swap(plan_cache());
    }
    gone.clear();                     // frees device memory of every plan nobody is launching from
    // ... and the pools' streams and events (created again on the next need)
    std::lock_guard<std::mutex> lock(g_poolMutex);
    int current = -1;
    (void)hipGetDevice(&current);
    for (auto &kv : pools()) {
        DevicePool &pool = *kv.second;
        std::lock_guard<std::mutex> plock(pool.m);
        if (hipSetDevice(kv.first) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (pool.build) { (void)hipStreamSynchronize(pool.build); (void)hipStreamDestroy(pool.build); pool.build = nullptr; }
        for (int k = 0; k < DevicePool::kSideSlots; ++k) {
            if (pool.side[k]) { (void)hipStreamSynchronize(pool.side[k]); (void)hipStreamDestroy(pool.side[k]); pool.side[k] = nullptr; }
            if (pool.fork[k]) { (void)hipEventDestroy(pool.fork[k]); pool.fork[k] = nullptr; }
            if (pool.join[k]) { (void)hipEventDestroy(pool.join[k]); pool.join[k] = nullptr; }
        }
        pool.sideReady = false; pool.sideFailed = false; pool.besideLaunches = 0;
    }
    if (current >= 0) (void)hipSetDevice(current);
}
constexpr size_t kMaxPlans = 32;       // per device
constexpr unsigned kMaxListedPixels = 1u << 24;      // beyond 16 M flagged pixels the whole image takes the double-precision pass

bool same_request(const aai_request &a, const aai_request &b)
{
    // (AAI_POLICY_DIAG_NO_FIXUP is a property of a launch, not of the plan: with and without it a request shares one plan)
    return a.mode == b.mode && (a.policy & ~AAI_POLICY_DIAG_NO_FIXUP) == (b.policy & ~AAI_POLICY_DIAG_NO_FIXUP) && a.src_width == b.src_width && a.src_height == b.src_height &&
           a.src_res_x == b.src_res_x && a.src_res_y == b.src_res_y && a.dst_res_x == b.dst_res_x &&
           a.dst_res_y == b.dst_res_y && a.src_iso_x == b.src_iso_x && a.src_iso_y == b.src_iso_y &&
           a.rotation_deg == b.rotation_deg;
}

int check_request(const aai_request *rq)
{
    if (!rq) return fail(AAI_ERR_BAD_ARGUMENT, "Null request.");
    if (rq->mode < AAI_MODE_AREA || rq->mode > AAI_MODE_BICUBIC) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown interpolation mode.");
    const int rule = rq->policy & AAI_POLICY_RULE_MASK;
    const int known = AAI_POLICY_RULE_MASK | AAI_POLICY_DOUBLE_PRECISION | AAI_POLICY_PREFER_CELL | AAI_POLICY_DIAG_NO_FIXUP;
    if ((rule != AAI_POLICY_REFERENCE && rule != AAI_POLICY_EXACT) || (rq->policy & ~known)) return fail(AAI_ERR_BAD_ARGUMENT, "Unknown weight policy.");
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
static aai::AxisLaunch make_axis_launch(const Plan &p, int channels, int64_t dstStride)
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
// TB/s across boxes of one pool: profiles/r01_axis_tune_sweep2.txt, profiles/r02_axis_autotune.txt), so the first plan of
// a CLASS of large streaming geometries on a device times the shapes that ever win -- output rows per workgroup,
// nontemporal or cached loads -- on scratch images larger than the Infinity Cache and keeps the fastest; later plans of
// the class (same device, same rows per footprint, same row sharing, same width class) take the result from a cache.
// Bounds: plain fp32 images, footprints of >= 4 source rows, un-transposed quadrants, whole image; scratch of at most
// ~1.25 GiB of source copies plus their outputs (images too large for that take the class's cached shape or the built-in
// one); a private stream; about 10 ms.  AAI_AXIS_AUTOTUNE=0 disables it.  (Documented in include/aai.h at aai_prepare.)
struct TuneKey {
    int device, rowSpan, rowsShared, widthClass;
    bool operator<(const TuneKey &o) const
    {
        return std::tie(device, rowSpan, rowsShared, widthClass) < std::tie(o.device, o.rowSpan, o.rowsShared, o.widthClass);
    }
};
struct TuneShape { int rows, nt, swap; };
static std::mutex g_tuneMutex;
static std::map<TuneKey, TuneShape> &tune_cache()
{
    static std::map<TuneKey, TuneShape> *m = new std::map<TuneKey, TuneShape>();
    return *m;
}

static void tune_axis_plan(Plan &p, int channels, int band0, hipStream_t stream)
{
    static const bool enabled = [] { const char *e = getenv("AAI_AXIS_AUTOTUNE"); return !(e && atoi(e) == 0); }();
    const aai::AxisTables &t = p.tabs;
    const aai::Geometry &g = p.g;
    const size_t srcBytes = sizeof(float) * (size_t)g.W * g.H, dstBytes = sizeof(float) * (size_t)g.dW * g.dH;
    if (!enabled || channels != 1 || band0 >= 0 || t.wide || t.transposed || t.maxRowSpan < 4 || t.maxOutputsPerStrip > 64 ||
        srcBytes < ((size_t)64 << 20) || !dstBytes)
        return;
    int widthClass = 0;
    while ((g.W >> widthClass) > 1) ++widthClass;
    const TuneKey key{p.device, std::min(t.maxRowSpan, 64), t.rowsShared ? 1 : 0, widthClass};
    {
        std::lock_guard<std::mutex> lock(g_tuneMutex);
        auto it = tune_cache().find(key);
        if (it != tune_cache().end()) {
            p.tuneRows = it->second.rows; p.tuneNt = it->second.nt; p.tuneSwap = it->second.swap; p.tuneSource = 2;
            return;
        }
    }
    constexpr size_t kScratchCap = (size_t)5 << 28;           // 1.25 GiB of source copies
    if (2 * srcBytes > kScratchCap) return;                   // too large to measure within the cap: built-in shape
    // at least two images and at least 512 MiB of source per launch: twice the 256 MiB Infinity Cache, so every launch streams
    // from HBM (1 GiB, as bench.py uses, measured the same shapes; half the scratch halves the cost of this measurement)
    const int images = (int)std::min<size_t>(8, std::max<size_t>(2, std::min((((size_t)1 << 29) + srcBytes - 1) / srcBytes, kScratchCap / srcBytes)));
    float *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMalloc((void **)&src, srcBytes * images) == hipSuccess && hipMalloc((void **)&dst, dstBytes * images) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    for (int b = 0; b < images && ok; ++b)         // realistic data: the memory system's speed depends on what it moves
        ok = aai::launch_synth(src + (size_t)b * g.W * g.H, g.W, g.H, g.W, (uint64_t)b + 1, stream) == hipSuccess;
    struct Shape { int rows, nt, swap; float ms; };
    Shape shapes[] = {{1, 1, 0, 0.f}, {2, 1, 0, 0.f}, {1, 0, 0, 0.f}, {2, 0, 0, 0.f}, {4, 1, 0, 0.f}};
    const aai::ImageView sv{g.W, (int64_t)g.W * g.H}, dv{g.dW, (int64_t)g.dW * g.dH};
    // the shapes take turns, two rounds after a warm-up round, two launches per turn; each keeps its faster turn
    for (int round = 0; round < 3 && ok; ++round)
        for (Shape &sh : shapes) {
            p.tuneRows = sh.rows; p.tuneNt = sh.nt; p.tuneSwap = sh.swap;
            const aai::AxisLaunch a = make_axis_launch(p, 1, g.dW);
            float ms = 0.f;
            ok = ok && hipEventRecord(e0, stream) == hipSuccess &&
                 aai::launch_axis(a, src, aai::SRC_F32, sv, dst, dv, images, stream, nullptr) == hipSuccess &&
                 aai::launch_axis(a, src, aai::SRC_F32, sv, dst, dv, images, stream, nullptr) == hipSuccess &&
                 hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (ok && round > 0 && (sh.ms == 0.f || ms < sh.ms)) sh.ms = ms;       // round 0 warms up
        }
    p.tuneRows = 0; p.tuneNt = 0; p.tuneSwap = 0;
    if (ok) {
        const Shape *best = &shapes[0];
        for (const Shape &sh : shapes)
            if (sh.ms < best->ms * 0.99f) best = &sh;          // a later shape must win by more than the timing noise
        p.tuneRows = best->rows; p.tuneNt = best->nt; p.tuneSwap = best->swap; p.tuneSource = 1;
        std::lock_guard<std::mutex> lock(g_tuneMutex);
        tune_cache()[key] = TuneShape{best->rows, best->nt, best->swap};
    }
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
}

// Everything a fresh plan needs from the device; the caller holds p.build, not the cache's lock.
static int build_plan(Plan &p, hipStream_t bs)
{
    const aai_request &rq = p.key;
    const aai::Geometry &g = p.g;
    const int band0 = p.band0, band1 = p.band1, channels = p.channels, form = p.form;
    // AAI_TRACE_PLAN=1: stage timings of every plan build on stderr (tools/plan_time.py)
    static const bool trace = [] { const char *v = getenv("AAI_TRACE_PLAN"); return v && atoi(v) != 0; }();
    auto tick = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[aai plan] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    // (bs: the caller's stream, or the device pool's build stream -- no stream is created or destroyed here; whatever was
    // enqueued is complete when this returns, also on failure)
    std::vector<int> spans;                                          // (declared before the guard: alive until the stream is synchronised)
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); } } guard{bs};
    if (p.kernel == AAI_KERNEL_AXIS) {
        aai::build_axis_tables(g, rq.mode, p.tabs, channels);
        if (band0 >= 0) aai::restrict_axis_tables_to_band(g, p.tabs, band0, band1, p.srcRow0, p.srcRow1, axis_band_margin(rq));
        if (p.tabs.wide) p.kernel = AAI_KERNEL_AXIS_WIDE;
        auto upload = [&](const void *h, size_t bytes, void **d) -> hipError_t {
            if (!bytes) { *d = nullptr; return hipSuccess; }
            hipError_t e = hipMalloc(d, bytes);
            if (e != hipSuccess) return e;
            return hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, bs);      // (p.tabs outlives the copy: synchronised below)
        };
        hipError_t e = upload(p.tabs.lane.data(), p.tabs.lane.size() * sizeof(aai::AxisEntry), (void **)&p.dLane);
        if (e == hipSuccess) e = upload(p.tabs.row.data(), p.tabs.row.size() * sizeof(aai::AxisEntry), (void **)&p.dRow);
        if (e == hipSuccess) e = upload(p.tabs.strips.data(), p.tabs.strips.size() * sizeof(aai::AxisStrip), (void **)&p.dStrips);
        if (e == hipSuccess) e = hipStreamSynchronize(bs);
        if (e != hipSuccess) return hip_fail(e, "uploading axis tables");
        stage("axis tables");
        if (p.kernel == AAI_KERNEL_AXIS) tune_axis_plan(p, channels, band0, bs);
        stage("launch-shape measurement");
    }
    if (p.kernel == AAI_KERNEL_ROTATED || p.kernel == AAI_KERNEL_FAST || p.kernel == AAI_KERNEL_SAMPLE) {
        // the corners of a rotated canvas hold nothing: which tiles of each tile row can (the whole image's table; bands index it by row)
        aai::rotated_live_spans(aai::make_rot_launch(g, rq.mode, rq.policy), p.kernel == AAI_KERNEL_SAMPLE, spans);
        if (!spans.empty()) {
            // (no synchronisation of its own: `spans` lives until the stream has been synchronised at the end of the build)
            hipError_t e = hipMalloc((void **)&p.dLive, spans.size() * sizeof(int));
            if (e == hipSuccess) e = hipMemcpyAsync(p.dLive, spans.data(), spans.size() * sizeof(int), hipMemcpyHostToDevice, bs);
            if (e != hipSuccess) return hip_fail(e, "uploading the live tile spans");
        }
        stage("live tile spans");
    }
    const bool axisKernel = p.kernel == AAI_KERNEL_AXIS || p.kernel == AAI_KERNEL_AXIS_WIDE;
    // K1's separable model against the reference's classifier (aai_axis_verify.hpp).  Both policies: they differ in the
    // corner-triangle rule of a slanted left/right edge only, which does not exist at multiples of 90 degrees.
    const bool verifyAxis = axisKernel && axis_band_margin(rq) != 0;
    if (p.kernel == AAI_KERNEL_ROTATED || p.kernel == AAI_KERNEL_FAST || verifyAxis) {
        const aai::RotLaunch r = aai::make_rot_launch(g, rq.mode, rq.policy);
        // (AAI_MAX_LISTED_PIXELS: test hook, lowers the threshold so that small geometries exercise the dense form)
        static const unsigned maxListed = [] { const char *v = getenv("AAI_MAX_LISTED_PIXELS"); return v ? (unsigned)strtoul(v, nullptr, 10) : kMaxListedPixels; }();      // documented in include/aai.h
        static const bool classVerify = [] { const char *v = aai::experiment_env("AAI_AXIS_CLASS_VERIFY"); return !(v && atoi(v) == 0); }();
        std::vector<std::pair<int, int>> hostPixels;
        bool hostDense = false;
        if (verifyAxis && classVerify && aai::axis_verify_by_class(r, hostPixels, hostDense, maxListed)) {
            std::vector<uint2> hostList(hostPixels.size());
            for (size_t i = 0; i < hostPixels.size(); ++i) hostList[i] = make_uint2((unsigned)hostPixels[i].first, (unsigned)hostPixels[i].second);
            // exact arithmetic: one representative per (column class, row class) checked on the host
            p.dense = hostDense;
            p.flaggedPixels = (unsigned)hostList.size();
            if (!hostList.empty()) {
                hipError_t e = hipMalloc(&p.dList, hostList.size() * sizeof(uint2));
                if (e == hipSuccess) e = hipMemcpyAsync(p.dList, hostList.data(), hostList.size() * sizeof(uint2), hipMemcpyHostToDevice, bs);
                if (e == hipSuccess) e = hipStreamSynchronize(bs);
                if (e != hipSuccess) return hip_fail(e, "axis model scan");
            }
            stage("axis model check (host)");
            return AAI_OK;
        }
        // one-off scans of this geometry (rotated: aai_knife_scan_kernel, and the scan of the fp32 formulation that serves
        // it; axis-aligned: aai_axis_verify_kernel); keeps the list of flagged pixels only if there are any
        const size_t waves = aai::rotated_flag_words(r);
        unsigned count = 0;
        hipError_t e = hipSuccess;
        if (waves) {
            // ONE allocation: lane masks | tile flags (one bit per 16 x 16 tile) | the scans' counter / the list's cursor
            const unsigned tilesX = (unsigned)((g.dW + 15) / 16), tilesY = (unsigned)((g.dH + 15) / 16);
            const int tileFlagWords = (int)aai::tile_flag_row_words(tilesX);
            const size_t maskBytes = waves * sizeof(unsigned long long);
            const size_t tileBytes = ((size_t)tilesY * (size_t)tileFlagWords * sizeof(unsigned) + 15) & ~(size_t)15;
            char *block = nullptr;
            e = hipMalloc((void **)&block, maskBytes + tileBytes + 16);
            unsigned long long *dMasks = reinterpret_cast<unsigned long long *>(block);
            unsigned *dTiles = reinterpret_cast<unsigned *>(block + maskBytes), *dCount = reinterpret_cast<unsigned *>(block + maskBytes + tileBytes);
            // (the scans write every lane-mask word they own; tile flags and the counter start at zero)
            if (e == hipSuccess) e = hipMemsetAsync(dTiles, 0, tileBytes + 16, bs);
            if (e == hipSuccess) e = verifyAxis ? aai::launch_axis_verify(r, dMasks, dCount, bs) : aai::launch_knife_scan(r, dMasks, dCount, bs);
            if (e == hipSuccess && r.quad) e = form == aai::ROT_FORM_CELL ? aai::launch_cell_scan(r, dMasks, dCount, bs) : aai::launch_quad_scan(r, dMasks, dCount, bs);
            if (e == hipSuccess && r.wide && !verifyAxis) e = aai::launch_wide_scan(r, dMasks, dCount, bs);
            if (e == hipSuccess) e = hipMemcpyAsync(&count, dCount, sizeof(unsigned), hipMemcpyDeviceToHost, bs);
            if (e == hipSuccess) e = hipStreamSynchronize(bs);
            stage("scans");
            if (e == hipSuccess && count > maxListed) { p.dense = true; count = 0; }
            if (e == hipSuccess && count) {
                e = hipMalloc(&p.dList, (size_t)count * 2 * sizeof(unsigned));
                if (e == hipSuccess) e = hipMemsetAsync(dCount, 0, sizeof(unsigned), bs);
                if (e == hipSuccess) e = aai::launch_flag_list(dMasks, waves, tilesX, p.dList, dCount, count, bs);
                if (e == hipSuccess && (r.quad || r.wide)) {
                    // keep the masks: the fp32 kernel skips the flagged pixels (the fix-up pass runs beside it) and asks for the
                    // per-pixel masks only in tiles that hold a flagged pixel
                    p.dScan = block; p.dMasks = dMasks; p.dTileFlags = dTiles; p.tileFlagWords = tileFlagWords;
                    block = nullptr;
                    e = aai::launch_tile_flags(p.dMasks, tilesX, tilesY, p.dTileFlags, bs);
                }
                if (e == hipSuccess) e = hipStreamSynchronize(bs);
            }
            if (block) (void)hipFree(block);
            stage("flag list + tile flags");
        }
        if (e != hipSuccess) return hip_fail(e, verifyAxis ? "axis model scan" : "knife-edge scan");
        p.flaggedPixels = count;
    }
    return AAI_OK;
}

int rot_form(const aai_request &rq, const aai::Geometry &g, int channels, int srcType, int64_t srcStride)
{
    if (pick_kernel(rq, g) != AAI_KERNEL_ROTATED) return aai::ROT_FORM_QUAD;
    aai::RotLaunch r = aai::make_rot_launch(g, rq.mode, rq.policy);
    r.chan = channels;
    return aai::cell_can_serve(r, srcType, aai::ImageView{srcStride, 0}) ? aai::ROT_FORM_CELL : aai::ROT_FORM_QUAD;
}

int acquire_plan(const aai_request &rq, int band0, int band1, int channels, int form, PlanRef *out, bool onCallerStream, hipStream_t stream)
{
    int dev = -1;
    AAI_HIP(hipGetDevice(&dev));
    aai::Geometry g;
    std::string msg;
    int rc = aai::make_geometry(rq, g, msg);
    if (rc != AAI_OK) return fail(rc, msg);
    const int kernel = pick_kernel(rq, g);
    if (kernel != AAI_KERNEL_ROTATED) form = aai::ROT_FORM_QUAD;
    if (kernel != AAI_KERNEL_AXIS && (band0 >= 0 || channels != 1)) {
        // Only K1's tables depend on the row band.  The per-pixel kernels take the band as launch parameters and their
        // one-off scans cover the whole image, so every band and every channel count of a rotated request shares one plan.
        band0 = band1 = -1; channels = 1;
    }
    PlanRef p;
    {
        std::lock_guard<std::mutex> lock(g_planMutex);
        std::list<PlanRef> &plans = plan_cache();
        for (auto it = plans.begin(); it != plans.end(); ++it) {
            const Plan &q = **it;
            if (q.device == dev && q.band0 == band0 && q.band1 == band1 && q.channels == channels && q.form == form && same_request(q.key, rq)) {
                plans.splice(plans.begin(), plans, it);
                p = plans.front();
                break;
            }
        }
        if (!p) {
            p = std::make_shared<Plan>();
            p->key = rq; p->band0 = band0; p->band1 = band1; p->channels = channels; p->form = form; p->device = dev; p->g = g; p->kernel = kernel;
            p->srcRow0 = 0; p->srcRow1 = g.H;
            plans.push_front(p);
            // the cache is sized per device: the least recently used plans of THIS device go first (a plan somebody is still
            // launching from lives on until its last reference is dropped)
            size_t mine = 0;
            for (const PlanRef &q : plans) mine += q->device == dev ? 1 : 0;
            for (auto it = plans.end(); mine > kMaxPlans && it != plans.begin();) {
                --it;
                if ((*it)->device == dev && *it != p) { it = plans.erase(it); --mine; }
            }
        }
    }
    {
        std::lock_guard<std::mutex> lock(p->build);
        if (!p->built) {
            const auto t0 = std::chrono::steady_clock::now();
            // where the build's kernels and copies go: the caller's stream when there is one to use (the resampling entry points;
            // not while it is being captured into a graph: the build synchronises), else the device pool's build stream
            hipStream_t bs = stream;
            bool own = onCallerStream;
            if (own) {
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
                if (cap != hipStreamCaptureStatusNone) own = false;
            }
            if (!own) {
                const hipError_t e = pool_build_stream(device_pool(dev), &bs);
                if (e != hipSuccess) { p->buildRc = hip_fail(e, "creating the build stream"); p->buildError = g_lastError; }
            }
            if (p->buildRc == AAI_OK) p->buildRc = build_plan(*p, bs);
            p->buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            p->buildError = p->buildRc == AAI_OK ? std::string() : g_lastError;
            p->built = true;
        }
    }
    if (p->buildRc != AAI_OK) {
        std::lock_guard<std::mutex> lock(g_planMutex);
        plan_cache().remove(p);               // a later call tries again
        return fail(p->buildRc, p->buildError);
    }
    *out = p;
    return AAI_OK;
}

std::string plan_description(const aai_request &rq, int channels)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return std::string();
    std::lock_guard<std::mutex> lock(g_planMutex);
    for (const PlanRef &q : plan_cache()) {
        const Plan &p = *q;
        const bool rotated = p.kernel != AAI_KERNEL_AXIS && p.kernel != AAI_KERNEL_AXIS_WIDE;
        if (p.device == dev && p.band0 < 0 && (rotated || p.channels == channels) && p.built && same_request(p.key, rq)) {
            char buf[256];
            snprintf(buf, sizeof buf, "kernel=%d rows=%d nt=%d swap=%d tune=%s flagged=%u dense=%d form=%s build_ms=%.3f", p.kernel, p.tuneRows, p.tuneNt,
                     p.tuneSwap, p.tuneSource == 1 ? "measured" : (p.tuneSource == 2 ? "cached" : "default"), p.flaggedPixels, p.dense ? 1 : 0,
                     p.kernel == AAI_KERNEL_ROTATED ? (p.form == aai::ROT_FORM_CELL ? "cell" : "quad") : "-", p.buildMs);
            return std::string(buf);
        }
    }
    return std::string();
}

constexpr int kMaxGridZ = 65535;

// element offset into a typed source buffer
static const void *src_at(const void *base, int srcType, int64_t elements)
{
    const int64_t esz = srcType == aai::SRC_U8 ? 1 : srcType == aai::SRC_U16 ? 2 : 4;
    return static_cast<const char *>(base) + elements * esz;
}

int enqueue(const aai_request &rq, int batch, const void *dSrc, int srcType, int64_t srcStride, int64_t srcImageStride,
            float *dDst, int64_t dstStride, int64_t dstImageStride, hipStream_t stream, int band0, int band1, int channels)
{
    PlanRef p;
    int rc;
    {
        aai::Geometry g0;
        std::string msg;
        rc = aai::make_geometry(rq, g0, msg);
        if (rc != AAI_OK) return fail(rc, msg);
        rc = acquire_plan(rq, band0, band1, channels, rot_form(rq, g0, channels, srcType, srcStride), &p, /*onCallerStream*/ true, stream);
    }
    if (rc != AAI_OK) return rc;
    // launches only enqueue; the plan's side stream and fork / join events are shared by its callers, hence the plan's lock
    std::lock_guard<std::mutex> lock(p->launch);
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
        r.dyBase = band0; r.dyEnd = band1; r.srcRow0 = srcRow0; r.srcRow1 = srcRow1;
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
        flags.masks = p->dMasks; flags.tileFlags = p->dTileFlags; flags.tileFlagWords = p->tileFlagWords; flags.live = p->dLive; flags.form = p->form;
        // A production kernel that skips the plan's listed pixels has the fix-up pass BESIDE it on a side stream of the device's
        // pool (fork / join events; the pool's lock is held while the launch is enqueued).  The pool's side streams are created by
        // the second such launch of the process: until then -- for the one call the reference's user makes -- the pass runs behind
        // the production kernel on the caller's stream, which then computes the listed pixels too and has them overwritten.
        DevicePool &pool = device_pool(p->device);
        std::unique_lock<std::mutex> poolLock(pool.m, std::defer_lock);
        if (p->flaggedPixels && p->dMasks && !p->dense) {
            poolLock.lock();
            if (!pool.sideReady && !pool.sideFailed && ++pool.besideLaunches >= 2) pool_create_sides(pool);
            if (pool.sideReady) {
                const unsigned slot = pool.nextSide++ % DevicePool::kSideSlots;
                flags.side = pool.side[slot]; flags.fork = pool.fork[slot]; flags.join = pool.join[slot];
            } else poolLock.unlock();
        }
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


}  // namespace engine
}  // namespace aai
