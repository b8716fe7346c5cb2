// aai_engine.hpp -- host-side engine behind the C ABI (aai_engine.cpp): error state, request checks, plan cache, dispatch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "aai_kernels.hpp"

namespace aai {
namespace engine {

// per-thread texts behind aai_last_error() / aai_last_kernel()
extern thread_local std::string g_lastError;
extern thread_local std::string g_lastKernel;
int fail(int code, const std::string &msg);                 // records msg, returns code
int hip_fail(hipError_t e, const char *what);               // AAI_ERR_NO_DEVICE for a missing / unusable device, else AAI_ERR_HIP

#define AAI_HIP(call)                                                     \
    do {                                                                  \
        hipError_t e__ = (call);                                          \
        if (e__ != hipSuccess) return ::aai::engine::hip_fail(e__, #call); \
    } while (0)

struct Plan {
    aai_request key{};
    int band0 = -1, band1 = -1;      // dst row band this plan serves (-1: the whole image)
    int channels = 1;                // interleaved channels the K1 tables were built for
    int form = 0;                    // rotated area / fast requests: the fp32 formulation whose scan flagged the pixels (aai::RotForm)
    int srcRow0 = 0, srcRow1 = 0;     // source rows the band reads; the source pointer addresses row srcRow0
    int device = -1;
    aai::Geometry g;
    int kernel = 0;
    // K1
    aai::AxisTables tabs;
    aai::AxisEntry *dLane = nullptr, *dRow = nullptr;
    aai::AxisStrip *dStrips = nullptr;
    int tuneRows = 0, tuneNt = 0, tuneSwap = 0;      // K1 launch shape for this device (0 rows = built-in default)
    int tuneSource = 0;                               // 0 = built-in default, 1 = measured for this plan, 2 = taken from the per-class cache
    // K2/K3: the dst pixels flagged by the one-off scans (knife edges of the reference's classifier; decisions the fp32
    // kernels leave to double precision) as a list of (dx, dy) the fix-up pass runs over; `dense` when there are
    // so many that the whole image takes that pass instead
    void *dList = nullptr;
    unsigned flaggedPixels = 0;
    bool dense = false;
    // fp32 rotated kernels: the lane masks of the flagged pixels (they skip them; the fix-up pass runs beside them on a side stream of
    // the device's pool)
    void *dScan = nullptr;           // ONE allocation: lane masks | tile flags | the scans' counter
    unsigned long long *dMasks = nullptr;      // (into dScan)
    unsigned *dTileFlags = nullptr;  // (into dScan) one bit per 16 x 16 tile with a flagged pixel (aai::launch_tile_flags), tileFlagWords words per tile row
    int tileFlagWords = 0;
    int *dLive = nullptr;            // per-pixel kernels on a rotated canvas: live tile span per tile row (aai::rotated_live_spans), or none
    double buildMs = 0.0;            // wall clock of build_plan (tables, scans, launch-shape measurement)
    // Built once, by whoever gets here first, under `build` -- NOT under the cache's lock: other requests, other devices
    // and other threads are not held up by this plan's scans or launch-shape measurement.  `launch` serialises the launches of a
    // plan between caller threads.
    std::mutex build, launch;
    bool built = false;
    int buildRc = AAI_OK;
    std::string buildError;
    ~Plan()
    {
        if (dScan) (void)hipFree(dScan);
        if (dLive) (void)hipFree(dLive);
        if (dList) (void)hipFree(dList);
        if (dLane) (void)hipFree(dLane);
        if (dRow) (void)hipFree(dRow);
        if (dStrips) (void)hipFree(dStrips);
    }
};
typedef std::shared_ptr<Plan> PlanRef;

// Streams and events the engine needs beside the callers' own, ONE set per device and process, created on first need: creating a
// stream costs 2.7 ... 5 ms of host time on this platform (a hardware queue each; profiles/r04_plan_time.txt), which at one private
// stream and four side streams per PLAN was 16 of the 19 ms a first rotated call took.
//   build   aai_prepare's stream (the resampling entry points build a missing plan on the CALLER's stream instead)
//   side[]  the fix-up pass beside a production kernel that skips the plan's listed pixels: dealt round-robin per launch, each with
//           its fork / join events, so that the passes of callers on different streams run beside each other.  Created by the SECOND
//           launch that wants them: a process that makes one call (the reference's user, Source.cpp:1565) runs its pass behind the
//           production kernel (~10 us) and never pays the ~11 ms.
// `m` guards creation and the slot rotation and is held while a launch with a pass beside it is being enqueued (fork ... join).
struct DevicePool {
    static constexpr int kSideSlots = 4;
    std::mutex m;
    hipStream_t build = nullptr;
    hipStream_t side[kSideSlots] = {};
    hipEvent_t fork[kSideSlots] = {}, join[kSideSlots] = {};
    bool sideReady = false, sideFailed = false;
    unsigned besideLaunches = 0, nextSide = 0;
};
DevicePool &device_pool(int device);      // (heap, never destroyed; aai_shutdown releases the streams while the runtime is alive)

// The cache (most recently used first) lives on the heap and is never destroyed: a static destructor would call into HIP
// after the runtime's own teardown.  aai_shutdown() empties it while the runtime is alive.
extern std::mutex g_planMutex;          // guards the cache's structure only; never held across device work
std::list<PlanRef> &plan_cache();
void drop_plans();

bool same_request(const aai_request &a, const aai_request &b);
int check_request(const aai_request *rq);
int axis_band_margin(const aai_request &rq);                 // extra source rows either side of a K1 row band (fix-up pass)
int pick_kernel(const aai_request &rq, const Geometry &g);
int resolved_kernel(const aai_request &rq, const Geometry &g);      // pick_kernel, with AXIS -> AXIS_WIDE where the tables say so
void fill_layout(const Geometry &g, int kernel, aai_layout *out);
int require_device();

// Finds the plan for (request, current device) or inserts a fresh one, then builds it if nobody has (blocking: table
// uploads, the one-off scans, K1's launch-shape measurement -- on `stream` when the caller has one to give (onCallerStream; a
// stream that is being captured into a graph is not used), else on the device pool's build stream).
// form: aai::RotForm of a rotated request's launch (rot_form below); ignored by the other kernels
int acquire_plan(const aai_request &rq, int band0, int band1, int channels, int form, PlanRef *out, bool onCallerStream = false, hipStream_t stream = nullptr);
// "kernel=K rows=R nt=N swap=S tune=T flagged=F dense=D form=M build_ms=B" of the cached whole-image plan ("" when there is none)
std::string plan_description(const aai_request &rq, int channels);
// which fp32 formulation serves a launch of this request: the cell formulation takes plain images below 4 GiB in area mode
int rot_form(const aai_request &rq, const Geometry &g, int channels, int srcType, int64_t srcStride);

// Enqueues one batched launch (plus the fix-up pass where the plan has one) on `stream`.  Strides in elements.
int enqueue(const aai_request &rq, int batch, const void *dSrc, int srcType, int64_t srcStride, int64_t srcImageStride,
            float *dDst, int64_t dstStride, int64_t dstImageStride, hipStream_t stream, int band0 = -1, int band1 = -1,
            int channels = 1);

}  // namespace engine
}  // namespace aai
