// aai_capi.cpp -- the C ABI declared in include/aai.h: the extern "C" entry points and the host-buffer convenience
// paths, on top of the engine (aai_engine.cpp: plan cache, dispatch, error state).
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

#include "aai_engine.hpp"

using namespace aai::engine;

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
    aai::Geometry g;
    {
        std::string msg;
        rc = aai::make_geometry(*req, g, msg);
        if (rc != AAI_OK) return fail(rc, msg);
    }
    rc = require_device();
    if (rc != AAI_OK) return rc;
    PlanRef p;
    // (the plan of a packed fp32 image: what the device entries build on their first call)
    rc = acquire_plan(*req, -1, -1, channels, rot_form(*req, g, channels, aai::SRC_F32, (int64_t)g.W * channels), &p);
    if (rc == AAI_OK) g_lastError.clear();
    return rc;
}

int aai_plan_info(const aai_request *req, int32_t channels, char *text, int32_t capacity)
{
    int rc = check_request(req);
    if (rc != AAI_OK) return rc;
    if (!text || capacity <= 0) return fail(AAI_ERR_BAD_ARGUMENT, "Null text buffer.");
    rc = require_device();
    if (rc != AAI_OK) return rc;
    const std::string d = plan_description(*req, channels);
    snprintf(text, (size_t)capacity, "%s", d.c_str());
    g_lastError.clear();
    return AAI_OK;
}

int aai_shutdown(void)
{
    drop_plans();
    g_lastError.clear();
    return AAI_OK;
}

#if defined(AAI_EXPERIMENTS)
/* experiments build only (tools/tune_axis.py): the run-time form of AAI_AXIS_TUNE; not declared in include/aai.h */
void aai_debug_axis_tune(const char *spec) { aai::set_axis_tune(spec); }
#endif

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
