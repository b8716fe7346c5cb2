/* aai.h -- C ABI of the MI355X-native area-average interpolation engine (libaai_hip.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (Ishikawa-lab/Area_average_interpolation, Source.cpp) has no FFI of its own: its boundary is the
 * public section of `class AreaAverageInterpolation` (Source.cpp:52-54),
 *
 *     pair<bool,string> areaAverageInterpolation    (IMG src, IMG &dst, dP srcResolution, dP dstResolution,
 *     pair<bool,string> fastAreaAverageInterpolation              dP srcIsocenter, dP &dstIsocenter, double rotationAngle)
 *
 * (Source.cpp:55-57 and 584-586, argument semantics Source.cpp:78-87), called from Source.cpp:1565 / 1569.
 * Every entry point below cites the reference interface it replaces.  include/AreaAverageInterpolation.hpp
 * re-creates the class on top of this ABI so that `main()`-style callers compile unchanged;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions (all taken from the reference):
 *   - images are row-major [y][x] (Source.cpp:31, 150); strides are in ELEMENTS, not bytes;
 *   - `first` = x, `second` = y for every pair (Source.cpp:43-46);
 *   - resolutions in pixel/mm or dpi, only their ratio matters, x and y must agree (Source.cpp:112-117);
 *   - isocenter in pixel-centre coordinates of the input image (Source.cpp:82);
 *   - rotation in degrees, clockwise positive, any real value (Source.cpp:83, 141-142);
 *   - the output size depends on the rotation (Source.cpp:179-180): query it with aai_query().
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * All functions return AAI_OK (0) or an AAI_ERR_* code; aai_last_error() gives the message.
 */
#ifndef AAI_H
#define AAI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AAI_VERSION_MAJOR 0
#define AAI_VERSION_MINOR 1

/* ---- status codes ------------------------------------------------------------------------- */
enum {
    AAI_OK = 0,
    AAI_ERR_RESOLUTION_MISMATCH = 1,  /* "Assumed X & Y resolution are same."                       Source.cpp:112-117 */
    AAI_ERR_RESOLUTION_NONPOSITIVE = 2, /* "0 or negative resolution is not acceptable."            Source.cpp:118-122 */
    AAI_ERR_NO_ROWS = 3,              /* "There is no data in src array."                           Source.cpp:123-127 */
    AAI_ERR_NO_COLUMNS = 4,           /* "There is no data in the second dimension of src array."   Source.cpp:128-132 */
    AAI_ERR_NONFINITE = 5,            /* NaN/Inf argument: the reference does not check (UB at Source.cpp:139); we reject */
    AAI_ERR_BAD_ARGUMENT = 6,         /* null pointer, unknown mode/policy, stride < width, batch < 0 ... */
    AAI_ERR_TOO_LARGE = 7,            /* output or virtual source would overflow 31-bit indexing */
    AAI_ERR_NO_DEVICE = 8,            /* no HIP device / runtime unavailable: the product has NO CPU fallback */
    AAI_ERR_HIP = 9,                  /* a HIP runtime call failed; message holds hipGetErrorString */
    AAI_ERR_EMPTY_OUTPUT = 10         /* dst width or height rounds to 0 (Source.cpp:179-180): the reference crashes there (its edge-line
                                         tables index dstSize - 1, Source.cpp:243-305); we report it */
};

/* ---- what to compute ---------------------------------------------------------------------- */
enum {
    AAI_MODE_AREA = 1,      /* areaAverageInterpolation,     Source.cpp:55   (interpolationMode 1, Source.cpp:1563) */
    AAI_MODE_FAST = 2,      /* fastAreaAverageInterpolation, Source.cpp:584  (interpolationMode 2, Source.cpp:1567) */
    AAI_MODE_BILINEAR = 3,  /* comparison path named in README.md:8; absent from the reference -> build-defined */
    AAI_MODE_BICUBIC = 4    /* comparison path named in README.md:8; absent from the reference -> build-defined */
};
enum {
    AAI_POLICY_REFERENCE = 0, /* overlap areas exactly as getArea() returns them, including the corner-triangle
                                 leg choice of Source.cpp:1055-1062 (the graded, reference-compatible answer) */
    AAI_POLICY_EXACT = 1,     /* geometrically exact overlap areas (differs from the reference for rotations
                                 that are not multiples of 90 degrees) */
    /* OR into either policy: general rotations compute in double precision throughout (about 3 x the time).  The
     * default kernels evaluate overlap areas in fp32 relative to the nearest source pixel: each area is off by at most
     * ~1e-7 ABSOLUTE (rotations near the axes and up-sampling take their edge parameters from double precision), a dst
     * value by ~5e-8 x the spread of the source values under its footprint -- inside the 1e-5 relative bar wherever a
     * dst value is not hundreds of times smaller than those neighbours (fuzz of round 2: worst 1-3e-6 of
     * max(|value|, 1e-3) per 3,000 random cases).  Images
     * whose neighbouring values span many orders of magnitude can ask for double precision here. */
    AAI_POLICY_DOUBLE_PRECISION = 0x100,
    /* OR into either policy -- a hint: general rotations in area mode take the "cell" formulation (every (dst, src) pair evaluated
     * once and shared between the dst pixels it feeds) also for outputs below ~720 x 720 pixels, which by default stay on the
     * one-lane-per-pixel kernel because the cell kernel's long-lived waves cannot fill the chip there.  Same results to ~1e-6
     * (both are pinned to the reference); tests use it to run every fixture through both formulations. */
    AAI_POLICY_PREFER_CELL = 0x200,
    /* OR into either policy -- a DIAGNOSTIC, per request (there is no process-wide switch): the double-precision pass over the
     * dst pixels the plan lists (aai_plan_info: flagged=) is not launched, so those pixels keep the bytes the caller's buffer
     * held.  The output of such a request is NOT the reference's answer in the listed pixels.  For tests (which pixels do the
     * fp32 kernels leave alone?) and timing (what does the pass cost beside the production kernel?). */
    AAI_POLICY_DIAG_NO_FIXUP = 0x400,
    AAI_POLICY_RULE_MASK = 0xff   /* the weight policy proper: AAI_POLICY_REFERENCE or AAI_POLICY_EXACT */
};

/* Source element types of the typed entry points (real images are rarely double).
 * Pixel values are used as they are (0..255, 0..65535); the output is always fp32. */
enum { AAI_DTYPE_F32 = 0, AAI_DTYPE_U8 = 1, AAI_DTYPE_U16 = 2 };

/* One resampling request = the by-value arguments of Source.cpp:55-57 plus the engine's mode switches. */
typedef struct aai_request {
    int32_t mode;            /* AAI_MODE_*    */
    int32_t policy;          /* AAI_POLICY_*  (ignored by FAST / BILINEAR / BICUBIC) */
    int32_t src_width;       /* src.front().size(), Source.cpp:150 */
    int32_t src_height;      /* src.size(),         Source.cpp:150 */
    double src_res_x, src_res_y;   /* srcResolution.first/.second  */
    double dst_res_x, dst_res_y;   /* dstResolution.first/.second  */
    double src_iso_x, src_iso_y;   /* srcIsocenter.first/.second   */
    double rotation_deg;           /* rotationAngle                */
} aai_request;

/* What the callee decides: `dst` size (Source.cpp:179-180, 411-414) and `dstIsocenter` (Source.cpp:181-186),
 * plus the derived quantities callers of the reference can only infer. */
typedef struct aai_layout {
    int32_t dst_width, dst_height;
    double dst_iso_x, dst_iso_y;     /* integer-valued, like the reference writes them */
    int32_t scale;                   /* integer pre-expansion factor, Source.cpp:139 */
    int32_t quadrant;                /* 0..3 = 0/90/180/270 degree pre-rotation, Source.cpp:140-146 */
    double reduced_angle_deg;        /* rotation left after the pre-rotation, in [0,90) */
    double side;                     /* dst pixel side in virtual-source pixels (dstSideLength, Source.cpp:178) */
    int32_t kernel;                  /* which device path serves this request: AAI_KERNEL_* */
    int32_t reserved;
} aai_layout;

enum {
    AAI_KERNEL_AXIS = 1,      /* separable streaming kernel: reduced angle == 0 (K1) */
    AAI_KERNEL_ROTATED = 2,   /* general clip kernel (K2) */
    AAI_KERNEL_FAST = 3,      /* centre-inclusion kernel (K3) */
    AAI_KERNEL_SAMPLE = 4,    /* bilinear / bicubic point samplers (K4/K5) */
    AAI_KERNEL_AXIS_WIDE = 5  /* axis-aligned, footprint wider than one wave strip: per-pixel fallback kernel */
};

/* ---- host-only entry points (no GPU needed) ---------------------------------------------------------- */

/* Validate a request and compute the output layout.  Replaces the argument checks and the affine
 * set-up of Source.cpp:112-200 (duplicated at 638-726).  On failure `out` is left untouched, like the
 * reference leaves dst/dstIsocenter untouched. */
int aai_query(const aai_request *req, aai_layout *out);

/* Message of the most recent failure on the calling thread ("" if none).  For the four argument errors
 * the text is byte-identical to the reference's (Source.cpp:115, 120, 125, 130). */
const char *aai_last_error(void);
const char *aai_error_string(int code);
int aai_version(void);   /* major*1000 + minor */

/* ---- device management ------------------------------------------------------------------------------ */
int aai_device_count(int *count);
int aai_set_device(int ordinal);          /* one process per GPU: call once with LOCAL_RANK */
int aai_device_synchronize(void);

/* ---- resampling: host buffers ------------------------------------------------------------------------
 * Replaces the whole call at Source.cpp:1565 / 1569 for callers that hold the image in host memory:
 * upload, run the hot path on the GPU, download.  `dst` must hold layout.dst_height rows of
 * `dst_stride` elements (query the size first); `layout` may be NULL. */
int aai_resample_f32(const aai_request *req, const float *src, int64_t src_stride,
                     float *dst, int64_t dst_stride, aai_layout *layout);
/* Same with the reference's element type (IMG = vector<vector<double>>, Source.cpp:31).  The device
 * computes on fp32 pixels (geometry in fp64, weights in fp32 relative to the nearest source pixel or fp64, see
 * AAI_POLICY_DOUBLE_PRECISION); results agree with the reference to 1e-5 relative. */
int aai_resample_f64(const aai_request *req, const double *src, int64_t src_stride,
                     double *dst, int64_t dst_stride, aai_layout *layout);

/* ---- resampling: device-resident buffers (the measured hot path) ---------------------------------------
 * Replaces the loops of Source.cpp:413-579 / 868-907 (and the modSrc / dstPos / edge-line tables of
 * Source.cpp:150-305, which are never materialised).  `stream` is a hipStream_t passed as void* (NULL =
 * default stream).  The call only enqueues work and returns -- except the FIRST call for a given (request, device):
 * that one builds the plan (K1: weight tables uploaded with blocking copies; rotated requests: one-off scans of the
 * geometry for pixels that need the double-precision pass, read back with a blocking copy), so it synchronises with the
 * device and must not run inside a stream capture.  aai_prepare takes that cost up front; plans are cached per
 * process (32 most recently used), shared by batches, row bands (rotated requests) and streams. */
int aai_prepare(const aai_request *req, int32_t channels /* 1 for plain images; 2..4: interleaved */);
/* What building a plan does on the device (inside aai_prepare, or inside the first resampling call of a request):
 *   - K1 (rotation by a multiple of 90 degrees): uploads the weight tables; checks the separable model against the
 *     reference's classifier -- on the host where the geometry's arithmetic is exact (integer and simple ratios), else by
 *     one scan kernel over the output; and, for the FIRST large plain-fp32 geometry of a class (same device, same source
 *     rows per output row, same width class) measures which launch shape streams fastest on this device: transient
 *     scratch of 2..8 source-sized images (at most 1.25 GiB; larger sources skip the measurement) plus their outputs,
 *     ~40 launches on a private stream, ~10 ms.  Later plans of the class reuse the result.  AAI_AXIS_AUTOTUNE=0 in the
 *     environment disables the measurement (built-in shape).
 *   - rotated area / fast requests: one or two scan kernels over the output (pixels left to the double-precision pass).
 * All of it runs on a private stream and blocks only the calling thread: plans of other requests, devices and threads are
 * built and launched from concurrently.
 *
 * Environment variables the shipping library reads (all of them; launch-heuristic overrides exist only in the experiments build,
 * `make -C area_average_interpolation_amd/csrc exp`):
 *   AAI_AXIS_AUTOTUNE=0      no launch-shape measurement for K1 (built-in shape), see above
 *   AAI_MAX_LISTED_PIXELS=n  a plan whose scans list more than n pixels (default 16 M) hands the WHOLE image to the double-precision
 *                            pass instead of keeping the list (`dense` in aai_plan_info); results are the same either way, only the
 *                            time differs -- tests lower it to exercise that form on small images
 *   AAI_TRACE_PLAN=1         stage timings of every plan build on stderr
 *
 * aai_plan_info writes a one-line description of the cached whole-image plan of `req` on the current device into `text`
 * ("" when there is none yet): "kernel=K rows=R nt=N swap=S tune=measured|cached|default flagged=F dense=D form=cell|quad|-
 * build_ms=B" -- the kernel family (AAI_KERNEL_*), K1's launch shape and where it came from, the dst pixels the
 * double-precision pass owns, the fp32 formulation of a rotated area request, and what building the plan cost. */
int aai_plan_info(const aai_request *req, int32_t channels, char *text, int32_t capacity);
/* Drops every cached plan (device tables, flag lists, side streams) while the HIP runtime is alive.  Optional: the cache is
 * never torn down from a static destructor, so a process may also simply exit. */
int aai_shutdown(void);
int aai_resample_device_f32(const aai_request *req, const float *d_src, int64_t src_stride,
                            float *d_dst, int64_t dst_stride, void *stream);

/* A batch of `batch` independent images of identical geometry (BASELINE config 4): image b starts at
 * d_src + b*src_image_stride and is written to d_dst + b*dst_image_stride.  One launch covers the batch. */
int aai_resample_batch_device_f32(const aai_request *req, int32_t batch,
                                  const float *d_src, int64_t src_stride, int64_t src_image_stride,
                                  float *d_dst, int64_t dst_stride, int64_t dst_image_stride,
                                  void *stream);

/* The same batch spread over SEVERAL GPUs of this process (images are independent, there is no
 * exchange step, so no collective library is needed): shard i = counts[i] images resident on device devices[i] at
 * d_src[i] / d_dst[i], enqueued on streams[i] (a stream of THAT device, or NULL).  Every shard is launched before the
 * call returns; nothing is synchronised (the first call per device builds that device's plan, see aai_prepare).  The
 * caller's current device is restored.  One process, N GPUs: what a C++ user of the reference's class needs to use a
 * whole node; the one-process-per-GPU form over RCCL is area_average_interpolation_amd/distributed.py. */
int aai_resample_batch_multi_device_f32(const aai_request *req, int32_t n_shards, const int32_t *devices, const int32_t *counts,
                                        const float *const *d_src, int64_t src_stride, int64_t src_image_stride,
                                        float *const *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *const *streams);

/* ---- typed sources (8-bit / 16-bit unsigned, or fp32): SURVEY.md section 8(f) N3 ----------------------------
 * Same semantics as the f32 entries above with `src` holding elements of `src_dtype` (AAI_DTYPE_*); strides are
 * in source ELEMENTS.  The reference only accepts doubles (Source.cpp:31); callers holding 8/16-bit images would
 * otherwise widen them on the host and move 4-8x the bytes over PCIe and HBM. */
int aai_resample_batch_device(const aai_request *req, int32_t batch, const void *d_src, int32_t src_dtype,
                              int64_t src_stride, int64_t src_image_stride,
                              float *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *stream);
int aai_resample_host(const aai_request *req, const void *src, int32_t src_dtype, int64_t src_stride,
                      float *dst, int64_t dst_stride, aai_layout *layout);

/* ---- interleaved channels --------------------------------------------------------------
 * Images whose pixels hold `channels` (1..4) interleaved values, e.g. RGB scans: element (x, y, c) of image b is
 * src[b*src_image_stride + y*src_stride + x*channels + c], and the output has the same layout (fp32).  Strides are in
 * ELEMENTS (src_stride >= width*channels, dst_stride >= dst_width*channels).  Every channel gets the result
 * of the single-channel call on that channel alone; the rotated-lattice kernels compute the overlap areas once per
 * pixel pair for all channels, the axis-aligned kernel reads every source byte once.  The reference only knows
 * single-channel images (IMG, Source.cpp:31). */
int aai_resample_interleaved_device(const aai_request *req, int32_t batch, int32_t channels,
                                    const void *d_src, int32_t src_dtype, int64_t src_stride, int64_t src_image_stride,
                                    float *d_dst, int64_t dst_stride, int64_t dst_image_stride, void *stream);
int aai_resample_interleaved_host(const aai_request *req, int32_t channels, const void *src, int32_t src_dtype, int64_t src_stride,
                                  float *dst, int64_t dst_stride, aai_layout *layout);

/* ---- host batches, pipelined ("pinned-memory streaming pipeline") ----------------------------
 * `batch` images in HOST memory (image b at src + b*src_image_stride elements, written to dst + b*dst_image_stride):
 * the images go through three device slots, each with its own HIP stream (upload -> kernel -> download in stream
 * order), so the upload of image b+1 and the download of image b-1 overlap the kernel of image b and each other.  The
 * copies are truly asynchronous only from page-locked memory: allocate the buffers with aai_host_alloc (hipHostMalloc)
 * or register them yourself; pageable buffers work too, at the runtime's staged-copy rate.  Replaces a loop over the
 * call at Source.cpp:1565 / 1569 for callers that hold many images. */
int aai_host_alloc(void **ptr, uint64_t bytes);
int aai_host_free(void *ptr);
int aai_resample_batch_host(const aai_request *req, int32_t batch, const void *src, int32_t src_dtype,
                            int64_t src_stride, int64_t src_image_stride,
                            float *dst, int64_t dst_stride, int64_t dst_image_stride, aai_layout *layout);

/* ---- row bands of one image -------------------------------------------------------
 * dst rows [dst_row0, dst_row1) only: for sharding ONE image over several GPUs (each rank computes a band, no
 * collective: a band only reads its own source footprint) or for images larger than device memory.
 * aai_band_source_rows (host only) tells which source rows [src_row0, src_row1) the band reads; the device call
 * takes d_src_rows = address of source row src_row0 (a buffer holding just those rows is enough) and
 * d_dst_rows = address of output row dst_row0.  For rotated requests dst_row0 must be a multiple of 16.  The band
 * results are bit-identical to the same rows of the full-image call. */
int aai_band_source_rows(const aai_request *req, int32_t dst_row0, int32_t dst_row1, int32_t *src_row0, int32_t *src_row1);
int aai_resample_band_device_f32(const aai_request *req, int32_t dst_row0, int32_t dst_row1,
                                 const float *d_src_rows, int64_t src_stride, float *d_dst_rows, int64_t dst_stride, void *stream);

/* ---- synthetic input ------------------------------------------------------------
 * Fill a device image with the stateless-hash fp32 uniform [0,1) pattern used by every benchmark and
 * known-answer test: idx = y*width + x, seed as given (image b of a batch uses seed b+1). */
int aai_synth_device_f32(float *d_dst, int32_t width, int32_t height, int64_t stride,
                         uint64_t seed, void *stream);
/* rows [row0, row1) of that width x height pattern only; d_dst addresses row row0 (the source footprint of a row band) */
int aai_synth_rows_device_f32(float *d_dst, int32_t width, int32_t height, int32_t row0, int32_t row1, int64_t stride,
                              uint64_t seed, void *stream);

/* Name and launch geometry of the kernel that served the most recent device call on this thread
 * (for profiling / bench bookkeeping). */
const char *aai_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* AAI_H */
