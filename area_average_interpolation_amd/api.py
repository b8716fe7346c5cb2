"""Host-side mirror of the reference's call surface on top of the C ABI (libaai_hip.so).

The reference boundary is ``class AreaAverageInterpolation`` with two methods of identical signature
(/root/reference/Source.cpp:55-57 and 584-586)::

    pair<bool,string> areaAverageInterpolation    (IMG src, IMG &dst, dP srcResolution, dP dstResolution,
    pair<bool,string> fastAreaAverageInterpolation              dP srcIsocenter, dP &dstIsocenter, double rotationAngle)

Python has no out-parameters, so the two methods below keep the names, the argument order and meaning and
the error behaviour, and return ``((ok, message), dst, dstIsocenter)``; on failure ``dst`` and
``dstIsocenter`` are ``None`` (the reference leaves them untouched).  The C++ drop-in with the exact
reference signature is include/AreaAverageInterpolation.hpp.

Everything here is plumbing around the C ABI; all arithmetic happens in the HIP kernels.
"""
import ctypes

import numpy as np

from . import _lib as L


class AaiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("aai error %d: %s" % (code, message))
        self.code, self.message = code, message


def _pair(v):
    if np.isscalar(v):
        return float(v), float(v)
    return float(v[0]), float(v[1])


def make_request(width, height, src_resolution, dst_resolution, src_isocenter, rotation_angle,
                 mode=L.MODE_AREA, policy=L.POLICY_REFERENCE):
    sr, dr, iso = _pair(src_resolution), _pair(dst_resolution), _pair(src_isocenter)
    return L.Request(int(mode), int(policy), int(width), int(height), sr[0], sr[1], dr[0], dr[1],
                     iso[0], iso[1], float(rotation_angle))


# Per-request hint / diagnostic bits the debug_* helpers below switch on for every request made through this module (the library
# itself has no process-wide switches: include/aai.h, AAI_POLICY_PREFER_CELL / AAI_POLICY_DIAG_NO_FIXUP)
_extra_policy = 0


def _ref(request):
    """ctypes reference to the request as the library should see it: with the module's extra policy bits, if any"""
    if not _extra_policy:
        return ctypes.byref(request)
    rq = L.Request.from_buffer_copy(request)
    rq.policy |= _extra_policy
    return ctypes.byref(rq)


def last_error():
    return L.load().aai_last_error().decode()


def query(request):
    """aai_query: validate + output layout.  Returns (code, message, Layout-or-None).  Needs no GPU."""
    lib = L.load()
    lay = L.Layout()
    rc = lib.aai_query(_ref(request), ctypes.byref(lay))
    if rc != L.OK:
        return rc, last_error(), None
    return rc, "", lay


def device_count():
    n = ctypes.c_int(0)
    L.load().aai_device_count(ctypes.byref(n))
    return n.value


def set_device(ordinal):
    rc = L.load().aai_set_device(int(ordinal))
    if rc != L.OK:
        raise AaiError(rc, last_error())


def synchronize():
    rc = L.load().aai_device_synchronize()
    if rc != L.OK:
        raise AaiError(rc, last_error())


def last_kernel():
    return L.load().aai_last_kernel().decode()


def resample_host(src, src_resolution, dst_resolution, src_isocenter, rotation_angle,
                  mode=L.MODE_AREA, policy=L.POLICY_REFERENCE):
    """Host-buffer path (aai_resample_f32 / aai_resample_f64 by dtype).

    Returns (code, message, dst ndarray or None, (dstIsoX, dstIsoY) or None, Layout or None)."""
    lib = L.load()
    a = np.asarray(src)
    if a.ndim == 2 and a.dtype in (np.uint8, np.uint16):
        # typed entry (aai_resample_host): 8/16-bit source, fp32 output
        a = np.ascontiguousarray(a)
        H, W = a.shape
        rq = make_request(W, H, src_resolution, dst_resolution, src_isocenter, rotation_angle, mode, policy)
        rc, msg, lay = query(rq)
        if rc != L.OK:
            return rc, msg, None, None, None
        dst = np.empty((lay.dst_height, lay.dst_width), dtype=np.float32)
        out_lay = L.Layout()
        rc = lib.aai_resample_host(_ref(rq), a.ctypes.data, L.DTYPE_U8 if a.dtype == np.uint8 else L.DTYPE_U16, W,
                                   dst.ctypes.data, max(lay.dst_width, 1), ctypes.byref(out_lay))
        if rc != L.OK:
            return rc, last_error(), None, None, None
        return rc, "", dst, (out_lay.dst_iso_x, out_lay.dst_iso_y), out_lay
    if a.ndim != 2:
        if a.size != 0:
            raise ValueError("src must be a 2-D image [H, W] (interleaved [H, W, C] images: resample_interleaved_host)")
        a = a.reshape(0, 0)        # the reference's two "no data" errors (Source.cpp:123-132)
    if a.dtype != np.float32:
        a = a.astype(np.float64, copy=False)
    a = np.ascontiguousarray(a)
    H, W = (a.shape[0], a.shape[1]) if a.ndim == 2 else (0, 0)
    rq = make_request(W, H, src_resolution, dst_resolution, src_isocenter, rotation_angle, mode, policy)
    rc, msg, lay = query(rq)
    if rc != L.OK:
        return rc, msg, None, None, None
    dst = np.empty((lay.dst_height, lay.dst_width), dtype=a.dtype)
    fn = lib.aai_resample_f32 if a.dtype == np.float32 else lib.aai_resample_f64
    out_lay = L.Layout()
    rc = fn(_ref(rq), a.ctypes.data, W, dst.ctypes.data, max(lay.dst_width, 1), ctypes.byref(out_lay))
    if rc != L.OK:
        return rc, last_error(), None, None, None
    return rc, "", dst, (out_lay.dst_iso_x, out_lay.dst_iso_y), out_lay


def precision_check(src, src_resolution, dst_resolution, src_isocenter, rotation_angle, mode=L.MODE_AREA, policy=L.POLICY_REFERENCE,
                    floor=None):
    """Does THIS image need AAI_POLICY_DOUBLE_PRECISION?  The default kernels take general rotations in fp32 (include/aai.h): a dst
    value is then off by ~5e-8 x the spread of the source values under its footprint, which only shows where a dst value is hundreds
    of times smaller than its neighbours.  That is a property of the data, not of the geometry, so no plan-time scan can flag it:
    this runs the request under both policies and returns (worst relative deviation, dst of the default policy, dst of the
    double-precision policy).  `floor`: absolute floor of the denominator (default 1e-6 x the largest |value|).  A deviation
    above ~1e-5 says: use `policy | POLICY_DOUBLE_PRECISION` for images like this one (about 3 x the time)."""
    rc, msg, fast, _, _ = resample_host(src, src_resolution, dst_resolution, src_isocenter, rotation_angle, mode=mode, policy=policy)
    if rc != L.OK:
        raise AaiError(rc, msg)
    rc, msg, exact, _, _ = resample_host(src, src_resolution, dst_resolution, src_isocenter, rotation_angle, mode=mode,
                                        policy=policy | L.POLICY_DOUBLE_PRECISION)
    if rc != L.OK:
        raise AaiError(rc, msg)
    if exact.size == 0:
        return 0.0, fast, exact
    a, b = fast.astype(np.float64), exact.astype(np.float64)
    fl = floor if floor is not None else 1e-6 * max(float(np.abs(b).max()), 1e-300)
    return float((np.abs(a - b) / np.maximum(np.abs(b), fl)).max()), fast, exact


_NP_DTYPES = {np.dtype(np.float32): L.DTYPE_F32, np.dtype(np.uint8): L.DTYPE_U8, np.dtype(np.uint16): L.DTYPE_U16}


class _PinnedBlock:
    """Owner of one hipHostMalloc allocation: freed when the last numpy view of it is gone."""

    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        rc = L.load().aai_host_alloc(ctypes.byref(self.ptr), nbytes)
        if rc != L.OK:
            raise AaiError(rc, last_error())

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr is not None and ptr.value:
            try:
                L.load().aai_host_free(ptr)
            except Exception:           # interpreter shutdown: the runtime may already be gone
                pass


class PinnedArray:
    """A numpy array in page-locked host memory (aai_host_alloc = hipHostMalloc), so that the pipelined host-batch
    entry copies asynchronously.  Use `.array`.  The allocation is owned by the array's base object: close() (or
    leaving the `with` block) only drops this object's reference, and the memory is released when the last numpy view
    of it -- `.array`, slices of it, an `out=` result of resample_batch_host -- has been garbage-collected, so a view
    held elsewhere never dangles."""

    def __init__(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        block = _PinnedBlock(max(n, 1))
        buf = (ctypes.c_char * max(n, 1)).from_address(block.ptr.value)
        buf._aai_owner = block            # ndarray.base -> buf -> block keeps the allocation alive
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        self.array = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def resample_batch_host(srcs, src_resolution, dst_resolution, src_isocenter, rotation_angle,
                        mode=L.MODE_AREA, policy=L.POLICY_REFERENCE, out=None):
    """Pipelined host-batch path (aai_resample_batch_host): srcs is [B, H, W] float32 / uint8 / uint16 (C-contiguous),
    the result [B, dH, dW] float32 (written into `out` when given, e.g. a PinnedArray's array).
    Returns (code, message, dst or None, Layout or None)."""
    lib = L.load()
    a = np.ascontiguousarray(srcs)
    if a.ndim != 3 or a.dtype not in _NP_DTYPES:
        raise ValueError("srcs must be a [B, H, W] array of float32, uint8 or uint16")
    B, H, W = a.shape
    rq = make_request(W, H, src_resolution, dst_resolution, src_isocenter, rotation_angle, mode, policy)
    rc, msg, lay = query(rq)
    if rc != L.OK:
        return rc, msg, None, None
    shape = (B, lay.dst_height, lay.dst_width)
    dst = np.empty(shape, dtype=np.float32) if out is None else out
    if dst.shape != shape or dst.dtype != np.float32 or not dst.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float32 array of shape %r" % (shape,))
    out_lay = L.Layout()
    rc = lib.aai_resample_batch_host(_ref(rq), B, a.ctypes.data, _NP_DTYPES[a.dtype], W, W * H,
                                     dst.ctypes.data, max(lay.dst_width, 1), lay.dst_width * lay.dst_height, ctypes.byref(out_lay))
    if rc != L.OK:
        return rc, last_error(), None, None
    return rc, "", dst, out_lay


def resample_interleaved_host(src, src_resolution, dst_resolution, src_isocenter, rotation_angle,
                              mode=L.MODE_AREA, policy=L.POLICY_REFERENCE):
    """Interleaved channels from host memory (aai_resample_interleaved_host): src is [H, W, C] float32 / uint8 / uint16
    with C in 1..4, the result [dH, dW, C] float32.  Returns (code, message, dst or None, Layout or None)."""
    lib = L.load()
    a = np.ascontiguousarray(src)
    if a.ndim != 3 or a.dtype not in _NP_DTYPES:
        raise ValueError("src must be an [H, W, C] array of float32, uint8 or uint16")
    H, W, C = a.shape
    rq = make_request(W, H, src_resolution, dst_resolution, src_isocenter, rotation_angle, mode, policy)
    rc, msg, lay = query(rq)
    if rc != L.OK:
        return rc, msg, None, None
    dst = np.empty((lay.dst_height, lay.dst_width, C), dtype=np.float32)
    out_lay = L.Layout()
    rc = lib.aai_resample_interleaved_host(_ref(rq), C, a.ctypes.data, _NP_DTYPES[a.dtype], W * C,
                                           dst.ctypes.data, max(lay.dst_width * C, 1), ctypes.byref(out_lay))
    if rc != L.OK:
        return rc, last_error(), None, None
    return rc, "", dst, out_lay


def resample_interleaved_device(request, channels, src_ptr, src_stride, dst_ptr, dst_stride, stream=0, batch=1,
                                src_image_stride=0, dst_image_stride=0, src_dtype=L.DTYPE_F32):
    """Device-resident interleaved images (aai_resample_interleaved_device); strides in elements."""
    lib = L.load()
    rc = lib.aai_resample_interleaved_device(_ref(request), int(batch), int(channels), src_ptr, int(src_dtype), src_stride,
                                             src_image_stride, dst_ptr, dst_stride, dst_image_stride, stream)
    if rc != L.OK:
        raise AaiError(rc, last_error())


def resample_device(request, src_ptr, src_stride, dst_ptr, dst_stride, stream=0, batch=None,
                    src_image_stride=0, dst_image_stride=0, src_dtype=L.DTYPE_F32):
    """Device-resident path: raw device pointers (ints) and a hipStream_t handle (int, 0 = default).
    src_dtype: DTYPE_F32 (default) / DTYPE_U8 / DTYPE_U16 -- strides are in source elements."""
    lib = L.load()
    if src_dtype != L.DTYPE_F32:
        rc = lib.aai_resample_batch_device(_ref(request), 1 if batch is None else int(batch), src_ptr, int(src_dtype),
                                           src_stride, src_image_stride, dst_ptr, dst_stride, dst_image_stride, stream)
        if rc != L.OK:
            raise AaiError(rc, last_error())
        return
    if batch is None:
        rc = lib.aai_resample_device_f32(_ref(request), src_ptr, src_stride, dst_ptr, dst_stride, stream)
    else:
        rc = lib.aai_resample_batch_device_f32(_ref(request), int(batch), src_ptr, src_stride, src_image_stride,
                                               dst_ptr, dst_stride, dst_image_stride, stream)
    if rc != L.OK:
        raise AaiError(rc, last_error())


def resample_multi_device(request, shards, src_stride, src_image_stride, dst_stride, dst_image_stride):
    """aai_resample_batch_multi_device_f32: `shards` is a list of (device, count, src_ptr, dst_ptr, stream) -- one batch of
    independent images spread over several GPUs of this process, no collective."""
    n = len(shards)
    devs = (ctypes.c_int32 * n)(*[int(s[0]) for s in shards])
    cnts = (ctypes.c_int32 * n)(*[int(s[1]) for s in shards])
    srcs = (ctypes.c_void_p * n)(*[s[2] for s in shards])
    dsts = (ctypes.c_void_p * n)(*[s[3] for s in shards])
    strs = (ctypes.c_void_p * n)(*[s[4] for s in shards])
    rc = L.load().aai_resample_batch_multi_device_f32(_ref(request), n, devs, cnts, srcs, src_stride, src_image_stride,
                                                     dsts, dst_stride, dst_image_stride, strs)
    if rc != L.OK:
        raise AaiError(rc, last_error())


def band_source_rows(request, dst_row0, dst_row1):
    """aai_band_source_rows: source rows [a, b) that dst rows [dst_row0, dst_row1) read.  Needs no GPU."""
    a, b = ctypes.c_int32(), ctypes.c_int32()
    rc = L.load().aai_band_source_rows(_ref(request), int(dst_row0), int(dst_row1), ctypes.byref(a), ctypes.byref(b))
    if rc != L.OK:
        raise AaiError(rc, last_error())
    return a.value, b.value


def resample_band_device(request, dst_row0, dst_row1, src_rows_ptr, src_stride, dst_rows_ptr, dst_stride, stream=0):
    """aai_resample_band_device_f32: src_rows_ptr addresses source row band_source_rows(...)[0], dst_rows_ptr output row dst_row0."""
    rc = L.load().aai_resample_band_device_f32(_ref(request), int(dst_row0), int(dst_row1), src_rows_ptr, src_stride,
                                               dst_rows_ptr, dst_stride, stream)
    if rc != L.OK:
        raise AaiError(rc, last_error())


def plan_shape(request, channels=1):
    """aai_plan_info: one line describing the cached whole-image plan of this request on the current device ("" if none):
    kernel family, K1 launch shape and its origin, flagged pixels, fp32 formulation, build time."""
    buf = ctypes.create_string_buffer(512)
    rc = L.load().aai_plan_info(_ref(request), int(channels), buf, 512)
    if rc != L.OK:
        raise AaiError(rc, last_error())
    return buf.value.decode()


def debug_cell_min_waves(waves):
    """tests: 0 = every rotated area request made through this module carries AAI_POLICY_PREFER_CELL (small outputs take the cell
    kernel too); anything else = the default (outputs below ~720 x 720 pixels stay on the quad kernel)"""
    global _extra_policy
    _extra_policy = (_extra_policy | L.POLICY_PREFER_CELL) if int(waves) == 0 else (_extra_policy & ~L.POLICY_PREFER_CELL)


def debug_skip_fixup(skip):
    """tests / timing: True = requests made through this module carry AAI_POLICY_DIAG_NO_FIXUP (the double-precision fix-up pass is
    not launched, the plan's listed pixels keep the caller's bytes)"""
    global _extra_policy
    _extra_policy = (_extra_policy | L.POLICY_DIAG_NO_FIXUP) if skip else (_extra_policy & ~L.POLICY_DIAG_NO_FIXUP)


def shutdown():
    """aai_shutdown: drop every cached plan now (optional; a process may simply exit)."""
    L.load().aai_shutdown()


def prepare(request, channels=1):
    """aai_prepare: build (and cache) the plan of this request on the current device now -- K1 tables, the one-off
    scans of a rotated geometry -- instead of inside the first resampling call, which would then synchronise."""
    rc = L.load().aai_prepare(_ref(request), int(channels))
    if rc != L.OK:
        raise AaiError(rc, last_error())


def synth_rows_device(dst_ptr, width, height, row0, row1, stride, seed, stream=0):
    """rows [row0, row1) of the synthetic width x height image; dst_ptr addresses row row0"""
    rc = L.load().aai_synth_rows_device_f32(dst_ptr, int(width), int(height), int(row0), int(row1), int(stride), int(seed), stream)
    if rc != L.OK:
        raise AaiError(rc, last_error())


def synth_device(dst_ptr, width, height, stride, seed, stream=0):
    rc = L.load().aai_synth_device_f32(dst_ptr, int(width), int(height), int(stride), int(seed), stream)
    if rc != L.OK:
        raise AaiError(rc, last_error())


class AreaAverageInterpolation:
    """Stateless, like the reference class (no data members, Source.cpp:52-54)."""

    def __init__(self, policy=L.POLICY_REFERENCE):
        self.policy = policy

    def _call(self, mode, src, srcResolution, dstResolution, srcIsocenter, rotationAngle):
        rc, msg, dst, iso, _ = resample_host(src, srcResolution, dstResolution, srcIsocenter, rotationAngle,
                                             mode=mode, policy=self.policy)
        if rc in (L.ERR_RESOLUTION_MISMATCH, L.ERR_RESOLUTION_NONPOSITIVE, L.ERR_NO_ROWS, L.ERR_NO_COLUMNS):
            return (False, msg), None, None          # the reference's {false, message}
        if rc != L.OK:
            raise AaiError(rc, msg)                  # conditions the reference cannot report
        return (True, ""), dst, iso

    def areaAverageInterpolation(self, src, srcResolution, dstResolution, srcIsocenter, rotationAngle):
        """Source.cpp:55"""
        return self._call(L.MODE_AREA, src, srcResolution, dstResolution, srcIsocenter, rotationAngle)

    def fastAreaAverageInterpolation(self, src, srcResolution, dstResolution, srcIsocenter, rotationAngle):
        """Source.cpp:584"""
        return self._call(L.MODE_FAST, src, srcResolution, dstResolution, srcIsocenter, rotationAngle)

    # build-defined comparison paths (README.md:8 of the reference names them; it implements neither)
    def bilinearInterpolation(self, src, srcResolution, dstResolution, srcIsocenter, rotationAngle):
        return self._call(L.MODE_BILINEAR, src, srcResolution, dstResolution, srcIsocenter, rotationAngle)

    def bicubicInterpolation(self, src, srcResolution, dstResolution, srcIsocenter, rotationAngle):
        return self._call(L.MODE_BICUBIC, src, srcResolution, dstResolution, srcIsocenter, rotationAngle)
