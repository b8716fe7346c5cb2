"""TEST INFRASTRUCTURE ONLY -- ctypes bindings for the checkers under oracle/.

* ``ref_run``      drives oracle/_ref/libaai_ref.so = the UNMODIFIED reference
                   (/root/reference/Source.cpp:55 and :584) built by oracle/Makefile.
* ``oracle_run``   drives oracle/liboracle.so = the plain-C restatement (oracle/aai_oracle.c).
* ``synth_image``  SURVEY.md Appendix C.1 stateless hash -> fp32 uniform [0,1).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (area_average_interpolation_amd/) must never import it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SO = os.path.join(_HERE, "_ref", "libaai_ref.so")
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")

ERR_RES_MISMATCH = "Assumed X & Y resolution are same."            # Source.cpp:115
ERR_RES_NONPOS = "0 or negative resolution is not acceptable."     # Source.cpp:120
ERR_NO_ROWS = "There is no data in src array."                     # Source.cpp:125
ERR_NO_COLS = "There is no data in the second dimension of src array."  # Source.cpp:130

MODE_EXACT = 1   # areaAverageInterpolation      (Source.cpp:55)
MODE_FAST = 2    # fastAreaAverageInterpolation  (Source.cpp:584)


def build(verbose=False):
    """Compile liboracle.so and (when /root/reference is present) _ref/libaai_ref.so."""
    r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout + r.stderr)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed")


def have_ref():
    return os.path.exists(_REF_SO)


def have_oracle():
    return os.path.exists(_ORACLE_SO)


_ref = None
_orc = None


def _load_ref():
    global _ref
    if _ref is None:
        lib = ctypes.CDLL(_REF_SO)
        lib.aai_ref_run.restype = ctypes.c_int
        lib.aai_ref_run.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
            ctypes.c_double, ctypes.c_double, ctypes.c_double,
            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
            ctypes.c_char_p, ctypes.c_int]
        lib.aai_ref_run_empty.restype = ctypes.c_int
        lib.aai_ref_run_empty.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        lib.aai_ref_free.restype = None
        lib.aai_ref_free.argtypes = [ctypes.c_void_p]
        _ref = lib
    return _ref


def _load_oracle():
    global _orc
    if _orc is None:
        lib = ctypes.CDLL(_ORACLE_SO)
        lib.aai_oracle_run.restype = ctypes.c_int
        lib.aai_oracle_run.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
            ctypes.c_double, ctypes.c_double, ctypes.c_double,
            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
            ctypes.c_char_p, ctypes.c_int]
        lib.aai_oracle_free.restype = None
        lib.aai_oracle_free.argtypes = [ctypes.c_void_p]
        lib.aai_oracle_synth_f32.restype = None
        lib.aai_oracle_synth_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        _orc = lib
    return _orc


class Result:
    def __init__(self, ok, msg, dst, dst_iso):
        self.ok, self.msg, self.dst, self.dst_iso = ok, msg, dst, dst_iso


def _run(lib_fn, free_fn, head_args, src, src_res, dst_res, iso, angle):
    src = np.ascontiguousarray(src, dtype=np.float64)
    H, W = src.shape
    if np.isscalar(src_res):
        src_res = (src_res, src_res)
    if np.isscalar(dst_res):
        dst_res = (dst_res, dst_res)
    out = ctypes.c_void_p()
    dW, dH = ctypes.c_int(), ctypes.c_int()
    ix, iy = ctypes.c_double(), ctypes.c_double()
    err = ctypes.create_string_buffer(256)
    ok = lib_fn(*head_args, src.ctypes.data, W, H,
                float(src_res[0]), float(src_res[1]), float(dst_res[0]), float(dst_res[1]),
                float(iso[0]), float(iso[1]), float(angle),
                ctypes.byref(out), ctypes.byref(dW), ctypes.byref(dH),
                ctypes.byref(ix), ctypes.byref(iy), err, 256)
    if not ok:
        return Result(False, err.value.decode(), None, None)
    n = dW.value * dH.value
    buf = (ctypes.c_double * max(n, 1)).from_address(out.value)
    dst = np.frombuffer(buf, dtype=np.float64, count=n).reshape(dH.value, dW.value).copy()
    free_fn(out)
    return Result(True, "", dst, (ix.value, iy.value))


def ref_run(mode, src, src_res, dst_res, iso, angle):
    """Run the unmodified reference.  mode: MODE_EXACT / MODE_FAST."""
    lib = _load_ref()
    return _run(lib.aai_ref_run, lib.aai_ref_free, (int(mode),), src, src_res, dst_res, iso, angle)


def ref_run_empty(mode, rows):
    lib = _load_ref()
    err = ctypes.create_string_buffer(256)
    ok = lib.aai_ref_run_empty(int(mode), int(rows), err, 256)
    return bool(ok), err.value.decode()


POLICY_REFERENCE = 0   # reproduce Source.cpp:1055-1062 as written (graded)
POLICY_EXACT = 1       # geometrically exact corner triangles (ungraded)


def oracle_run(mode, src, src_res, dst_res, iso, angle, policy=POLICY_REFERENCE):
    """Run the CPU restatement (oracle/aai_oracle.c)."""
    lib = _load_oracle()
    return _run(lib.aai_oracle_run, lib.aai_oracle_free, (int(mode), int(policy)), src, src_res, dst_res, iso, angle)


def oracle_rows(mode, src_f32, src_res, dst_res, iso, angle, row0, row1, dst_width, policy=POLICY_REFERENCE):
    """Rows [row0, row1) of the output only (aai_oracle_rows): for images too large for a full CPU run.  src_f32 is an
    [H, W] float32 array; returns a float64 array [row1 - row0, dst_width]."""
    lib = _load_oracle()
    lib.aai_oracle_rows.restype = ctypes.c_int
    lib.aai_oracle_rows.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
        [ctypes.c_double] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    a = np.ascontiguousarray(src_f32, dtype=np.float32)
    H, W = a.shape
    out = np.empty((row1 - row0, dst_width), np.float64)
    err = ctypes.create_string_buffer(256)
    ok = lib.aai_oracle_rows(int(mode), int(policy), a.ctypes.data, 1, W, H, float(src_res), float(src_res), float(dst_res), float(dst_res),
                             float(iso[0]), float(iso[1]), float(angle), int(row0), int(row1), out.ctypes.data, err, 256)
    if not ok:
        raise RuntimeError(err.value.decode())
    return out


def oracle_pixels(mode, src_f32, src_res, dst_res, iso, angle, xs, ys, policy=POLICY_REFERENCE):
    """A list of dst pixels (aai_oracle_pixels): unbiased samples of images too large for a full CPU run (~80 us per pixel at
    3:1).  src_f32 is an [H, W] float32 array; returns float64 values, one per (xs[k], ys[k])."""
    lib = _load_oracle()
    lib.aai_oracle_pixels.restype = ctypes.c_int
    lib.aai_oracle_pixels.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
        [ctypes.c_double] * 7 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    a = np.ascontiguousarray(src_f32, dtype=np.float32)
    H, W = a.shape
    x = np.ascontiguousarray(xs, dtype=np.int32)
    y = np.ascontiguousarray(ys, dtype=np.int32)
    assert x.shape == y.shape and x.ndim == 1
    out = np.empty(x.shape[0], np.float64)
    err = ctypes.create_string_buffer(256)
    ok = lib.aai_oracle_pixels(int(mode), int(policy), a.ctypes.data, 1, W, H, float(src_res), float(src_res), float(dst_res), float(dst_res),
                               float(iso[0]), float(iso[1]), float(angle), int(x.shape[0]), x.ctypes.data, y.ctypes.data, out.ctypes.data, err, 256)
    if not ok:
        raise RuntimeError(err.value.decode())
    return out


def synth_image(W, H, seed=1):
    """SURVEY.md Appendix C.1 generator, vectorised numpy (uint64 wraparound arithmetic)."""
    with np.errstate(over="ignore"):
        g = np.uint64(0x9E3779B97F4A7C15)
        idx = np.arange(W * H, dtype=np.uint64)
        z = np.uint64(seed) * g + idx + g
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        v = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return v.reshape(H, W)


def dose_image(W, H, seed=1, field=0.44, dmax=250.0):
    """A film-dose-like test image (the kind of input the reference was written for, Source.cpp:1529): a flat-topped square
    field with error-function penumbrae, scatter tails that fall to ~1e-4 of the maximum at the image border, and 1 % film
    noise from the Appendix C.1 hash -- three to four decades of dynamic range next to each other, which uniform noise
    never has.  Deterministic in (W, H, seed); float32, like every image the GPU path takes."""
    from math import erf
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    cx, cy = (W - 1) * 0.5 + 3.25, (H - 1) * 0.5 - 2.5            # field centre off the image centre
    half = field * min(W, H) * 0.5
    sigma = 0.012 * min(W, H)                                     # penumbra width
    verf = np.vectorize(erf)
    px = 0.5 * (verf((half - np.abs(x - cx)) / (sigma * 2 ** 0.5)) + 1.0)
    py = 0.5 * (verf((half - np.abs(y - cy)) / (sigma * 2 ** 0.5)) + 1.0)
    r = np.maximum(np.maximum(np.abs(x - cx), np.abs(y - cy)) - half, 0.0)
    tail = 8e-3 * np.exp(-r / (0.05 * min(W, H))) + 1e-4          # out-of-field scatter and film base
    noise = synth_image(W, H, seed).astype(np.float64)
    dose = dmax * (px * py * (1.0 - tail) + tail) * (1.0 + 0.01 * (noise - 0.5))
    return dose.astype(np.float32)
