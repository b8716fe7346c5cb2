"""Shared fixtures.  `-m "not gpu"` runs in a container without a GPU; `-m gpu` runs on an MI355X box
(where /root/reference does not exist: nothing here reads it at run time)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
BUILD = os.path.join(ROOT, "tests", "_build")

# Golden cases sitting on DBL_EPSILON knife edges of the reference's classifier: a dst edge
# passes exactly through source-pixel corners (reduced angle 30/45/60 degrees with commensurate sizes), so the
# reference's answer for a handful of pixels is decided by its own last-bit rounding.  The CPU oracle
# reproduces them bit for bit, and since the strict replay path (csrc/aai_strict.hpp) the GPU path matches
# them too: NO pixel of any golden case is allowed to differ.  The indices are kept so that tests can
# assert these cases really exercise the knife-edge machinery.
KNIFE_EDGE_CASES = [(48, "exact"), (49, "exact"), (73, "exact"), (105, "fast"), (108, "exact"), (109, "exact"),
                    (131, "exact"), (132, "exact")]
KNIFE_EDGE = {}      # (case, tag) -> pixels allowed to differ: none


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


@pytest.fixture(scope="session")
def po():
    """oracle bindings (builds liboracle.so; builds oracle/_ref only where the reference is present)."""
    from oracle import pyoracle
    if not pyoracle.have_oracle() or (os.path.exists("/root/reference/Source.cpp") and not pyoracle.have_ref()):
        pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def aai():
    """the product package; builds libaai_hip.so if it is missing"""
    so = os.path.join(ROOT, "area_average_interpolation_amd", "libaai_hip.so")
    if not os.path.exists(so):
        _run(["make", "-C", os.path.join(ROOT, "area_average_interpolation_amd", "csrc")])
    import area_average_interpolation_amd as pkg
    return pkg


@pytest.fixture(scope="session")
def hostemu(aai):
    """tests/emulation/host_emulation.cpp compiled with g++ (serial replay of the kernels' arithmetic)"""
    from area_average_interpolation_amd import _lib as L
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libaai_hostemu.so")
    srcs = [os.path.join(ROOT, "tests", "emulation", "host_emulation.cpp"),
            os.path.join(ROOT, "area_average_interpolation_amd", "csrc", "aai_plan.cpp"),
            os.path.join(ROOT, "area_average_interpolation_amd", "csrc", "aai_plan.hpp"),
            os.path.join(ROOT, "area_average_interpolation_amd", "csrc", "aai_rot_math.hpp"),
            os.path.join(ROOT, "area_average_interpolation_amd", "csrc", "aai_rot_quad.hpp"),
            os.path.join(ROOT, "area_average_interpolation_amd", "csrc", "aai_rot_cell.hpp"),
            os.path.join(ROOT, "area_average_interpolation_amd", "csrc", "aai_strict.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, srcs[0]])
    lib = ctypes.CDLL(so)
    lib.aai_emu_resample.restype = ctypes.c_int
    lib.aai_emu_resample.argtypes = [ctypes.POINTER(L.Request), ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.aai_emu_strip_stats.restype = ctypes.c_int
    lib.aai_emu_strip_stats.argtypes = [ctypes.POINTER(L.Request)] + [ctypes.POINTER(ctypes.c_int)] * 4

    def resample(rq, src):
        src = np.ascontiguousarray(src, dtype=np.float32)
        dW, dH, ax = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = lib.aai_emu_resample(ctypes.byref(rq), src.ctypes.data, None, ctypes.byref(dW), ctypes.byref(dH), ctypes.byref(ax))
        assert rc == 0, rc
        out = np.empty((dH.value, dW.value), np.float32)
        lib.aai_emu_resample(ctypes.byref(rq), src.ctypes.data, out.ctypes.data, ctypes.byref(dW), ctypes.byref(dH), ctypes.byref(ax))
        return out, bool(ax.value)

    def strip_stats(rq):
        v = [ctypes.c_int() for _ in range(4)]
        rc = lib.aai_emu_strip_stats(ctypes.byref(rq), *[ctypes.byref(x) for x in v])
        return rc, [x.value for x in v]

    lib.aai_emu_axis_invariants.restype = ctypes.c_int
    lib.aai_emu_axis_invariants.argtypes = [ctypes.POINTER(L.Request)]
    lib.aai_emu_resample_channels.restype = ctypes.c_int
    lib.aai_emu_resample_channels.argtypes = [ctypes.POINTER(L.Request), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.aai_emu_check_line_runs.restype = ctypes.c_long
    lib.aai_emu_check_line_runs.argtypes = [ctypes.POINTER(L.Request)]
    lib.aai_emu_wide_band_cover.restype = ctypes.c_long
    lib.aai_emu_wide_band_cover.argtypes = [ctypes.POINTER(L.Request), ctypes.c_int, ctypes.c_int]
    lib.aai_emu_live_spans.restype = ctypes.c_int
    lib.aai_emu_live_spans.argtypes = [ctypes.POINTER(L.Request), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.aai_emu_wide_parts.restype = ctypes.c_int
    lib.aai_emu_wide_parts.argtypes = [ctypes.POINTER(L.Request)]
    lib.aai_emu_uses_runs.restype = ctypes.c_int
    lib.aai_emu_uses_runs.argtypes = [ctypes.POINTER(L.Request)]
    lib.aai_emu_force_general.restype = None
    lib.aai_emu_force_general.argtypes = [ctypes.c_int]
    lib.aai_emu_set_strict.restype = None
    lib.aai_emu_set_strict.argtypes = [ctypes.c_int]
    lib.aai_emu_knife_stats.restype = None
    lib.aai_emu_knife_stats.argtypes = [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]

    def knife_stats():
        a, b = ctypes.c_long(), ctypes.c_long()
        lib.aai_emu_knife_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    lib.knife_stats = knife_stats
    lib.aai_emu_missed_knife_pairs.restype = ctypes.c_long
    lib.aai_emu_missed_knife_pairs.argtypes = []
    # the fp32 quad formulation of the rotated area kernel (csrc/aai_rot_quad.hpp)
    lib.aai_emu_axis_fixups.restype = ctypes.c_long
    lib.aai_emu_skip_axis_fixup.argtypes = [ctypes.c_int]
    lib.aai_emu_use_quad.restype = None
    lib.aai_emu_use_quad.argtypes = [ctypes.c_int]
    lib.aai_emu_axis_class_verify.restype = ctypes.c_long
    lib.aai_emu_axis_class_verify.argtypes = [ctypes.POINTER(L.Request), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
    lib.aai_emu_cell_live_rows_check.restype = ctypes.c_long
    lib.aai_emu_cell_live_rows_check.argtypes = [ctypes.POINTER(L.Request)]
    lib.aai_emu_fast_tile_cover.restype = ctypes.c_long
    lib.aai_emu_fast_tile_cover.argtypes = [ctypes.POINTER(L.Request), ctypes.POINTER(ctypes.c_long)]
    lib.aai_emu_replicated_indices.restype = ctypes.c_long
    lib.aai_emu_replicated_indices.argtypes = [ctypes.c_int] * 4
    lib.aai_emu_band_cover.restype = ctypes.c_long
    lib.aai_emu_band_cover.argtypes = [ctypes.POINTER(L.Request), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_long)]
    lib.aai_emu_cell_band_cover.restype = ctypes.c_long
    lib.aai_emu_cell_band_cover.argtypes = [ctypes.POINTER(L.Request), ctypes.c_int, ctypes.c_int]
    lib.aai_emu_use_cell.restype = None
    lib.aai_emu_use_cell.argtypes = [ctypes.c_int]
    lib.aai_emu_quad_stats.restype = None
    lib.aai_emu_quad_stats.argtypes = [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
    lib.aai_emu_quad_pair_check.restype = ctypes.c_long
    lib.aai_emu_quad_pair_check.argtypes = [ctypes.POINTER(L.Request), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]

    def quad_stats():
        a, b = ctypes.c_long(), ctypes.c_long()
        lib.aai_emu_quad_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def quad_pair_check(rq):
        a, b = ctypes.c_double(), ctypes.c_double()
        n = lib.aai_emu_quad_pair_check(ctypes.byref(rq), ctypes.byref(a), ctypes.byref(b))
        return n, a.value, b.value

    lib.quad_stats = quad_stats
    lib.quad_pair_check = quad_pair_check
    lib.resample = resample
    lib.strip_stats = strip_stats
    return lib


@pytest.fixture(scope="session")
def small_golden():
    z = np.load(os.path.join(GOLDEN, "small_cases.npz"))
    manifest = json.loads(bytes(z["manifest"]).decode())
    return z, manifest


@pytest.fixture(scope="session")
def knife_golden():
    """outputs of the UNMODIFIED reference on the structured knife-edge geometries (tests/golden/make_golden.py knife)"""
    z = np.load(os.path.join(GOLDEN, "knife_cases.npz"))
    manifest = json.loads(bytes(z["manifest"]).decode())
    return z, manifest


@pytest.fixture(scope="session")
def axis_knife_golden():
    """outputs of the UNMODIFIED reference at rotations 0/90/180/270 with edges on pixel boundaries / through pixel centres
    (tests/golden/make_golden.py axisknife)"""
    z = np.load(os.path.join(GOLDEN, "axis_knife_cases.npz"))
    manifest = json.loads(bytes(z["manifest"]).decode())
    return z, manifest


@pytest.fixture(scope="session")
def refdefault_golden():
    """the reference's own default call (Source.cpp:1528-1534: 150 -> 25.4 dpi about (455, 455), 1.5 degrees) on a 911 x 911
    dose-like image, both modes, outputs of the UNMODIFIED reference (tests/golden/make_golden.py refdefault)"""
    z = np.load(os.path.join(GOLDEN, "refdefault.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def load_full(name):
    z = np.load(os.path.join(GOLDEN, "full_%s.npz" % name))
    return z, json.loads(bytes(z["meta"]).decode())


def sample_points(rq, lay, rows, device):
    """Continuous original-image coordinates (pixel centres at integers) of the dst pixel centres of the given dst rows:
    the reference's affine map restated from SURVEY.md Appendix A.1-A.3 (Source.cpp:139-219) in float64 torch."""
    import math
    import torch
    W, H = rq.src_width, rq.src_height
    rho = rq.dst_res_x / rq.src_res_x
    scale = int(rho * math.sqrt(2.0) + 1 + 2.220446049250313e-16)
    ang = rq.rotation_deg % 360.0
    quadrant = int(ang // 90)
    th = math.radians(ang - 90 * quadrant)
    c, s = math.cos(th), math.sin(th)
    mW, mH = ((H, W) if quadrant & 1 else (W, H))
    mW, mH = mW * scale, mH * scale
    isoX = rq.src_iso_x * scale + (scale - 1) / 2.0
    isoY = rq.src_iso_y * scale + (scale - 1) / 2.0
    r = rho / scale
    L = 1.0 / r
    Dx = (isoX * c + (mH - isoY) * s) * r
    Dy = (isoX * s + isoY * c) * r
    fx, fy = Dx - int(Dx), Dy - int(Dy)
    corners = [(-isoX, -isoY), (mW - 1 - isoX, -isoY), (-isoX, mH - 1 - isoY), (mW - 1 - isoX, mH - 1 - isoY)]
    offX = min([0.0] + [u * c - v * s + isoX for (u, v) in corners])
    offY = min([0.0] + [u * s + v * c + isoY for (u, v) in corners])
    assert (lay.scale, lay.quadrant) == (scale, quadrant)
    dx = torch.arange(lay.dst_width, dtype=torch.float64, device=device)[None, :]
    dy = torch.as_tensor(rows, dtype=torch.float64, device=device)[:, None]
    Px = (dx + fx) * L - isoX + offX
    Py = (dy + fy) * L - isoY + offY
    X = Px * c + Py * s + isoX
    Y = -Px * s + Py * c + isoY
    if quadrant == 0:
        sx, sy = (X + 0.5) / scale - 0.5, (Y + 0.5) / scale - 0.5
    elif quadrant == 1:
        sx, sy = (Y + 0.5) / scale - 0.5, (mW - 1 - X + 0.5) / scale - 0.5
    elif quadrant == 2:
        sx, sy = (mW - 1 - X + 0.5) / scale - 0.5, (mH - 1 - Y + 0.5) / scale - 0.5
    else:
        sx, sy = (mH - 1 - Y + 0.5) / scale - 0.5, (X + 0.5) / scale - 0.5
    return sx, sy


def rel_err(got, gold, floor=1e-3):
    """|got-gold| / max(|gold|, floor): relative error with an absolute floor for near-zero pixels
    (synthetic images are uniform [0,1), so 1e-3 is far below any non-empty pixel's value scale)."""
    got = np.asarray(got, dtype=np.float64)
    gold = np.asarray(gold, dtype=np.float64)
    return np.abs(got - gold) / np.maximum(np.abs(gold), floor)


TOL = 1e-5   # BASELINE.json north_star: outputs within 1e-5 relative of the reference CPU path


RUNS_CASES = [  # (W, H, srcRes, dstRes, angle, iso offset): footprints wide enough for the rows-as-runs kernel
    (96, 80, 5.5, 1.0, 17.5, (0.0, 0.0)), (96, 96, 6.0, 1.0, 0.5, (0.3, -0.2)), (120, 90, 8.0, 1.0, 33.3, (0.0, 0.0)),
    (128, 128, 12.0, 1.0, 17.5, (1.5, 2.5)), (100, 140, 7.5, 1.0, 117.5, (0.0, 0.0)), (140, 100, 6.5, 1.0, 200.25, (-3.0, 4.0)),
    (96, 96, 9.0, 1.0, 305.0, (0.0, 0.0)), (128, 96, 16.0, 1.0, 45.0, (0.0, 0.0)), (96, 128, 8.0, 1.0, 30.0, (0.0, 0.0)),
    (128, 128, 8.0, 1.0, 60.0, (0.5, 0.5)), (90, 90, 5.5, 1.0, 1e-6, (0.0, 0.0)), (90, 90, 5.5, 1.0, 89.999999, (0.0, 0.0)),
    (64, 64, 40.0, 1.0, 17.5, (0.0, 0.0)), (200, 40, 6.0, 1.0, 12.0, (0.0, 0.0)),
]
