"""GPU (MI355X): parity of the HIP path against the reference, through the C ABI.

Comparisons are against (a) the committed golden vectors produced by the unmodified reference,
(b) the CPU oracle (oracle/aai_oracle.c, itself bit-exact against (a)) on the same seeded inputs, and
(c) size-independent properties at BASELINE.json's full sizes.  Tolerance: 1e-5 relative
(BASELINE.json north_star), exact zeros must stay exact.  Nothing here reads /root/reference.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import KNIFE_EDGE, KNIFE_EDGE_CASES, ROOT, RUNS_CASES, TOL, load_full, rel_err, sample_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(aai):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from area_average_interpolation_amd import _lib as L
    L.load()                       # raises if libaai_hip.so is missing: no silent fallback
    assert aai.device_count() >= 1
    aai.set_device(0)
    return aai


@pytest.fixture(params=["default", "cell"])
def formulation(request, gpu):
    """Small rotated area requests stay on the quad kernel by default (too few cell waves to fill the chip); "cell" sends them
    to the cell kernel as well, so that the golden vectors and the oracle check BOTH fp32 formulations on every case."""
    gpu.debug_cell_min_waves(0 if request.param == "cell" else -1)
    yield request.param
    gpu.debug_cell_min_waves(-1)


def _host(gpu, src, c, mode, policy=0):
    rc, msg, dst, iso, lay = gpu.resample_host(src, c["src_res"], c["dst_res"], c["iso"], c["angle"], mode=mode, policy=policy)
    assert rc == 0, msg
    return dst, iso, lay


# ---- (a) golden vectors of the unmodified reference ---------------------------------------------------------
def test_small_golden_cases_f32(gpu, formulation, po, small_golden):
    z, manifest = small_golden
    for i, c in enumerate(manifest):
        src = po.synth_image(c["W"], c["H"], c["seed"])
        for mode, tag in ((1, "exact"), (2, "fast")):
            dst, iso, lay = _host(gpu, src, c, mode)
            gold = z["c%03d_%s" % (i, tag)]
            assert dst.dtype == np.float32 and dst.shape == gold.shape, (i, tag)
            assert list(iso) == c["dst_iso"]
            bad = int((rel_err(dst, gold) > TOL).sum())
            allowed = KNIFE_EDGE.get((i, tag), 0)
            assert bad <= allowed, (i, tag, c, bad, gpu.last_kernel())
            if allowed == 0:
                assert np.array_equal(gold == 0, dst == 0), (i, tag)


def test_small_golden_cases_f64_entry(gpu, po, small_golden):
    """aai_resample_f64: the reference's own element type (IMG = vector<vector<double>>)."""
    z, manifest = small_golden
    for i in range(0, len(manifest), 7):
        c = manifest[i]
        if (i, "exact") in KNIFE_EDGE:
            continue
        src = po.synth_image(c["W"], c["H"], c["seed"]).astype(np.float64)
        dst, iso, lay = _host(gpu, src, c, 1)
        assert dst.dtype == np.float64
        assert rel_err(dst, z["c%03d_exact" % i]).max() <= TOL, i


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5s", "cfg5"])
def test_baseline_configs_against_reference_known_answers(gpu, name):
    """BASELINE.json configs at full size: strided sample grid, complete rows, sum and zero count of the
    unmodified reference's output: cfg1-4, cfg5 at 1/8 linear scale (cfg5s) and, since round 3, the full 4096^2 -> 23170^2
    config 5 as well (2 h 26 min of the reference on one core; the oracle's known answers held there before were reproduced bit for bit)."""
    import torch
    z, meta = load_full(name)
    W, H = meta["W"], meta["H"]
    src = torch.empty((H, W), dtype=torch.float32, device="cuda")
    gpu.synth_device(src.data_ptr(), W, H, W, 1)
    for tag, mode in (("exact", 1), ("fast", 2)):
        if tag not in meta:
            continue
        m = meta[tag]
        rq = gpu.make_request(W, H, meta["src_res"], meta["dst_res"], meta["iso"], meta["angle"], mode=mode)
        rc, msg, lay = gpu.query(rq)
        assert rc == 0 and [lay.dst_height, lay.dst_width] == m["shape"]
        if m.get("dst_iso") is not None:
            assert [lay.dst_iso_x, lay.dst_iso_y] == m["dst_iso"]
        dst = torch.full((lay.dst_height, lay.dst_width), -1.0, dtype=torch.float32, device="cuda")
        gpu.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), lay.dst_width, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        step = m["step"]
        grid = dst[::step, ::step].cpu().numpy()
        rows = dst[m["rows"], :].cpu().numpy()
        assert rel_err(grid, z[tag + "_grid"]).max() <= TOL, (name, tag)
        assert rel_err(rows, z[tag + "_rows"]).max() <= TOL, (name, tag)
        assert np.array_equal(rows == 0, z[tag + "_rows"] == 0)
        assert int((dst == 0).sum().item()) == m["zeros"], (name, tag)
        total = float(dst.sum(dtype=torch.float64).item())
        assert abs(total - float(m["sum"])) <= 2e-7 * float(m["sum"]), (name, tag, total, m["sum"])
        del dst


def _ring(dW, dH):
    """every pixel of the canvas border ring"""
    xs = np.concatenate([np.arange(dW), np.arange(dW), np.zeros(dH - 2, int), np.full(dH - 2, dW - 1)])
    ys = np.concatenate([np.zeros(dW, int), np.full(dW, dH - 1), np.arange(1, dH - 1), np.arange(1, dH - 1)])
    return xs, ys


@pytest.mark.parametrize("name,image,samples,ring", [("cfg2", "noise", 4000, True), ("cfg3", "noise", 4000, True), ("cfg5s", "noise", 4000, True),
                                                      ("cfg5", "noise", 500, True), ("cfg3", "dose", 4000, False)])
def test_full_size_configs_at_random_pixels_and_on_the_canvas_border(gpu, po, name, image, samples, ring):
    """The known-answer test above looks at a fixed strided grid and seven rows -- always the same ~26 k pixels, never the canvas
    border as a whole.  Here: seeded-random dst pixels (4,000; 500 at the full config 5) plus EVERY pixel of the canvas border ring,
    both modes, against the CPU oracle evaluated pixel by pixel (aai_oracle_pixels: each pixel right after its predecessor in the
    reference's loop order) on the same image -- 1e-5 relative with an absolute floor of 1e-6 (not the 1e-3 of the fixture tests),
    exact zeros exact.  "dose": config 3's ratio and rotation on a 2048 x 2048 dose-like image (flat field, penumbrae, tails at 1e-4 of
    the maximum)."""
    import torch
    z, meta = load_full(name)
    W, H = meta["W"], meta["H"]
    if image == "dose":
        # (a quarter of the linear size: the dose image is synthesised on the host, ~6 s at 2048^2; same ratio, same rotation, still
        # large enough for the cell kernel)
        W, H = W // 4, H // 4
        meta = dict(meta, iso=[(W - 1) / 2, (H - 1) / 2])
        host = po.dose_image(W, H, 1)
        src = torch.from_numpy(host).cuda()
    else:
        host = po.synth_image(W, H, 1)
        src = torch.empty((H, W), dtype=torch.float32, device="cuda")
        gpu.synth_device(src.data_ptr(), W, H, W, 1)
        assert torch.equal(src[:4].cpu(), torch.from_numpy(host[:4]))
    rng = np.random.default_rng({"cfg2": 2, "cfg3": 3, "cfg5s": 55, "cfg5": 5}[name] + (100 if image == "dose" else 0))
    for tag, mode, omode in (("exact", 1, po.MODE_EXACT), ("fast", 2, po.MODE_FAST)):
        rq = gpu.make_request(W, H, meta["src_res"], meta["dst_res"], meta["iso"], meta["angle"], mode=mode)
        rc, msg, lay = gpu.query(rq)
        assert rc == 0, msg
        dW, dH = lay.dst_width, lay.dst_height
        dst = torch.full((dH, dW), -1.0, dtype=torch.float32, device="cuda")
        gpu.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        xs, ys = rng.integers(0, dW, samples), rng.integers(0, dH, samples)
        if ring:
            rx, ry = _ring(dW, dH)
            xs, ys = np.concatenate([xs, rx]), np.concatenate([ys, ry])
        gold = po.oracle_pixels(omode, host, meta["src_res"], meta["dst_res"], meta["iso"], meta["angle"], xs, ys)
        got = dst[torch.from_numpy(ys).cuda(), torch.from_numpy(xs).cuda()].cpu().numpy()
        err = rel_err(got, gold, floor=1e-6)
        worst = int(err.argmax())
        assert err.max() <= TOL, (name, image, tag, gpu.last_kernel(), int(xs[worst]), int(ys[worst]), float(got[worst]), float(gold[worst]))
        assert np.array_equal(got == 0, gold == 0), (name, image, tag)
        assert (gold[:samples] != 0).sum() > samples // 4            # (the random pixels are not all in the canvas's empty corners)
        del dst


# ---- (b) oracle on the same seeded inputs -------------------------------------------------------------------
def test_reference_default_call_on_a_dose_like_image(gpu, formulation, po, refdefault_golden):
    """The reference's own example call (Source.cpp:1528-1534: a 911 x 911 film at 150 dpi to 25.4 dpi about (455, 455), rotated
    by 1.5 degrees; mode 2 is its default) on a dose-like image -- flat field, penumbrae, tails at 1e-4 of the maximum, so
    neighbouring values differ by decades -- against the UNMODIFIED reference's output.  1e-5 relative with an absolute floor
    of 1e-6 (values run from 0.03 to 250), exact zeros exact; also through the double-precision policy."""
    z, meta = refdefault_golden
    src = po.dose_image(meta["W"], meta["H"], meta["seed"])
    for mode, tag in ((1, "exact"), (2, "fast")):
        for policy in (0, gpu.POLICY_DOUBLE_PRECISION):
            rc, msg, dst, iso, lay = gpu.resample_host(src, meta["src_res"], meta["dst_res"], meta["iso"], meta["angle"], mode=mode, policy=policy)
            assert rc == 0, msg
            gold = z[tag]
            assert dst.shape == gold.shape and list(iso) == meta[tag]["dst_iso"]
            err = rel_err(dst, gold, floor=1e-6)
            assert err.max() <= TOL, (tag, policy, gpu.last_kernel(), float(err.max()))
            assert np.array_equal(dst == 0, gold == 0), (tag, policy)


def test_random_geometries_against_oracle(gpu, formulation, po):
    rng = np.random.default_rng(11)
    for k in range(60):
        W, H = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        sr, dr = float(rng.uniform(0.5, 6)), float(rng.uniform(0.5, 3))
        if dr / sr > 2.0:
            dr = sr * 2.0
        ang = float(rng.uniform(-400, 400))
        if k % 6 == 0:
            ang = float(rng.choice([0, 90, 180, 270, 360, -90]))
        iso = (float(rng.uniform(-3, W + 3)), float(rng.uniform(-3, H + 3)))
        src = rng.random((H, W)).astype(np.float32)
        c = dict(src_res=sr, dst_res=dr, iso=iso, angle=ang)
        for mode, omode in ((1, po.MODE_EXACT), (2, po.MODE_FAST)):
            gold = po.oracle_run(omode, src.astype(np.float64), sr, dr, iso, ang)
            if gold.dst.size == 0:
                # an output extent that rounds to 0: the reference crashes there; the library reports it
                rc = gpu.resample_host(src, sr, dr, iso, ang, mode=mode)[0]
                assert rc == 10, (k, W, H, sr, dr, ang, rc)            # AAI_ERR_EMPTY_OUTPUT
                continue
            dst, giso, lay = _host(gpu, src, c, mode)
            assert dst.shape == gold.dst.shape and tuple(giso) == gold.dst_iso
            assert rel_err(dst, gold.dst).max() <= TOL, (k, mode, W, H, sr, dr, ang, iso, gpu.last_kernel())
            assert np.array_equal(gold.dst == 0, dst == 0), (k, mode)


def test_axis_kernel_code_paths_against_oracle(gpu, po):
    """Every branch of the axis-aligned path: one output per lane, many outputs per strip (ratios < 4 and
    up-sampling), the right-edge fix-up (widths not a multiple of 4), the wide fallback (footprints wider than
    a strip, or images narrower than 4 columns), all four quadrants, both modes."""
    from area_average_interpolation_amd import _lib as L
    rng = np.random.default_rng(17)
    cases = [  # W, H, srcRes, dstRes, expected kernel
        (517, 40, 4, 1, L.KERNEL_AXIS), (1030, 9, 8, 1, L.KERNEL_AXIS), (300, 33, 3, 1, L.KERNEL_AXIS),
        (301, 21, 2, 1, L.KERNEL_AXIS), (259, 17, 1, 1, L.KERNEL_AXIS), (70, 50, 1, 2, L.KERNEL_AXIS),
        (40, 30, 1, 4, L.KERNEL_AXIS), (263, 31, 8192, 2731, L.KERNEL_AXIS), (1500, 20, 10, 9, L.KERNEL_AXIS),
        (3000, 800, 700, 1, L.KERNEL_AXIS_WIDE), (3, 50, 2, 1, L.KERNEL_AXIS_WIDE), (1, 1, 1, 1, L.KERNEL_AXIS_WIDE),
        (2, 300, 1, 1, L.KERNEL_AXIS_WIDE), (4, 4, 2, 1, L.KERNEL_AXIS), (5, 700, 3, 1, L.KERNEL_AXIS),
    ]
    seen = set()
    for (W, H, sr, dr, kern) in cases:
        for ang in (0, 90, 180, 270):
            iso = ((W - 1) / 2, (H - 1) / 2) if ang in (0, 180) else (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
            src = rng.random((H, W)).astype(np.float32)
            for mode, omode in ((1, po.MODE_EXACT), (2, po.MODE_FAST)):
                gold = po.oracle_run(omode, src.astype(np.float64), sr, dr, iso, ang)
                dst, giso, lay = _host(gpu, src, dict(src_res=float(sr), dst_res=float(dr), iso=iso, angle=float(ang)), mode)
                assert dst.shape == gold.dst.shape and tuple(giso) == gold.dst_iso
                seen.add(gpu.last_kernel())
                if ang in (0, 180):
                    assert lay.kernel == kern, (W, H, sr, dr, ang, lay.kernel)
                if dst.size:
                    assert rel_err(dst, gold.dst).max() <= TOL, (W, H, sr, dr, ang, mode, gpu.last_kernel())
                assert np.array_equal(gold.dst == 0, dst == 0), (W, H, sr, dr, ang, mode)
    # the transposed quadrants at ratios below 2 go through the LDS-tile kernel
    assert {"aai_axis_kernel", "aai_axis_wide_kernel", "aai_axis_tile_kernel"} <= seen, seen


def test_wide_footprints_against_oracle(gpu, formulation, po):
    """Large footprints (heavy down-sampling at an angle): aai_wide_kernel -- the fp32 formulation over a window split into
    2 x 2 or 4 x 4 parts, a lane per part -- up to windows of 32 x 32 source pixels, aai_rotated_runs_kernel (double precision:
    boundary / interior / boundary runs per source row) beyond that, within 0.006 degrees of an axis and under
    AAI_POLICY_DOUBLE_PRECISION.  Same cases as the CPU replay test, plus 8- and 16-bit sources."""
    from area_average_interpolation_amd import _lib as L
    rng = np.random.default_rng(78)
    kernels = set()
    for k, (W, H, sr, dr, ang, off) in enumerate(RUNS_CASES):
        iso = ((W - 1) / 2 + off[0], (H - 1) / 2 + off[1])
        src = rng.random((H, W)).astype(np.float32)
        for policy in (0, 1):
            gold = po.oracle_run(po.MODE_EXACT, src.astype(np.float64), sr, dr, iso, ang, policy=policy)
            rc, msg, dst, giso, lay = gpu.resample_host(src, sr, dr, iso, ang, mode=1, policy=policy)
            assert rc == 0, msg
            # (footprints whose window still fits the fp32 formulations -- up to about 5.5 : 1 -- take the cell kernel)
            assert any(n in gpu.last_kernel() for n in ("aai_rotated_runs_kernel", "aai_wide_kernel", "aai_cell_kernel", "aai_quad_kernel")), (k, gpu.last_kernel())
            kernels.add(gpu.last_kernel().split("<")[0])
            assert dst.shape == gold.dst.shape and tuple(giso) == gold.dst_iso
            assert rel_err(dst, gold.dst).max() <= TOL, (k, policy, W, H, sr, ang)
            assert np.array_equal(gold.dst == 0, dst == 0), (k, policy)
        if k % 3 == 0:
            for dt, top in ((np.uint8, 256), (np.uint16, 65536)):
                isrc = rng.integers(0, top, size=(H, W)).astype(dt)
                gold = po.oracle_run(po.MODE_EXACT, isrc.astype(np.float64), sr, dr, iso, ang)
                rc, msg, dst, giso, lay = gpu.resample_host(isrc, sr, dr, iso, ang, mode=1)
                assert rc == 0, msg
                assert (np.abs(dst - gold.dst) / np.maximum(np.abs(gold.dst), 1e-3 * top)).max() <= TOL, (k, dt)
                assert np.array_equal(gold.dst == 0, dst == 0), (k, dt)
                # fast mode over the same footprint and source type (aai_wide_fast_kernel / aai_quad_fast_kernel / the line-walking kernel)
                gold = po.oracle_run(po.MODE_FAST, isrc.astype(np.float64), sr, dr, iso, ang)
                rc, msg, dst, giso, lay = gpu.resample_host(isrc, sr, dr, iso, ang, mode=2)
                assert rc == 0, msg
                kernels.add(gpu.last_kernel().split("<")[0])
                assert (np.abs(dst - gold.dst) / np.maximum(np.abs(gold.dst), 1e-3 * top)).max() <= TOL, (k, dt, "fast", gpu.last_kernel())
                assert np.array_equal(gold.dst == 0, dst == 0), (k, dt, "fast")
    assert {"aai_wide_kernel", "aai_rotated_runs_kernel", "aai_wide_fast_kernel"} <= kernels, kernels
    # the same request under the double-precision policy stays on the runs kernel and agrees to fp32 rounding
    W, H, sr, dr, ang, off = RUNS_CASES[2]
    iso = ((W - 1) / 2, (H - 1) / 2)
    src = rng.random((H, W)).astype(np.float32)
    gold = po.oracle_run(po.MODE_EXACT, src.astype(np.float64), sr, dr, iso, ang).dst
    rc, msg, dst, _, _ = gpu.resample_host(src, sr, dr, iso, ang, mode=1, policy=gpu.POLICY_DOUBLE_PRECISION)
    assert rc == 0 and "aai_rotated_runs_kernel" in gpu.last_kernel() and rel_err(dst, gold).max() <= 2e-7, (msg, gpu.last_kernel())


def test_near_axis_rotations_against_oracle(gpu, formulation, po):
    """Rotations a hair away from a multiple of 90 degrees: the rotated kernels run with sin or cos around 1e-9
    (1/sin ~ 1e9 inside the closed forms), or the request snaps to the axis-aligned kernel (|tan| < DBL_EPSILON)."""
    rng = np.random.default_rng(3)
    for ang in (1e-13, 1e-10, 1e-7, 1e-4, 0.01, 89.99, 89.9999999, 90 - 1e-10, 90 + 1e-9, 179.9999, 359.999999):
        for (sr, dr) in ((3, 1), (1, 1), (2.5, 1)):
            W, H, iso = 37, 29, (17.3, 12.1)
            src = rng.random((H, W)).astype(np.float32)
            for mode, omode in ((1, po.MODE_EXACT), (2, po.MODE_FAST)):
                gold = po.oracle_run(omode, src.astype(np.float64), sr, dr, iso, ang).dst
                dst, _, lay = _host(gpu, src, dict(src_res=float(sr), dst_res=float(dr), iso=iso, angle=float(ang)), mode)
                assert dst.shape == gold.shape
                assert rel_err(dst, gold).max() <= TOL, (ang, sr, dr, mode, lay.kernel)
                assert np.array_equal(gold == 0, dst == 0), (ang, sr, dr, mode)


def test_typed_sources_u8_u16_against_oracle(gpu, formulation, po):
    """8-bit and 16-bit unsigned sources through aai_resample_host /
    aai_resample_batch_device; pixel values are used as they are, output fp32."""
    import torch
    from area_average_interpolation_amd import _lib as L
    rng = np.random.default_rng(23)
    geoms = [(257, 130, 4, 1, 0.0, 1), (300, 64, 2, 1, 0.0, 1), (131, 97, 3, 1, 90.0, 1), (90, 70, 3, 1, 17.5, 1),
             (90, 70, 3, 1, 17.5, 2), (40, 30, 1, 2, 45.0, 1), (3000, 650, 600, 1, 0.0, 1), (64, 64, 4, 1, 180.0, 2)]
    for (W, H, sr, dr, ang, mode) in geoms:
        iso = ((W - 1) / 2, (H - 1) / 2)
        for npdt, code, hi in ((np.uint8, L.DTYPE_U8, 256), (np.uint16, L.DTYPE_U16, 65536)):
            src = rng.integers(0, hi, size=(H, W)).astype(npdt)
            gold = po.oracle_run(po.MODE_EXACT if mode == 1 else po.MODE_FAST, src.astype(np.float64), sr, dr, iso, ang).dst
            dst, giso, lay = _host(gpu, src, dict(src_res=float(sr), dst_res=float(dr), iso=iso, angle=float(ang)), mode)
            assert dst.dtype == np.float32 and dst.shape == gold.shape
            assert (np.abs(dst - gold) / np.maximum(np.abs(gold), 1e-3 * hi)).max() <= TOL, (W, H, sr, dr, ang, mode, npdt)
            assert np.array_equal(gold == 0, dst == 0)
            # device-resident typed batch of two images, with a padded source stride
            pad = 5
            t = torch.zeros((2, H, W + pad), dtype=torch.uint8 if npdt == np.uint8 else torch.int16, device="cuda")
            view = src if npdt == np.uint8 else src.view(np.int16)
            t[0, :, :W] = torch.from_numpy(view.copy()).cuda()
            t[1, :, :W] = torch.from_numpy(view[::-1].copy()).cuda()
            out = torch.empty((2, lay.dst_height, lay.dst_width), dtype=torch.float32, device="cuda")
            rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=mode)
            gpu.resample_device(rq, t.data_ptr(), W + pad, out.data_ptr(), lay.dst_width, torch.cuda.current_stream().cuda_stream,
                                batch=2, src_image_stride=H * (W + pad), dst_image_stride=lay.dst_width * lay.dst_height, src_dtype=code)
            torch.cuda.synchronize()
            assert np.array_equal(out[0].cpu().numpy(), dst)
            gold2 = po.oracle_run(po.MODE_EXACT if mode == 1 else po.MODE_FAST, src[::-1].astype(np.float64), sr, dr, iso, ang).dst
            assert (np.abs(out[1].cpu().numpy() - gold2) / np.maximum(np.abs(gold2), 1e-3 * hi)).max() <= TOL


def test_exact_policy_against_oracle(gpu, formulation, po):
    rng = np.random.default_rng(12)
    from area_average_interpolation_amd import _lib as L
    for k in range(12):
        W, H = int(rng.integers(8, 60)), int(rng.integers(8, 60))
        sr, dr, ang = float(rng.uniform(1, 5)), 1.0, float(rng.uniform(1, 89)) + 90 * (k % 4)
        iso = ((W - 1) / 2, (H - 1) / 2)
        src = rng.random((H, W)).astype(np.float32)
        gold = po.oracle_run(po.MODE_EXACT, src.astype(np.float64), sr, dr, iso, ang, policy=po.POLICY_EXACT).dst
        dst, _, _ = _host(gpu, src, dict(src_res=sr, dst_res=dr, iso=iso, angle=ang), 1, policy=L.POLICY_EXACT)
        assert rel_err(dst, gold).max() <= TOL, (k, ang)


def _clip_polygon(poly, x0, x1, y0, y1):
    """Sutherland-Hodgman: convex polygon (list of (x, y)) clipped to the axis-aligned rectangle [x0, x1] x [y0, y1]."""
    def clip(pts, inside, cross):
        out = []
        for i, p in enumerate(pts):
            q = pts[(i + 1) % len(pts)]
            if inside(p):
                out.append(p)
                if not inside(q):
                    out.append(cross(p, q))
            elif inside(q):
                out.append(cross(p, q))
        return out
    def at_x(x):
        return lambda p, q: (x, p[1] + (q[1] - p[1]) * (x - p[0]) / (q[0] - p[0]))
    def at_y(y):
        return lambda p, q: (p[0] + (q[0] - p[0]) * (y - p[1]) / (q[1] - p[1]), y)
    for inside, cross in ((lambda p: p[0] >= x0, at_x(x0)), (lambda p: p[0] <= x1, at_x(x1)),
                          (lambda p: p[1] >= y0, at_y(y0)), (lambda p: p[1] <= y1, at_y(y1))):
        poly = clip(poly, inside, cross)
        if len(poly) < 3:
            return []
    return poly


def _polygon_area(poly):
    return 0.5 * abs(sum(p[0] * q[1] - q[0] * p[1] for p, q in zip(poly, poly[1:] + poly[:1])))


def test_exact_policy_against_polygon_clipping(gpu, formulation):
    """AAI_POLICY_EXACT pinned by code that shares nothing with the product or the oracle: every dst pixel is the square
    spanned by the steps between neighbouring dst centres (the reference's affine map restated in conftest.sample_points),
    clipped against every source pixel of the ORIGINAL image with a Sutherland-Hodgman clip written here, areas by the
    shoelace formula, value = sum(area x pixel) / sum(area) over the pixels inside the image (Source.cpp:426-429, 577).
    20 geometries, all four quadrants, down- and up-sampling (replicated source pixels), off-centre isocenters."""
    import math
    import torch
    rng = np.random.default_rng(21)
    worst = 0.0
    for k in range(20):
        W, H = int(rng.integers(7, 22)), int(rng.integers(7, 22))
        sr = float(rng.uniform(1.0, 4.0))
        dr = 1.0 if k % 5 else float(rng.uniform(1.2, 2.6)) * sr             # every fifth case up-samples
        ang = float(rng.uniform(2.0, 88.0)) + 90.0 * (k % 4)
        iso = (float(rng.uniform(0, W - 1)), float(rng.uniform(0, H - 1))) if k % 3 else ((W - 1) / 2, (H - 1) / 2)
        src = rng.random((H, W)).astype(np.float32)
        rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=1, policy=gpu.POLICY_EXACT)
        rc, msg, lay = gpu.query(rq)
        assert rc == 0, msg
        rc, msg, dst, _, _ = gpu.resample_host(src, sr, dr, iso, ang, mode=1, policy=gpu.POLICY_EXACT)
        assert rc == 0, msg
        dH, dW = lay.dst_height, lay.dst_width
        sx, sy = sample_points(rq, lay, list(range(dH + 1)), "cpu")
        sx, sy = sx.numpy(), sy.numpy()
        # affine map: the steps between neighbouring centres are the square's edge vectors
        eax, eay = sx[0, 1] - sx[0, 0], sy[0, 1] - sy[0, 0]
        ebx, eby = sx[1, 0] - sx[0, 0], sy[1, 0] - sy[0, 0]
        gold = np.zeros((dH, dW))
        for y in range(dH):
            for x in range(dW):
                cx, cy = sx[y, x], sy[y, x]
                sq = [(cx + 0.5 * (sa * eax + sb * ebx), cy + 0.5 * (sa * eay + sb * eby)) for sa, sb in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
                xs, ys = [p[0] for p in sq], [p[1] for p in sq]
                sa_, sva = 0.0, 0.0
                for j in range(max(0, math.floor(min(ys) + 0.5)), min(H - 1, math.ceil(max(ys) - 0.5)) + 1):
                    for i in range(max(0, math.floor(min(xs) + 0.5)), min(W - 1, math.ceil(max(xs) - 0.5)) + 1):
                        a = _polygon_area(_clip_polygon(sq, i - 0.5, i + 0.5, j - 0.5, j + 0.5) or [(0, 0)] * 3)
                        sa_ += a
                        sva += a * float(src[j, i])
                gold[y, x] = sva / sa_ if sa_ > 1e-12 else 0.0
        # pixels that only graze the image (area below 1e-9 of a source pixel) are decided by rounding on either side
        err = rel_err(dst, gold)
        graze = (np.abs(gold) == 0) != (dst == 0)
        assert graze.sum() <= 2, (k, int(graze.sum()))
        worst = max(worst, float(err[~graze].max()))
        assert err[~graze].max() <= TOL, (k, W, H, sr, dr, ang, float(err[~graze].max()))
    assert worst > 0.0


def test_comparison_samplers_against_cpu_restatement(gpu, po):
    """Bilinear / bicubic are build-defined (the reference has none): parity is against our own CPU
    restatement only ('parity unpinned')."""
    import ctypes as C
    lib = po._load_oracle()
    rng = np.random.default_rng(13)
    for k in range(16):
        W, H = int(rng.integers(4, 50)), int(rng.integers(4, 50))
        sr, dr = [(1, 4), (1, 2), (1, 1), (3, 1)][k % 4]
        ang = [0, 45, 17.5, 200, 270, 300][k % 6]
        iso = ((W - 1) / 2, (H - 1) / 2)
        src = rng.random((H, W)).astype(np.float32)
        for mode in (3, 4):
            gold = po.oracle_run(mode, src.astype(np.float64), sr, dr, iso, ang).dst
            dst, _, _ = _host(gpu, src, dict(src_res=sr, dst_res=dr, iso=iso, angle=ang), mode)
            assert dst.shape == gold.shape
            # fp32 taps vs fp64 restatement; cubic weights reach 1.125, values in [0,1)
            assert np.abs(dst - gold).max() <= 2e-5, (k, mode, float(np.abs(dst - gold).max()))


def test_knife_edge_geometries_against_oracle(gpu, po):
    """Structured geometries (edges through pixel corners, vertices on pixel sides): the production pass flags
    the waves concerned and the fix-up pass replays the reference's own arithmetic (csrc/aai_strict.hpp);
    every pixel must then match the reference restatement."""
    import math
    rng = np.random.default_rng(4)
    angs = [30, 45, 60, math.degrees(math.atan(0.5)), math.degrees(math.atan(0.75)), 22.5, 135, 210, 315]
    ratios = [(2, 1), (3, 1), (4, 1), (1, 1), (1, 2), (2.8284271247461903, 1), (1.4142135623730951, 1)]
    runs = 0
    for ang in angs:
        for (sr, dr) in ratios:
            for kind in range(3):
                W, H = int(rng.integers(16, 36)), int(rng.integers(16, 36))
                if dr / sr > 1:
                    W, H = W // 3 + 4, H // 3 + 4
                iso = [((W - 1) / 2, (H - 1) / 2), (0.0, 0.0), (float(rng.integers(0, W)), float(rng.integers(0, H)) + 0.5)][kind]
                src = rng.random((H, W)).astype(np.float32)
                for mode, omode in ((1, po.MODE_EXACT), (2, po.MODE_FAST)):
                    gold = po.oracle_run(omode, src.astype(np.float64), sr, dr, iso, ang).dst
                    dst, _, _ = _host(gpu, src, dict(src_res=float(sr), dst_res=float(dr), iso=iso, angle=float(ang)), mode)
                    assert dst.shape == gold.shape
                    assert (rel_err(dst, gold) > TOL).sum() == 0, (W, H, sr, dr, iso, ang, mode)
                    assert np.array_equal(gold == 0, dst == 0), (W, H, sr, dr, iso, ang, mode)
                    runs += 1
    assert runs > 300


def test_knife_edge_geometries_against_reference_goldens(gpu, formulation, po, knife_golden):
    """The structured knife-edge geometries against outputs of the UNMODIFIED reference (tests/golden/knife_cases.npz,
    generated by tests/golden/make_golden.py knife from oracle/_ref): 189 geometries x both modes, no pixel excepted,
    exact zeros exact.  This is the reference-held evidence for the fix-up pass (Source.cpp:330-342, 401-408, 500-564, 1430)."""
    z, manifest = knife_golden
    for i, c in enumerate(manifest):
        src = po.synth_image(c["W"], c["H"], c["seed"])
        for mode, tag in ((1, "exact"), (2, "fast")):
            dst, iso, lay = _host(gpu, src, c, mode)
            gold = z["k%03d_%s" % (i, tag)]
            assert dst.shape == gold.shape and list(iso) == c["dst_iso"], (i, tag)
            assert (rel_err(dst, gold) > TOL).sum() == 0, (i, tag, c, gpu.last_kernel())
            assert np.array_equal(gold == 0, dst == 0), (i, tag, c)


def test_axis_aligned_knife_geometries_against_reference_goldens(gpu, po, axis_knife_golden):
    """Rotations 0/90/180/270 with edges on pixel boundaries / through pixel centres against outputs of the UNMODIFIED
    reference (tests/golden/axis_knife_cases.npz): K1 plus the fix-up pass over the dst pixels the plan's model scan
    flags (csrc/aai_axis_verify.hpp) -- no pixel excepted.  Also as 16-bit and as interleaved images (same plan family,
    fix-up with channels), in a batch with padded strides, and as row bands from footprint-only buffers."""
    import torch
    z, manifest = axis_knife_golden
    st = torch.cuda.current_stream().cuda_stream
    for i, c in enumerate(manifest):
        src = po.synth_image(c["W"], c["H"], c["seed"])
        fast, iso, lay = _host(gpu, src, c, 2)
        gold = z["a%03d_fast" % i]
        assert "axis" in gpu.last_kernel() and fast.shape == gold.shape
        assert rel_err(fast, gold).max() <= 1e-6 and np.array_equal(gold == 0, fast == 0), (i, c, float(rel_err(fast, gold).max()))
        dst, iso, lay = _host(gpu, src, c, 1)
        gold = z["a%03d_exact" % i]
        assert dst.shape == gold.shape and list(iso) == c["dst_iso"], i
        assert "axis" in gpu.last_kernel()
        assert rel_err(dst, gold).max() <= 1e-6 and np.array_equal(gold == 0, dst == 0), (i, c, float(rel_err(dst, gold).max()))
        if i % 7:
            continue
        # interleaved: channel 1 = the image, channels 0 / 2 = something else
        inter = np.stack([src[::-1, ::-1], src, 1.0 - src], axis=2).astype(np.float32)
        rc, msg, idst, ilay = gpu.resample_interleaved_host(inter, c["src_res"], c["dst_res"], tuple(c["iso"]), c["angle"], mode=1)
        assert rc == 0 and rel_err(idst[:, :, 1], gold).max() <= 1e-6, (i, c)
        # a batch of two with padded strides
        H, W = src.shape
        rq = gpu.make_request(W, H, c["src_res"], c["dst_res"], tuple(c["iso"]), c["angle"], mode=1)
        dH, dW = gold.shape
        bsrc = torch.zeros((2, H + 1, W + 5), dtype=torch.float32, device="cuda")
        bsrc[0, :H, :W] = torch.from_numpy(src).cuda()
        bsrc[1, :H, :W] = torch.from_numpy(src[::-1].copy()).cuda()
        bdst = torch.full((2, dH + 2, dW + 3), -1.0, dtype=torch.float32, device="cuda")
        gpu.resample_device(rq, bsrc.data_ptr(), W + 5, bdst.data_ptr(), dW + 3, st, batch=2,
                            src_image_stride=(H + 1) * (W + 5), dst_image_stride=(dH + 2) * (dW + 3))
        torch.cuda.synchronize()
        assert np.array_equal(bdst[0, :dH, :dW].cpu().numpy(), dst)
        # row bands from buffers that hold only their footprint
        if dH >= 2:
            r0 = dH // 2
            for (b0, b1) in ((0, r0), (r0, dH)):
                a, b = gpu.band_source_rows(rq, b0, b1)
                band_src = torch.from_numpy(src[a:b].copy()).cuda()
                band_dst = torch.empty((b1 - b0, dW), dtype=torch.float32, device="cuda")
                gpu.resample_band_device(rq, b0, b1, band_src.data_ptr(), W, band_dst.data_ptr(), dW, st)
                torch.cuda.synchronize()
                assert np.array_equal(band_dst.cpu().numpy(), dst[b0:b1]), (i, c, b0, b1, a, b)
    # 16-bit sources through the same plans
    for i in (0, 61, 122, 183, 300, 500):
        c = manifest[i]
        src16 = (po.synth_image(c["W"], c["H"], c["seed"]) * 65535).astype(np.uint16)
        gold = po.oracle_run(po.MODE_EXACT, src16.astype(np.float64), c["src_res"], c["dst_res"], tuple(c["iso"]), c["angle"]).dst
        rc, msg, dst, _, lay = gpu.resample_host(src16, c["src_res"], c["dst_res"], tuple(c["iso"]), c["angle"], mode=1)
        assert rc == 0 and rel_err(dst, gold, floor=65.5).max() <= 1e-6, (i, c)


# ---- (c) properties at full BASELINE sizes -------------------------------------------------------------------
def _device_run(gpu, rq, src, batch=None):
    import torch
    rc, msg, lay = gpu.query(rq)
    assert rc == 0, msg
    shape = (lay.dst_height, lay.dst_width) if batch is None else (batch, lay.dst_height, lay.dst_width)
    dst = torch.empty(shape, dtype=torch.float32, device="cuda")
    W = rq.src_width
    if batch is None:
        gpu.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), lay.dst_width, torch.cuda.current_stream().cuda_stream)
    else:
        gpu.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), lay.dst_width, torch.cuda.current_stream().cuda_stream,
                            batch=batch, src_image_stride=rq.src_width * rq.src_height,
                            dst_image_stride=lay.dst_width * lay.dst_height)
    torch.cuda.synchronize()
    return dst


@pytest.mark.parametrize("cfg", [(8192, 8192, 4.0, 1.0, 0.0), (8192, 8192, 8192.0, 2731.0, 17.5),
                                 (8192, 8192, 8.0, 1.0, 107.5), (8192, 8192, 4.0, 1.0, 270.0)])      # + rows-as-runs kernel, transposed K1
def test_full_size_constant_and_linearity(gpu, cfg):
    """Constant image -> the same constant wherever the dst pixel touches the image, exact 0 elsewhere
    (Source.cpp:577); and resample(a*x + b*y) == a*resample(x) + b*resample(y) (weights do not depend on data)."""
    import torch
    W, H, sr, dr, ang = cfg
    rq = gpu.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang)
    const = torch.full((H, W), 0.7310586, dtype=torch.float32, device="cuda")
    out = _device_run(gpu, rq, const)
    nz = out[out != 0]
    assert nz.numel() > 0.6 * out.numel()
    assert float((nz - 0.7310586).abs().max()) <= 2e-6
    x = torch.empty((H, W), dtype=torch.float32, device="cuda")
    y = torch.empty((H, W), dtype=torch.float32, device="cuda")
    gpu.synth_device(x.data_ptr(), W, H, W, 21)
    gpu.synth_device(y.data_ptr(), W, H, W, 22)
    lhs = _device_run(gpu, rq, 0.25 * x + 3.0 * y)
    rhs = 0.25 * _device_run(gpu, rq, x) + 3.0 * _device_run(gpu, rq, y)
    assert float((lhs - rhs).abs().max()) <= 2e-5
    assert torch.equal(lhs == 0, rhs == 0)


def test_integer_ratio_is_a_box_mean(gpu):
    """theta = 0, integer ratio: output = mean of an aligned LxL block with the reference's
    isocenter-anchored phase (cfg2 covers source columns 4dx+2 .. 4dx+5)."""
    import torch
    W = H = 8192
    rq = gpu.make_request(W, H, 4, 1, ((W - 1) / 2, (H - 1) / 2), 0.0)
    x = torch.empty((H, W), dtype=torch.float32, device="cuda")
    gpu.synth_device(x.data_ptr(), W, H, W, 5)
    out = _device_run(gpu, rq, x)
    pad = torch.nn.functional.pad(x.double()[2:, 2:], (0, 2, 0, 2))            # shift by the phase, zero-fill
    box = torch.nn.functional.avg_pool2d(pad[None, None], 4)[0, 0]
    # interior: all 16 source pixels exist
    assert float((out[:-1, :-1].double() - box[:-1, :-1]).abs().max()) <= 1e-6
    # last row/column: only 2 of 4 source rows/columns exist and the mean renormalises (Source.cpp:577)
    assert float((out[-1, :-1].double() - box[-1, :-1] * 2).abs().max()) <= 1e-6
    assert float((out[-1, -1].double() - box[-1, -1] * 4).abs()) <= 1e-6


def test_batch_equals_singles_and_strides(gpu):
    """BASELINE config 4 shape: a batch launch is bit-identical to per-image launches; padded row strides work."""
    import torch
    W = H = 1024
    B = 5
    rq = gpu.make_request(W, H, 4, 1, ((W - 1) / 2, (H - 1) / 2), 0.0)
    src = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    for b in range(B):
        gpu.synth_device(src[b].data_ptr(), W, H, W, b + 1)
    batched = _device_run(gpu, rq, src, batch=B)
    for b in range(B):
        assert torch.equal(batched[b], _device_run(gpu, rq, src[b]))
    # strided source and destination
    rc, _, lay = gpu.query(rq)
    wide_src = torch.zeros((H, W + 24), dtype=torch.float32, device="cuda")
    wide_src[:, :W] = src[0]
    wide_dst = torch.full((lay.dst_height, lay.dst_width + 8), -7.0, dtype=torch.float32, device="cuda")
    gpu.resample_device(rq, wide_src.data_ptr(), W + 24, wide_dst.data_ptr(), lay.dst_width + 8, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(wide_dst[:, :lay.dst_width], batched[0])
    assert float(wide_dst[:, lay.dst_width:].max()) == -7.0            # padding untouched
    # rotated kernel, batch + stride
    rq2 = gpu.make_request(W, H, 3, 1, ((W - 1) / 2, (H - 1) / 2), 33.0)
    b2 = _device_run(gpu, rq2, src, batch=B)
    assert torch.equal(b2[3], _device_run(gpu, rq2, src[3]))
    # more images than one launch's grid.z can carry (65,535): the entry splits the batch
    big = 70000
    tiny = torch.rand((big, 8, 8), dtype=torch.float32, device="cuda")
    for ang in (0.0, 30.0):
        rq3 = gpu.make_request(8, 8, 2, 1, (3.5, 3.5), ang)
        b3 = _device_run(gpu, rq3, tiny, batch=big)
        for b in (0, 65534, 65535, 65536, big - 1):
            assert torch.equal(b3[b], _device_run(gpu, rq3, tiny[b])), (ang, b)


def test_config4_batch_of_64_equals_singles(gpu):
    """BASELINE config 4 at its full size: 64 independent 4096^2 -> 1024^2 images in ONE batched launch; sampled images
    are bit-identical to their single-image launches, image 0 (seed 1) matches the reference's known answers."""
    import torch
    W = H = 4096
    B = 64
    rq = gpu.make_request(W, H, 4, 1, ((W - 1) / 2, (H - 1) / 2), 0.0)
    src = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    for b in range(B):
        gpu.synth_device(src[b].data_ptr(), W, H, W, b + 1)
    batched = _device_run(gpu, rq, src, batch=B)
    assert batched.shape == (B, 1024, 1024)
    for b in (0, 31, 63):
        assert torch.equal(batched[b], _device_run(gpu, rq, src[b])), b
    z, meta = load_full("cfg4")
    m = meta["exact"]
    assert rel_err(batched[0][::m["step"], ::m["step"]].cpu().numpy(), z["exact_grid"]).max() <= TOL
    assert rel_err(batched[0][m["rows"], :].cpu().numpy(), z["exact_rows"]).max() <= TOL
    # every image is a different image (no aliasing of batch slots)
    sums = batched.double().sum(dim=(1, 2))
    assert torch.unique(sums).numel() == B


@pytest.mark.parametrize("cfg", [(4096, 4096, 1.0, 4.0, 45.0), (2048, 1536, 3.0, 2.0, 200.0)])
def test_samplers_at_full_size(gpu, cfg):
    """BASELINE config 5's comparison legs at 4096^2 -> 23170^2 (and a down-sampling case in another quadrant).  The
    reference implements neither sampler (README.md:8 only names them), so they are pinned to something this repository
    did not write: bilinear against torch.nn.functional.grid_sample(mode='bilinear', padding_mode='border',
    align_corners=False) -- exactly clamp-to-edge bilinear -- at the same sample points, in float64, <= 1e-5; both
    samplers: constant -> constant inside the image extent, exact 0 outside, linear in the image."""
    import torch
    W, H, sr, dr, ang = cfg
    iso = ((W - 1) / 2, (H - 1) / 2)
    x = torch.empty((H, W), dtype=torch.float32, device="cuda")
    y = torch.empty((H, W), dtype=torch.float32, device="cuda")
    gpu.synth_device(x.data_ptr(), W, H, W, 31)
    gpu.synth_device(y.data_ptr(), W, H, W, 32)
    const = torch.full((H, W), 0.7310586, dtype=torch.float32, device="cuda")
    for mode in (3, 4):
        rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=mode)
        rc, msg, lay = gpu.query(rq)
        assert rc == 0, msg
        dH, dW = lay.dst_height, lay.dst_width
        if cfg[0] == 4096:
            assert (dH, dW) == (23170, 23170)
        out = _device_run(gpu, rq, x)
        bands = [range(0, 192), range(dH // 2 - 96, dH // 2 + 96), range(dH - 192, dH), range(dH // 3, dH // 3 + 64)]
        inside_total = 0
        for rows in bands:
            rows = list(rows)
            sx, sy = sample_points(rq, lay, rows, "cuda")
            inside = (sx >= -0.5 - 1e-9) & (sx <= W - 0.5 + 1e-9) & (sy >= -0.5 - 1e-9) & (sy <= H - 0.5 + 1e-9)      # the samplers' extent guard
            got = out[rows, :].double()
            assert float(got[~inside].abs().max()) == 0.0 if (~inside).any() else True          # exact 0 outside the image extent
            inside_total += int(inside.sum())
            if mode == 3:
                grid = torch.stack(((2 * sx + 1) / W - 1, (2 * sy + 1) / H - 1), dim=-1)[None]
                ref = torch.nn.functional.grid_sample(x.double()[None, None], grid, mode="bilinear", padding_mode="border", align_corners=False)[0, 0]
                assert float((got - ref)[inside].abs().max()) <= 1e-5, (cfg, rows[0])
            else:
                # Bicubic, pinned independently: Keys' cubic convolution kernel (a = -0.5) as published -- piecewise in the
                # DISTANCE d from the sample point: (a + 2) d^3 - (a + 3) d^2 + 1 for d <= 1, a d^3 - 5a d^2 + 8a d - 4a for
                # 1 < d < 2 -- evaluated in float64 torch at the 4 x 4 clamp-to-edge taps around the same sample points
                def keys_kernel(d, a=-0.5):
                    d = d.abs()
                    near = ((a + 2.0) * d - (a + 3.0)) * d * d + 1.0
                    far = ((a * d - 5.0 * a) * d + 8.0 * a) * d - 4.0 * a
                    return torch.where(d <= 1.0, near, torch.where(d < 2.0, far, torch.zeros_like(d)))
                x64 = x.double()
                fx, fy = torch.floor(sx), torch.floor(sy)
                ref = torch.zeros_like(sx)
                for ky in range(-1, 3):
                    wy = keys_kernel(sy - (fy + ky))
                    iy = (fy + ky).clamp(0, H - 1).long()
                    for kx in range(-1, 3):
                        ix = (fx + kx).clamp(0, W - 1).long()
                        ref += wy * keys_kernel(sx - (fx + kx)) * x64[iy, ix]
                assert float((got - ref)[inside].abs().max()) <= 1e-5, (cfg, rows[0], float((got - ref)[inside].abs().max()))
                del x64, fx, fy, ref
            del sx, sy, inside, got
        assert inside_total > 0
        # constant -> constant wherever the sample point lies inside the image extent, exact 0 elsewhere
        oc = _device_run(gpu, rq, const)
        nz = oc[oc != 0]
        assert nz.numel() > 0.4 * oc.numel() and float((nz - 0.7310586).abs().max()) <= 2e-6
        assert torch.equal(oc == 0, out == 0) or float(((oc == 0) != (out == 0)).sum()) <= 1e-6 * oc.numel()      # a sample may be exactly 0
        del oc, nz
        # linear in the image
        oy = _device_run(gpu, rq, y)
        lhs = _device_run(gpu, rq, 0.25 * x + 3.0 * y)
        lhs -= 0.25 * out
        lhs -= 3.0 * oy
        assert float(lhs.abs().max()) <= 2e-5
        del out, oy, lhs
        torch.cuda.empty_cache()


def test_pipelined_host_batch_equals_single_calls(gpu):
    """aai_resample_batch_host (three device slots, one stream each) from pageable and from page-locked
    buffers, all source types, more images than slots: bit-identical to one aai_resample_host call per image."""
    rng = np.random.default_rng(31)
    for (W, H, sr, dr, ang, mode, dt) in [(301, 211, 4.0, 1.0, 0.0, 1, np.float32), (128, 96, 3.0, 1.0, 17.5, 1, np.uint8),
                                          (96, 128, 7.0, 1.0, 200.0, 1, np.uint16), (150, 90, 2.0, 1.0, 33.0, 2, np.float32),
                                          (64, 64, 1.0, 2.0, 45.0, 3, np.float32)]:
        B = 7
        iso = ((W - 1) / 2, (H - 1) / 2)
        if dt == np.float32:
            srcs = rng.random((B, H, W)).astype(np.float32)
        else:
            srcs = rng.integers(0, np.iinfo(dt).max + 1, size=(B, H, W)).astype(dt)
        singles = [gpu.resample_host(srcs[b], sr, dr, iso, ang, mode=mode)[2] for b in range(B)]
        rc, msg, dst, lay = gpu.resample_batch_host(srcs, sr, dr, iso, ang, mode=mode)
        assert rc == 0, msg
        assert dst.shape == (B,) + singles[0].shape
        for b in range(B):
            assert np.array_equal(dst[b], singles[b]), (W, H, ang, mode, dt, b)
        with gpu.PinnedArray(srcs.shape, srcs.dtype) as ps, gpu.PinnedArray(dst.shape, np.float32) as pd:
            ps.array[...] = srcs
            pd.array[...] = -1.0
            rc, msg, dst2, lay = gpu.resample_batch_host(ps.array, sr, dr, iso, ang, mode=mode, out=pd.array)
            assert rc == 0, msg
            assert np.array_equal(dst2, dst)
    # argument errors come before any device work; an empty batch is a no-op
    rc, msg, dst, lay = gpu.resample_batch_host(np.zeros((0, 8, 8), np.float32), 2.0, 1.0, (3.5, 3.5), 0.0)
    assert rc == 0 and dst.shape[0] == 0
    rc, msg, dst, lay = gpu.resample_batch_host(np.zeros((2, 8, 8), np.float32), 2.0, 0.0, (3.5, 3.5), 0.0)
    assert rc != 0 and "resolution" in msg


def test_interleaved_channels_equal_planar_calls(gpu, po):
    """interleaved [H, W, C] images.  Every channel must equal the single-channel call on that
    channel alone, bit for bit -- the axis-aligned kernel in all four quadrants and all its output paths, the rotated
    area / fast kernels (incl. a knife-edge angle and the large-footprint regime), the samplers, 8-/16-bit sources --
    and one case is checked against the oracle directly."""
    rng = np.random.default_rng(41)
    kernels = set()
    cases = [  # W, H, srcRes, dstRes, angle, mode
        (517, 40, 4, 1, 0.0, 1), (517, 40, 4, 1, 180.0, 1), (300, 33, 3, 1, 90.0, 1), (301, 21, 2, 1, 270.0, 2),
        (259, 17, 1, 1, 0.0, 1), (70, 50, 1, 2, 90.0, 1), (40, 30, 1, 4, 180.0, 1), (1030, 9, 8, 1, 0.0, 2),
        (3000, 800, 700, 1, 0.0, 1), (3, 50, 2, 1, 0.0, 1), (5, 700, 3, 1, 270.0, 1),
        (128, 96, 3, 1, 17.5, 1), (96, 128, 3, 1, 200.0, 2), (64, 64, 2, 1, 45.0, 1), (120, 90, 8, 1, 33.3, 1),
        (96, 96, 6, 1, 107.5, 2), (64, 48, 1, 3, 30.0, 1), (80, 60, 2, 1, 17.5, 3), (80, 60, 1, 2, 300.0, 4),
    ]
    for k, (W, H, sr, dr, ang, mode) in enumerate(cases):
        C = 1 + k % 4
        dt = (np.float32, np.uint8, np.uint16)[k % 3] if mode in (1, 2) else np.float32
        iso = ((W - 1) / 2, (H - 1) / 2) if k % 2 else (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
        if dt == np.float32:
            src = rng.random((H, W, C)).astype(np.float32)
        else:
            src = rng.integers(0, np.iinfo(dt).max + 1, size=(H, W, C)).astype(dt)
        rc, msg, dst, lay = gpu.resample_interleaved_host(src, sr, dr, iso, ang, mode=mode)
        assert rc == 0, (k, msg)
        assert dst.shape == (lay.dst_height, lay.dst_width, C)
        kern_interleaved = gpu.last_kernel()
        kernels.add(kern_interleaved)
        for c in range(C):
            rc, msg, planar, giso, _ = gpu.resample_host(np.ascontiguousarray(src[:, :, c]), sr, dr, iso, ang, mode=mode)
            assert rc == 0, msg
            if W * C >= 4 > W:
                # the planar image is too narrow for the strip kernel (per-pixel fallback), the interleaved one is not:
                # same weights, different summation order
                assert rel_err(dst[:, :, c], planar, floor=1e-3 * float(src.max())).max() <= 1e-6, (k, c)
            elif any(t in gpu.last_kernel() for t in ("aai_quad", "aai_cell", "aai_wide")) != any(t in kern_interleaved for t in ("aai_quad", "aai_cell", "aai_wide")):
                # one of the two calls takes the fp32 quad formulation, the other a double-precision kernel (an interleaved
                # window too large to stage, or a wide footprint: areas shared between the channels in fp64): same areas to ~1e-7
                assert rel_err(dst[:, :, c], planar, floor=1e-3 * float(src.max())).max() <= 2e-6, (k, c)
                assert np.array_equal(dst[:, :, c] == 0, planar == 0), (k, c)
            else:
                assert np.array_equal(dst[:, :, c], planar), (k, W, H, sr, dr, ang, mode, C, c, dt, gpu.last_kernel())
    assert any("aai_quad_multi_kernel" in k for k in kernels), kernels
    # against the oracle itself
    src = rng.random((90, 120, 3)).astype(np.float32)
    rc, msg, dst, lay = gpu.resample_interleaved_host(src, 3.0, 1.0, (59.5, 44.5), 17.5)
    assert rc == 0, msg
    for c in range(3):
        gold = po.oracle_run(po.MODE_EXACT, src[:, :, c].astype(np.float64), 3.0, 1.0, (59.5, 44.5), 17.5)
        assert rel_err(dst[:, :, c], gold.dst).max() <= TOL and np.array_equal(gold.dst == 0, dst[:, :, c] == 0)
    # device entry: batch of interleaved images with padded strides
    import torch
    B, W, H, C = 3, 200, 150, 3
    rq = gpu.make_request(W, H, 4, 1, ((W - 1) / 2, (H - 1) / 2), 33.0)
    rc, msg, lay = gpu.query(rq)
    hsrc = rng.random((B, H, W + 5, C)).astype(np.float32)
    dsrc = torch.from_numpy(hsrc).cuda()
    ddst = torch.full((B, lay.dst_height, lay.dst_width + 2, C), -5.0, dtype=torch.float32, device="cuda")
    gpu.resample_interleaved_device(rq, C, dsrc.data_ptr(), (W + 5) * C, ddst.data_ptr(), (lay.dst_width + 2) * C,
                                    torch.cuda.current_stream().cuda_stream, batch=B, src_image_stride=H * (W + 5) * C,
                                    dst_image_stride=lay.dst_height * (lay.dst_width + 2) * C)
    torch.cuda.synchronize()
    got = ddst.cpu().numpy()
    for b in range(B):
        rc, msg, ref, _ = gpu.resample_interleaved_host(np.ascontiguousarray(hsrc[b, :, :W, :]), 4, 1, ((W - 1) / 2, (H - 1) / 2), 33.0)
        assert np.array_equal(got[b, :, :lay.dst_width, :], ref), b
    assert float(got[:, :, lay.dst_width:, :].max()) == -5.0          # padding untouched
    with pytest.raises(gpu.AaiError):
        gpu.resample_interleaved_device(rq, 5, dsrc.data_ptr(), (W + 5) * C, ddst.data_ptr(), (lay.dst_width + 2) * C)


def test_row_bands_equal_full_image(gpu, formulation):
    """dst row bands computed from buffers holding only their source footprint are
    bit-identical to the same rows of the full-image call -- every kernel family, all quadrants."""
    import torch
    from area_average_interpolation_amd.distributed import shard_rows
    rng = np.random.default_rng(29)
    cases = [(1024, 768, 4, 1, 0.0, 1), (1024, 768, 4, 1, 180.0, 1), (700, 900, 3, 1, 90.0, 1), (700, 900, 3, 1, 270.0, 2),
             (640, 480, 3, 1, 17.5, 1), (640, 480, 3, 1, 200.0, 2), (200, 160, 1, 3, 45.0, 1), (300, 200, 1, 2, 30.0, 4),
             (300, 200, 2, 1, 0.0, 3), (512, 512, 8192, 2731, 0.0, 1),
             # large footprints: rows-as-runs area kernel and the line-walking fast kernel, in every quadrant
             (900, 700, 6, 1, 17.5, 1), (900, 700, 6, 1, 107.5, 1), (700, 900, 8, 1, 200.0, 1), (700, 900, 7, 1, 300.0, 1),
             (900, 700, 6, 1, 107.5, 2), (700, 900, 8, 1, 300.0, 2)]
    for (W, H, sr, dr, ang, mode) in cases:
        rq = gpu.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang, mode=mode)
        rc, msg, lay = gpu.query(rq)
        assert rc == 0, msg
        src = torch.from_numpy(rng.random((H, W)).astype(np.float32)).cuda()
        full = torch.empty((lay.dst_height, lay.dst_width), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        gpu.resample_device(rq, src.data_ptr(), W, full.data_ptr(), lay.dst_width, st)
        world = 3
        covered = 0
        for rank in range(world):
            r0, r1 = shard_rows(lay.dst_height, rank, world)
            if r0 >= r1:
                continue
            a, b = gpu.band_source_rows(rq, r0, r1)
            assert 0 <= a < b <= H
            band_src = src[a:b].clone()                       # ONLY the footprint rows live in this buffer
            band_dst = torch.full((r1 - r0, lay.dst_width), -3.0, dtype=torch.float32, device="cuda")
            gpu.resample_band_device(rq, r0, r1, band_src.data_ptr(), W, band_dst.data_ptr(), lay.dst_width, st)
            torch.cuda.synchronize()
            assert torch.equal(band_dst, full[r0:r1]), (W, H, sr, dr, ang, mode, rank, r0, r1, a, b)
            covered += r1 - r0
        assert covered == lay.dst_height
    # argument checks: unaligned band start on a rotated request, rows out of range
    rq = gpu.make_request(640, 480, 3, 1, (319.5, 239.5), 17.5)
    with pytest.raises(gpu.AaiError):
        gpu.band_source_rows(rq, 8, 40)
    with pytest.raises(gpu.AaiError):
        gpu.band_source_rows(rq, 0, 100000)


def test_device_synth_matches_appendix_c_generator(gpu, po):
    import torch
    t = torch.empty((300, 517), dtype=torch.float32, device="cuda")
    gpu.synth_device(t.data_ptr(), 517, 300, 517, 3)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), po.synth_image(517, 300, 3))


def test_errors_before_device_work(gpu):
    from area_average_interpolation_amd import _lib as L
    rc, msg, dst, iso, lay = gpu.resample_host(np.ones((4, 4), np.float32), (1, 2), 1, (0, 0), 0)
    assert rc == L.ERR_RESOLUTION_MISMATCH and msg == "Assumed X & Y resolution are same." and dst is None
    with pytest.raises(gpu.AaiError):
        gpu.resample_device(gpu.make_request(4, 4, 1, 1, (0, 0), 0), 0, 4, 0, 4)      # null pointers


def test_reference_style_class(gpu, po):
    a = gpu.AreaAverageInterpolation()
    src = po.synth_image(64, 48, 4).astype(np.float64)
    (ok, msg), dst, iso = a.areaAverageInterpolation(src, (150, 150), (25.4, 25.4), (31, 23), 1.5)     # Source.cpp:1530-1533 style
    gold = po.oracle_run(po.MODE_EXACT, src, 150, 25.4, (31, 23), 1.5)
    assert ok and msg == "" and tuple(iso) == gold.dst_iso
    assert rel_err(dst, gold.dst).max() <= TOL
    (ok, msg), dst, iso = a.fastAreaAverageInterpolation(src, (150, 150), (25.4, 25.4), (31, 23), 1.5)
    gold = po.oracle_run(po.MODE_FAST, src, 150, 25.4, (31, 23), 1.5)
    assert ok and rel_err(dst, gold.dst).max() <= TOL


def test_multi_device_entry_and_prepare(gpu):
    """aai_resample_batch_multi_device_f32: one batch spread over several shards (device, count, pointers, stream) of ONE
    process -- on this one-GPU box both shards live on device 0, on two streams -- bit-identical to the batched call.
    aai_prepare builds the plan up front, so that a captured stream only sees launches."""
    import torch
    # (the third geometry: K1 with a fix-up list behind it; the fourth: fast mode, fix-up beside the kernel on the side stream)
    # (the last two: a wide footprint through aai_wide_kernel / aai_wide_fast_kernel)
    for (W, H, sr, dr, ang, mode) in ((768, 768, 4, 1, 0.0, 1), (768, 768, 3, 1, 17.5, 1), (40, 9, 3, 1, 0.0, 1), (768, 768, 2, 1, 45.0, 2),
                                      (768, 768, 8, 1, 17.5, 1), (768, 768, 8, 1, 17.5, 2)):
        rq = gpu.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang, mode=mode)
        gpu.prepare(rq)
        rc, msg, lay = gpu.query(rq)
        dW, dH = lay.dst_width, lay.dst_height
        src = torch.empty((5, H, W), dtype=torch.float32, device="cuda")
        for b in range(5):
            gpu.synth_device(src[b].data_ptr(), W, H, W, b + 1)
        whole = _device_run(gpu, rq, src, batch=5)
        dst = torch.full((5, dH, dW), -1.0, dtype=torch.float32, device="cuda")
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        gpu.resample_multi_device(rq, [(0, 2, src[0].data_ptr(), dst[0].data_ptr(), s1.cuda_stream),
                                       (0, 3, src[2].data_ptr(), dst[2].data_ptr(), s2.cuda_stream)], W, W * H, dW, dW * dH)
        torch.cuda.synchronize()
        assert torch.equal(dst, whole), (sr, ang)
        # after aai_prepare a stream capture only records launches (no allocation, no blocking copy)
        g = torch.cuda.CUDAGraph()
        out = torch.full((dH, dW), -1.0, dtype=torch.float32, device="cuda")
        cs = torch.cuda.Stream()
        with torch.cuda.graph(g, stream=cs):
            gpu.resample_device(rq, src[1].data_ptr(), W, out.data_ptr(), dW, torch.cuda.current_stream().cuda_stream)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, whole[1]), (sr, ang)


def test_plans_are_built_outside_the_cache_lock(gpu):
    """Building a plan (tables, scans, K1's launch-shape measurement: ~10 ms for a config-2-sized geometry) blocks only the
    thread that asked for it: while one thread prepares a large axis-aligned request, another prepares and runs small
    rotated requests on its own stream and finishes long before.  Also: aai_plan_info describes what was built, a second
    geometry of the same class takes the launch shape from the cache, and aai_shutdown drops the plans."""
    import threading
    import time
    import torch
    big = gpu.make_request(8192, 8192, 4.0, 1.0, (4095.5, 4095.5), 0.0, mode=1)
    done = {}

    def slow():
        gpu.set_device(0)
        t0 = time.perf_counter()
        gpu.prepare(big)
        done["big"] = (t0, time.perf_counter())

    def quick():
        gpu.set_device(0)
        s = torch.cuda.Stream()
        src = torch.rand((96, 128), dtype=torch.float32, device="cuda")
        time.sleep(0.002)                       # let the other thread get going
        t0 = time.perf_counter()
        for k in range(4):
            rq = gpu.make_request(128, 96, 3.0, 1.0, (63.5, 47.5), 10.0 + 7.0 * k, mode=1)
            rc, msg, lay = gpu.query(rq)
            out = torch.empty((lay.dst_height, lay.dst_width), dtype=torch.float32, device="cuda")
            gpu.resample_device(rq, src.data_ptr(), 128, out.data_ptr(), lay.dst_width, s.cuda_stream)
        s.synchronize()
        done["small"] = (t0, time.perf_counter())

    gpu.shutdown()                              # cold: no plan, but possibly a cached launch shape from earlier tests
    torch.cuda.synchronize()
    a, b = threading.Thread(target=slow), threading.Thread(target=quick)
    a.start(); b.start(); a.join(); b.join()
    big_t, small_t = done["big"], done["small"]
    text = gpu.plan_shape(big)
    assert "kernel=" in text and "build_ms=" in text and "tune=" in text, text
    build_ms = float(text.split("build_ms=")[1].split()[0])
    if build_ms > 4.0 and small_t[0] < big_t[1]:
        # the two really overlapped: the quick thread must not have waited for the big plan to finish
        assert small_t[1] - small_t[0] < 0.75 * (big_t[1] - big_t[0]) or small_t[1] < big_t[1], (big_t, small_t, text)
    # same class, other geometry: the launch shape comes from the cache (or was never measured: autotune disabled)
    other = gpu.make_request(8192, 8064, 4.0, 1.0, (4095.5, 4031.5), 0.0, mode=1)
    gpu.prepare(other)
    assert "tune=cached" in gpu.plan_shape(other) or "tune=default" in gpu.plan_shape(other), gpu.plan_shape(other)
    gpu.shutdown()
    assert gpu.plan_shape(big) == ""


def test_pinned_array_views_outlive_close(gpu):
    """PinnedArray.close() drops the owner's reference only: a view held elsewhere keeps the page-locked block alive."""
    import gc
    p = gpu.PinnedArray((64, 64), np.float32)
    p.array[...] = 3.5
    view = p.array[10:20]
    p.close()
    gc.collect()
    assert p.array is None and float(view.sum()) == 3.5 * 10 * 64
    del view
    gc.collect()


def test_quad_kernel_equals_its_cpu_replay_bit_for_bit(gpu, hostemu, po):
    """The fp32 quad kernel spells out its fused multiply-adds and is compiled without contraction, and so is the CPU
    replay of the same header (tests/emulation/host_emulation.cpp): outside the pixels the scans hand to the
    double-precision pass both execute the same IEEE operations, so the GPU's output equals the replay's bit for bit --
    which makes the CPU suite's golden-vector checks of the replay checks of the device arithmetic itself."""
    rng = np.random.default_rng(77)
    hostemu.aai_emu_use_quad(1)
    try:
        for (W, H, sr, dr, ang, policy) in ((200, 160, 8192.0, 2731.0, 17.5, 0), (96, 96, 1.0, 4.0, 45.0, 0), (180, 140, 4.0, 1.0, 0.5, 0),
                                            (150, 150, 2.0, 1.0, 117.3, 1), (120, 90, 1.0, 1.0, 200.0, 0), (128, 128, 3.0, 2.0, 300.0, 0)):
            iso = (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
            src = rng.random((H, W)).astype(np.float32)
            # (small outputs -- these -- stay on the quad kernel in area mode; larger plain images take the cell kernel, next test)
            for mode, kernel in ((1, "aai_quad_kernel"), (2, "aai_quad_fast_kernel")):
                rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=mode, policy=policy)
                ref, axis = hostemu.resample(rq, src)
                quad, flagged = hostemu.quad_stats()
                rc, msg, dst, _, lay = gpu.resample_host(src, sr, dr, iso, ang, mode=mode, policy=policy)
                assert rc == 0 and kernel in gpu.last_kernel(), (msg, gpu.last_kernel())
                differ = int((dst != ref).sum())
                # flagged pixels go through the double-precision kernels, whose summation order differs from the replay's
                assert differ <= flagged, (W, H, sr, dr, ang, mode, differ, flagged, quad)
                assert rel_err(dst, ref).max() <= 3e-7
    finally:
        hostemu.aai_emu_use_quad(0)


def test_wide_kernel_equals_its_cpu_replay_bit_for_bit(gpu, hostemu, po):
    """aai_wide_kernel (a dst pixel = 4 or 16 lanes, one per part of its window, partial sums exchanged in a butterfly) against
    the CPU replay, which evaluates the parts one after the other and adds them in the butterfly's order: bit for bit outside
    the pixels the scan leaves to the double-precision pass.  2 x 2 and 4 x 4 parts, every part size (5 ... 8), every
    quadrant, both policies, hiPrec (close to an axis), 8-bit sources, more than one tile row and column."""
    rng = np.random.default_rng(79)
    hostemu.aai_emu_use_quad(1)
    fast_kernels = set()
    try:
        for (W, H, sr, dr, ang, policy, parts) in ((512, 400, 8.0, 1.0, 17.5, 0, 2), (400, 512, 6.0, 1.0, 45.0, 1, 2), (640, 512, 10.5, 1.0, 123.0, 0, 2),
                                                   (512, 512, 7.0, 1.0, 211.0, 0, 2), (600, 600, 9.3, 1.1, 300.0, 0, 2), (700, 512, 16.0, 1.0, 45.0, 0, 4),
                                                   (800, 640, 21.0, 1.0, 100.0, 1, 4), (640, 800, 13.0, 1.0, 250.0, 0, 4), (512, 512, 9.0, 1.0, 0.7, 0, 2),
                                                   (900, 500, 26.0, 1.0, 88.5, 0, 4), (1400, 1100, 8.0, 1.0, 17.5, 0, 2), (480, 400, 6.0, 1.0, 17.5, 0, 2)):
            iso = (float(rng.uniform(0.3 * W, 0.7 * W)), float(rng.uniform(0.3 * H, 0.7 * H)))
            src = rng.random((H, W)).astype(np.float32)
            rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=1, policy=policy)
            assert hostemu.aai_emu_wide_parts(ctypes.byref(rq)) == parts, (sr, ang)
            ref, axis = hostemu.resample(rq, src)
            wide, flagged = hostemu.quad_stats()
            rc, msg, dst, _, lay = gpu.resample_host(src, sr, dr, iso, ang, mode=1, policy=policy)
            assert rc == 0 and "aai_wide_kernel" in gpu.last_kernel(), (msg, gpu.last_kernel())
            differ = int((dst != ref).sum())
            assert wide > 0 and differ <= flagged, (W, H, sr, dr, ang, differ, flagged, wide)
            assert rel_err(dst, ref).max() <= 3e-7
            # fast mode over the same footprint: aai_wide_fast_kernel (its window of centres is two positions narrower: the
            # smallest of these geometries fit one window and take aai_quad_fast_kernel)
            rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=2, policy=policy)
            ref, axis = hostemu.resample(rq, src)
            wide, flagged = hostemu.quad_stats()
            rc, msg, dst, _, lay = gpu.resample_host(src, sr, dr, iso, ang, mode=2, policy=policy)
            want = "aai_wide_fast_kernel" if hostemu.aai_emu_wide_parts(ctypes.byref(rq)) else "aai_quad_fast_kernel"
            assert rc == 0 and want in gpu.last_kernel(), (msg, gpu.last_kernel(), sr, ang)
            fast_kernels.add(want)
            assert wide > 0 and int((dst != ref).sum()) <= flagged, (W, H, sr, dr, ang, int((dst != ref).sum()), flagged, wide)
            assert rel_err(dst, ref).max() <= 3e-7
        assert fast_kernels == {"aai_wide_fast_kernel", "aai_quad_fast_kernel"}, fast_kernels
        isrc = rng.integers(0, 256, size=(480, 640)).astype(np.uint8)
        rq = gpu.make_request(640, 480, 8.0, 1.0, (300.0, 250.0), 33.0, mode=1)
        ref, _ = hostemu.resample(rq, isrc.astype(np.float32))
        wide, flagged = hostemu.quad_stats()
        rc, msg, dst, _, _ = gpu.resample_host(isrc, 8.0, 1.0, (300.0, 250.0), 33.0, mode=1)
        assert rc == 0 and "aai_wide_kernel" in gpu.last_kernel() and int((dst != ref).sum()) <= flagged
    finally:
        hostemu.aai_emu_use_quad(0)


def test_cell_kernel_equals_its_cpu_replay_bit_for_bit(gpu, hostemu, po):
    """The cell kernel (one lane per cell of the dst grid, every (dst, src) pair evaluated once and shared between the dst
    pixels it feeds: csrc/aai_rot_cell.hpp) against the CPU replay of the same header, which adds a dst pixel's four parts
    in the kernel's order: bit for bit outside the pixels the scan leaves to the double-precision pass.  Plain fp32 and
    8-bit sources, every quadrant, both policies, up- and down-sampling, a strip boundary (more than 63 dst columns) and
    row bands of the kernel's strips (more than 32 dst rows)."""
    rng = np.random.default_rng(78)
    hostemu.aai_emu_use_cell(1)
    gpu.debug_cell_min_waves(0)
    try:
        for (W, H, sr, dr, ang, policy) in ((200, 160, 8192.0, 2731.0, 17.5, 0), (96, 96, 1.0, 4.0, 45.0, 0), (180, 140, 4.0, 1.0, 0.5, 0),
                                            (150, 150, 2.0, 1.0, 117.3, 1), (120, 90, 1.0, 1.0, 200.0, 0), (128, 128, 3.0, 2.0, 300.0, 0),
                                            (256, 64, 2.5, 1.0, 33.0, 0), (90, 200, 1.0, 1.9, 251.0, 0)):
            iso = (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
            src = rng.random((H, W)).astype(np.float32)
            rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=1, policy=policy)
            ref, axis = hostemu.resample(rq, src)
            cell, flagged = hostemu.quad_stats()
            rc, msg, dst, _, lay = gpu.resample_host(src, sr, dr, iso, ang, mode=1, policy=policy)
            assert rc == 0 and "aai_cell_kernel" in gpu.last_kernel(), (msg, gpu.last_kernel())
            differ = int((dst != ref).sum())
            assert cell > 0 and differ <= flagged, (W, H, sr, dr, ang, differ, flagged, cell)
            assert rel_err(dst, ref).max() <= 3e-7
    finally:
        hostemu.aai_emu_use_cell(0)
        gpu.debug_cell_min_waves(-1)


def test_a_nan_source_pixel_stays_local(gpu):
    """A NaN (or Inf) in the source reaches exactly the dst pixels whose footprint overlaps that pixel: a pair with area 0 must not
    carry 0 x NaN into a dst pixel it does not touch (the cell kernel feeds four dst pixels from every pair, and its fast path for
    finite values is taken by a vote over the wave).  Every fp32 formulation, fast mode, up- and down-sampling."""
    rng = np.random.default_rng(12)
    gpu.debug_cell_min_waves(0)
    try:
        for (W, H, sr, dr, ang, mode) in ((160, 120, 3.0, 1.0, 17.5, 1), (64, 64, 1.0, 4.0, 45.0, 1), (200, 160, 8192.0, 2731.0, 200.0, 1), (400, 400, 8.0, 1.0, 33.0, 1),
                                          (160, 120, 3.0, 1.0, 17.5, 2), (400, 400, 8.0, 1.0, 33.0, 2), (64, 64, 1.0, 2.0, 30.0, 2)):
            iso = ((W - 1) / 2, (H - 1) / 2)
            src = rng.random((H, W)).astype(np.float32) + 0.5
            y, x = H // 2 + 3, W // 2 - 5
            for bad in (np.nan, np.inf):
                poisoned = src.copy(); poisoned[y, x] = bad
                zeroed = src.copy(); zeroed[y, x] = 0.0
                delta = np.zeros_like(src); delta[y, x] = 1.0
                rc, msg, out_bad, _, _ = gpu.resample_host(poisoned, sr, dr, iso, ang, mode=mode)
                kernel = gpu.last_kernel()
                rc2, _, out_zero, _, _ = gpu.resample_host(zeroed, sr, dr, iso, ang, mode=mode)
                rc3, _, out_delta, _, _ = gpu.resample_host(delta, sr, dr, iso, ang, mode=mode)
                assert rc == 0 and rc2 == 0 and rc3 == 0, msg
                touched = out_delta > 0
                broken = ~np.isfinite(out_bad)
                assert touched.any() and np.array_equal(broken, touched), (W, sr, ang, mode, kernel, int(touched.sum()), int(broken.sum()))
                assert np.array_equal(out_bad[~broken], out_zero[~broken]), (W, sr, ang, mode, kernel)
    finally:
        gpu.debug_cell_min_waves(-1)


def test_precision_check_tells_benign_from_risky_data(gpu):
    """precision_check: uniform noise and the dose-like image are far inside 1e-5 under the fp32 kernels; a value a
    million times below its neighbours is not, and the helper says so."""
    from oracle import pyoracle as po                  # (only for the synthetic dose image)
    rng = np.random.default_rng(14)
    benign = rng.random((300, 280)).astype(np.float32)
    dev, _, _ = gpu.precision_check(benign, 3.0, 1.0, (139.5, 149.5), 17.5)
    assert dev <= 2e-6, dev
    dose = po.dose_image(400, 400, 2).astype(np.float32)
    dev, _, _ = gpu.precision_check(dose, 5.9, 1.0, (199.5, 199.5), 1.5)
    assert dev <= 1e-5, dev
    spiky = np.full((200, 200), 1e-6, dtype=np.float32)
    spiky[::7, ::5] = 1.0e3                               # isolated spikes nine decades above the background
    dev, a, b = gpu.precision_check(spiky, 1.0, 2.0, (99.5, 99.5), 30.0, floor=1e-12)
    assert dev > 1e-5 and np.isfinite(a).all() and np.isfinite(b).all(), dev


def test_double_precision_policy(gpu, po):
    """AAI_POLICY_DOUBLE_PRECISION routes general rotations to the double-precision kernels: exact to fp32 rounding on
    the geometry class where the fp32 formulation has its tail (dst values far below their neighbours: slight
    up-sampling within a degree of an axis), both modes, plain and interleaved."""
    W, H, sr, dr, ang, iso = 203, 367, 2.6611805249461478, 2.9963661425245225, 90.69816691569076, (101.0, 183.0)
    rng = np.random.default_rng(5)
    src = rng.random((H, W)).astype(np.float32)
    isrc = rng.random((H, W, 3)).astype(np.float32)
    flag = gpu.POLICY_DOUBLE_PRECISION
    for mode, omode in ((1, po.MODE_EXACT), (2, po.MODE_FAST)):
        for policy in (0, 1):
            gold = po.oracle_run(omode, src.astype(np.float64), sr, dr, iso, ang, policy=policy).dst
            rc, msg, dst, _, lay = gpu.resample_host(src, sr, dr, iso, ang, mode=mode, policy=policy)
            assert rc == 0 and ("aai_quad_kernel" if mode == 1 else "aai_quad_fast_kernel") in gpu.last_kernel() and rel_err(dst, gold).max() <= TOL
            rc, msg, dst, _, lay = gpu.resample_host(src, sr, dr, iso, ang, mode=mode, policy=policy | flag)
            assert rc == 0 and "quad" not in gpu.last_kernel() and "cell" not in gpu.last_kernel(), gpu.last_kernel()
            assert rel_err(dst, gold).max() <= 1.5e-7 and np.array_equal(gold == 0, dst == 0)
        rc, msg, idst, ilay = gpu.resample_interleaved_host(isrc, sr, dr, iso, ang, mode=mode, policy=1 | flag)
        assert rc == 0 and "quad" not in gpu.last_kernel() and "cell" not in gpu.last_kernel()
        for c in range(3):
            g = po.oracle_run(omode, isrc[:, :, c].astype(np.float64), sr, dr, iso, ang, policy=1).dst
            assert rel_err(idst[:, :, c], g).max() <= 1.5e-7
        # multiples of 90 degrees and the samplers have no fp32 formulation to leave: the flag changes nothing there
        for (a2, m2) in ((90.0, mode), (17.5, 3)):
            rc, msg, a, _, _ = gpu.resample_host(src, sr, dr, iso, a2, mode=m2, policy=0)
            rc2, msg2, b, _, _ = gpu.resample_host(src, sr, dr, iso, a2, mode=m2, policy=flag)
            assert rc == 0 and rc2 == 0 and np.array_equal(a, b), (a2, m2)


def test_outputs_taller_than_one_grid(gpu, po):
    """More than 1,048,560 output rows (65,535 tiles of 16: the most one launch's grid.y can carry): the rotated kernels,
    their one-off scans and the samplers go band by band (aai_rotated.hip: launch_rotated_typed), K1 doubles its rows per
    workgroup.  A 40 x 1,100,000 image; sampled row bands against the CPU oracle's rows."""
    import torch
    W, H = 40, 1_100_000
    src = torch.empty((H, W), dtype=torch.float32, device="cuda")
    gpu.synth_device(src.data_ptr(), W, H, W, 3)
    host = src.cpu().numpy()
    for (sr, dr, ang, mode, omode) in ((1.0, 1.0, 0.05, 1, po.MODE_EXACT), (1.0, 1.0, 0.0, 1, po.MODE_EXACT), (1.0, 1.0, 0.05, 2, po.MODE_FAST),
                                        (1.0, 1.0, 0.05, 3, 3)):
        iso = ((W - 1) / 2, (H - 1) / 2)
        rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=mode)
        rc, msg, lay = gpu.query(rq)
        assert rc == 0 and lay.dst_height > 65535 * 16, (msg, lay.dst_height)
        out = _device_run(gpu, rq, src)
        dH, dW = lay.dst_height, lay.dst_width
        for r0 in (0, 65535 * 16 - 8, 65535 * 16 + 40, dH - 24):
            gold = po.oracle_rows(omode, host, sr, dr, iso, ang, r0, r0 + 16, dW)
            got = out[r0:r0 + 16].cpu().numpy()
            if mode == 3:
                assert np.abs(got - gold).max() <= 2e-5, (ang, mode, r0)
            else:
                assert rel_err(got, gold).max() <= TOL, (ang, mode, r0, gpu.last_kernel())
                assert np.array_equal(gold == 0, got == 0), (ang, mode, r0)
        del out


def test_concurrent_host_threads(gpu):
    """Four host threads, each with its own HIP stream, issue 60 requests over 45 distinct geometries (more than the plan
    cache holds per device, so plans are built, reused and evicted under contention; ctypes releases the GIL during the
    calls): every result equals the one a single thread computed before, bit for bit, and errors stay per thread."""
    import threading
    import torch
    geoms = []
    for k in range(45):
        W, H = 96 + 8 * (k % 7), 80 + 4 * (k % 5)
        geoms.append((W, H, float(1 + k % 4), 1.0, ((W - 1) / 2, (H - 1) / 2), (0.0, 90.0, 17.5, 45.0, 200.0, 0.5)[k % 6], 1 + k % 4))
    rng = np.random.default_rng(3)
    srcs = {}
    for (W, H, *_rest) in geoms:
        if (W, H) not in srcs:
            srcs[(W, H)] = torch.from_numpy(rng.random((H, W)).astype(np.float32)).cuda()

    def run(g, stream):
        W, H, sr, dr, iso, ang, mode = g
        rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=mode)
        rc, msg, lay = gpu.query(rq)
        assert rc == 0, msg
        out = torch.empty((lay.dst_height, lay.dst_width), dtype=torch.float32, device="cuda")
        gpu.resample_device(rq, srcs[(W, H)].data_ptr(), W, out.data_ptr(), lay.dst_width, stream.cuda_stream)
        stream.synchronize()
        return out.cpu().numpy()

    main = torch.cuda.current_stream()
    expected = [run(g, main) for g in geoms]
    failures = []

    def worker(tid):
        try:
            stream = torch.cuda.Stream()
            order = np.random.default_rng(100 + tid).integers(0, len(geoms), size=60)
            for k in order:
                got = run(geoms[int(k)], stream)
                if not np.array_equal(got, expected[int(k)]):
                    failures.append((tid, int(k), "result differs"))
                if tid == 0 and k % 7 == 0:         # an error on this thread must not leak into the others' last_error
                    rc, msg, _lay = gpu.query(gpu.make_request(8, 8, 1.0, 0.0, (0.0, 0.0), 0.0))
                    if rc == 0 or "resolution" not in msg:
                        failures.append((tid, int(k), "error text", msg))
        except Exception as e:                      # noqa: BLE001
            failures.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures[:5]


def test_interleaved_channels_through_the_cell_kernel(gpu, po):
    """Interleaved area requests in the cell formulation (aai_cell_multi_kernel: areas once per (dst, src) pair, one sum per channel):
    every channel must equal the plain cell kernel's result on that channel alone BIT FOR BIT (same per-cell arithmetic, same order of
    additions), for 8- / 16-bit / fp32 pixels, 2 - 4 channels, with and without replication, in all quadrants, padded strides and batches;
    one case against the oracle and one large enough to take the kernel without the hint."""
    import torch
    rng = np.random.default_rng(77)
    gpu.debug_cell_min_waves(0)          # AAI_POLICY_PREFER_CELL: small outputs take the cell kernels too
    try:
        cases = [  # W, H, srcRes, dstRes, angle, C, dtype
            (300, 260, 2.39, 1.0, 30.0, 3, np.uint8), (256, 256, 3.0, 1.0, 17.5, 4, np.uint8), (200, 180, 2.0, 1.0, 107.5, 3, np.float32),
            (160, 200, 2.0, 1.0, 200.0, 2, np.uint16), (128, 96, 1.0, 2.0, 30.0, 3, np.uint8), (120, 90, 2.0, 1.0, 290.0, 4, np.uint16),
            (96, 128, 1.0, 3.0, 45.0, 2, np.float32), (140, 100, 3.0, 1.0, 61.0, 2, np.float32), (150, 130, 1.5, 1.0, 333.0, 3, np.uint16),
            (140, 100, 2.2, 1.0, 17.5, 4, np.float32),
        ]
        for k, (W, H, sr, dr, ang, C, dt) in enumerate(cases):
            iso = ((W - 1) / 2, (H - 1) / 2) if k % 2 else (float(rng.uniform(0, W)), float(rng.uniform(0, H)))
            if dt == np.float32:
                src = rng.random((H, W, C)).astype(np.float32)
            else:
                src = rng.integers(0, np.iinfo(dt).max + 1, size=(H, W, C)).astype(dt)
            rc, msg, dst, lay = gpu.resample_interleaved_host(src, sr, dr, iso, ang)
            assert rc == 0, (k, msg)
            assert "aai_cell_multi_kernel" in gpu.last_kernel(), (k, gpu.last_kernel())
            for c in range(C):
                rc, msg, planar, giso, _ = gpu.resample_host(np.ascontiguousarray(src[:, :, c]), sr, dr, iso, ang)
                assert rc == 0, msg
                assert "aai_cell_kernel" in gpu.last_kernel(), gpu.last_kernel()
                assert np.array_equal(dst[:, :, c], planar), (k, W, H, sr, dr, ang, C, c, dt)
        # against the oracle itself (reference policy and the exact one)
        src = rng.random((110, 140, 3)).astype(np.float32)
        for policy, omode in ((0, po.MODE_EXACT),):
            rc, msg, dst, lay = gpu.resample_interleaved_host(src, 2.5, 1.0, (69.5, 54.5), 33.0, policy=policy)
            assert rc == 0 and "aai_cell_multi_kernel" in gpu.last_kernel(), (msg, gpu.last_kernel())
            for c in range(3):
                gold = po.oracle_run(omode, src[:, :, c].astype(np.float64), 2.5, 1.0, (69.5, 54.5), 33.0, policy=policy)
                assert rel_err(dst[:, :, c], gold.dst).max() <= TOL and np.array_equal(gold.dst == 0, dst[:, :, c] == 0)
        # device entry: a batch of interleaved images with padded strides, NaN in one pixel of one channel
        B, W, H, C = 3, 220, 170, 3
        rq = gpu.make_request(W, H, 2.0, 1.0, ((W - 1) / 2, (H - 1) / 2), 33.0)
        rc, msg, lay = gpu.query(rq)
        hsrc = rng.random((B, H, W + 5, C)).astype(np.float32)
        hsrc[1, 80, 100, 1] = np.nan
        dsrc = torch.from_numpy(hsrc).cuda()
        ddst = torch.full((B, lay.dst_height, lay.dst_width + 2, C), -5.0, dtype=torch.float32, device="cuda")
        gpu.resample_interleaved_device(rq, C, dsrc.data_ptr(), (W + 5) * C, ddst.data_ptr(), (lay.dst_width + 2) * C,
                                        torch.cuda.current_stream().cuda_stream, batch=B, src_image_stride=H * (W + 5) * C,
                                        dst_image_stride=lay.dst_height * (lay.dst_width + 2) * C)
        torch.cuda.synchronize()
        assert "aai_cell_multi_kernel" in gpu.last_kernel(), gpu.last_kernel()
        got = ddst.cpu().numpy()
        assert np.all(got[:, :, lay.dst_width:, :] == -5.0)           # the padding is not written
        for b in range(B):
            for c in range(C):
                rc, msg, planar, _, _ = gpu.resample_host(np.ascontiguousarray(hsrc[b, :, :W, c]), 2.0, 1.0, ((W - 1) / 2, (H - 1) / 2), 33.0)
                assert rc == 0, msg
                assert np.array_equal(got[b, :, :lay.dst_width, c], planar, equal_nan=True), (b, c)
        # (the NaN reaches only the dst pixels its source pixel overlaps, and only in its own channel)
        assert np.isnan(got[1, :, :lay.dst_width, 1]).sum() in range(1, 10) and not np.isnan(got[1, :, :lay.dst_width, 0]).any()
    finally:
        gpu.debug_cell_min_waves(-1)
    # large enough for the cell kernels without the hint: RGB fp32, 1600 x 1400 at 2 : 1, and RGB 8-bit x2 up-sampling (8-bit pixels
    # without replication stay on the quad kernel: it is the faster one there)
    for (W, H, sr, dr, dt) in ((1600, 1400, 2.0, 1.0, np.float32), (700, 600, 1.0, 2.0, np.uint8), (1600, 1400, 2.0, 1.0, np.uint8)):
        C = 3
        src = rng.random((H, W, C)).astype(np.float32) if dt == np.float32 else rng.integers(0, 256, size=(H, W, C)).astype(np.uint8)
        rc, msg, dst, lay = gpu.resample_interleaved_host(src, sr, dr, ((W - 1) / 2, (H - 1) / 2), 17.5)
        assert rc == 0, msg
        assert ("aai_cell_multi_kernel" in gpu.last_kernel()) == (dt == np.float32 or dr > sr), (dt, sr, dr, gpu.last_kernel())
        if "aai_cell_multi_kernel" in gpu.last_kernel():
            rc, msg, planar, _, _ = gpu.resample_host(np.ascontiguousarray(src[:, :, 2]), sr, dr, ((W - 1) / 2, (H - 1) / 2), 17.5)
            assert rc == 0 and "aai_cell_kernel" in gpu.last_kernel() and np.array_equal(dst[:, :, 2], planar)


def test_sources_larger_than_4_gib(gpu, po):
    """A 33,000 x 33,000 fp32 source (4.36 GB: byte offsets from the image's first element no longer fit 32 bits) and, for
    K1, a 46,500 x 46,500 one (8.6 GB, 2.16 G elements: past 32-bit element offsets too): K1 streams them, the fp32
    window kernels of the rotated requests (one window per dst pixel, or a window in parts for wide footprints) take their 32-bit
    lane offsets from an anchor row per wave (QuadMap::anchorRows) and the cell kernel rebases every wave on its own first source row
    (QuadMap::rebaseWaves), in every quadrant.  Sampled row bands against the CPU oracle's rows."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    for (W, H, cases) in ((33000, 33000, ((4.0, 1.0, 0.0, 1, po.MODE_EXACT), (3.0, 1.0, 17.5, 1, po.MODE_EXACT), (3.0, 1.0, 17.5, 2, po.MODE_FAST),
                                          (4.0, 1.0, 90.0, 1, po.MODE_EXACT), (3.0, 1.0, 107.5, 1, po.MODE_EXACT), (2.5, 1.0, 200.0, 2, po.MODE_FAST),
                                          (3.0, 1.0, 290.0, 1, po.MODE_EXACT),
                                          # wide footprints: a dst pixel is 4 lanes, each with a part of the window
                                          (8.0, 1.0, 17.5, 1, po.MODE_EXACT), (8.0, 1.0, 107.5, 2, po.MODE_FAST), (11.0, 1.0, 225.0, 1, po.MODE_EXACT))),
                          (46500, 46500, ((4.0, 1.0, 0.0, 1, po.MODE_EXACT), (5.0, 1.0, 180.0, 2, po.MODE_FAST)))):
        src = torch.empty((H, W), dtype=torch.float32, device="cuda")
        gpu.synth_device(src.data_ptr(), W, H, W, 11)
        host = src.cpu().numpy()
        for (sr, dr, ang, mode, omode) in cases:
            iso = ((W - 1) / 2, (H - 1) / 2)
            rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=mode)
            rc, msg, lay = gpu.query(rq)
            assert rc == 0, msg
            out = _device_run(gpu, rq, src)
            assert any(t in gpu.last_kernel() for t in ("quad", "wide", "cell")) == (ang % 90.0 != 0.0), gpu.last_kernel()
            # (plain area requests at 3:1: the cell kernel, every wave with its offsets rebased on its own first source row)
            if mode == 1 and sr == 3.0:
                assert "cell" in gpu.last_kernel(), gpu.last_kernel()
            dH, dW = lay.dst_height, lay.dst_width
            for r0 in (0, dH // 3, dH // 2, dH - 8):
                gold = po.oracle_rows(omode, host, sr, dr, iso, ang, r0, r0 + 8, dW)
                got = out[r0:r0 + 8].cpu().numpy()
                assert rel_err(got, gold).max() <= TOL, (W, ang, mode, r0, gpu.last_kernel(), float(rel_err(got, gold).max()))
                assert np.array_equal(gold == 0, got == 0), (W, ang, mode, r0)
            del out
        del src, host
        torch.cuda.empty_cache()


def test_dense_knife_geometry_takes_the_strict_pass_for_the_whole_image(po):
    """When (nearly) every dst pixel is flagged the plan keeps no list and the double-precision pass computes the whole image
    (threshold 16 M pixels; lowered to 10 through the test hook AAI_MAX_LISTED_PIXELS, in a child process because the
    hook is read once)."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
import torch            # (before the library: one HIP runtime per process)
sys.path.insert(0, %r)
import area_average_interpolation_amd as aai
from oracle import pyoracle as po
aai.set_device(0)
for (W, H, sr, dr, ang, mode, omode) in ((96, 80, 2.0, 1.0, 45.0, 1, po.MODE_EXACT), (96, 80, 2.0, 1.0, 30.0, 1, po.MODE_EXACT), (96, 80, 2.0, 1.0, 45.0, 2, po.MODE_FAST)):
    src = po.synth_image(W, H, 5)
    iso = ((W - 1) / 2, (H - 1) / 2)
    rc, msg, dst, giso, lay = aai.resample_host(src, sr, dr, iso, ang, mode=mode)
    assert rc == 0, msg
    if mode == 1:
        assert "strict" in aai.last_kernel(), aai.last_kernel()
    gold = po.oracle_run(omode, src.astype(np.float64), sr, dr, iso, ang).dst
    err = np.abs(dst - gold) / np.maximum(np.abs(gold), 1e-3)
    assert err.max() <= 1e-5 and np.array_equal(gold == 0, dst == 0), (W, H, ang, mode, float(err.max()))
# an axis-aligned geometry whose model scan flags more pixels than the list keeps (13 of 39 here): the whole image takes the
# per-pixel replay instead of K1 -- plain, as a band from a footprint-only buffer, and interleaved
W, H, sr, dr, ang = 40, 9, 3.0, 1.0, 0.0
iso = ((W - 1) / 2, (H - 1) / 2)
src = po.synth_image(W, H, 7)
gold = po.oracle_run(po.MODE_EXACT, src.astype(np.float64), sr, dr, iso, ang).dst
rc, msg, dst, giso, lay = aai.resample_host(src, sr, dr, iso, ang, mode=1)
assert rc == 0 and "strict" in aai.last_kernel(), (msg, aai.last_kernel())
assert (np.abs(dst - gold) / np.maximum(np.abs(gold), 1e-3)).max() <= 1e-6
rq = aai.make_request(W, H, sr, dr, iso, ang, mode=1)
dH, dW = gold.shape
a, b = aai.band_source_rows(rq, 1, dH)
band_src = torch.from_numpy(src[a:b].copy()).cuda()
band_dst = torch.empty((dH - 1, dW), dtype=torch.float32, device="cuda")
aai.resample_band_device(rq, 1, dH, band_src.data_ptr(), W, band_dst.data_ptr(), dW, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert np.array_equal(band_dst.cpu().numpy(), dst[1:]), "band"
inter = np.stack([src, 1.0 - src], axis=2).astype(np.float32)
rc, msg, idst, ilay = aai.resample_interleaved_host(inter, sr, dr, iso, ang, mode=1)
assert rc == 0 and (np.abs(idst[:, :, 0] - gold) / np.maximum(np.abs(gold), 1e-3)).max() <= 1e-6
print("dense ok")
""" % ROOT
    env = dict(os.environ, AAI_MAX_LISTED_PIXELS="10")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "dense ok" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


@pytest.mark.gpu
def test_production_kernels_leave_exactly_the_flagged_pixels_alone(gpu):
    """The fp32 kernels skip the dst pixels the plan's scans flagged (the double-precision pass writes those, beside them on a side
    stream): with that pass switched off (tests-only hook aai_debug_skip_fixup) a dst buffer filled with a sentinel must keep the
    sentinel in exactly `flagged` pixels (aai_plan_info), and the normal call must differ from that run in exactly those pixels.
    Every skipping kernel: cell (with its one-bit-per-tile summary of the masks: strips without a flagged tile never look at the
    masks), quad, fast (16 x 4 and row-shaped waves), wide area / fast; plain fp32, up- and down-sampling."""
    import re
    import torch
    SENTINEL = -7.0
    st = torch.cuda.current_stream().cuda_stream
    cases = [  # (W, H, srcRes, dstRes, angle, mode, kernel)
        (2048, 2048, 8192.0, 2731.0, 17.5, 1, "aai_cell_kernel"), (512, 512, 1.0, 4.0, 45.0, 1, "aai_cell_kernel"), (1500, 1100, 1.0, 1.0, 30.0, 1, "aai_cell_kernel"),
        (300, 260, 3.0, 1.0, 30.0, 1, "aai_quad_kernel"),
        (2048, 2048, 8192.0, 2731.0, 17.5, 2, "aai_quad_fast_kernel"), (512, 512, 1.0, 4.0, 40.0, 2, "aai_quad_fast_kernel"), (1200, 1000, 1.5, 1.0, 30.0, 2, "aai_quad_fast_kernel"),
        (512, 512, 1.0, 2.0, 45.0, 2, "aai_quad_fast_kernel"), (600, 600, 2.0, 1.0, 45.0, 2, "aai_quad_fast_kernel"), (512, 512, 1.0, 4.0, 45.0, 2, "aai_quad_fast_kernel"),
        (2048, 2048, 8.0, 1.0, 17.5, 1, "aai_wide_kernel"), (2048, 2048, 8.0, 1.0, 17.5, 2, "aai_wide_fast_kernel"),
    ]
    exercised = set()
    try:
        for (W, H, sr, dr, ang, mode, kernel) in cases:
            iso = ((W - 1) / 2.0, (H - 1) / 2.0)         # isocenter on the lattice: the symmetric geometry has knife-edge pixels at these angles
            rq = gpu.make_request(W, H, sr, dr, iso, ang, mode=mode)
            rc, msg, lay = gpu.query(rq)
            assert rc == 0, msg
            src = torch.rand((H, W), dtype=torch.float32, device="cuda") + 0.25
            gpu.prepare(rq)
            m = re.search(r"flagged=(\d+) dense=(\d+)", gpu.plan_shape(rq))
            assert m, gpu.plan_shape(rq)
            flagged, dense = int(m.group(1)), int(m.group(2))
            assert dense == 0, (W, H, sr, dr, ang, mode, gpu.plan_shape(rq))
            if flagged == 0:
                continue                                   # (fast mode flags little: a centre has to lie within ~1e-7 of an edge)
            full = torch.full((lay.dst_height, lay.dst_width), SENTINEL, dtype=torch.float32, device="cuda")
            gpu.resample_device(rq, src.data_ptr(), W, full.data_ptr(), lay.dst_width, st)
            torch.cuda.synchronize()
            assert kernel in gpu.last_kernel(), (kernel, gpu.last_kernel())
            assert int((full == SENTINEL).sum().item()) == 0
            gpu.debug_skip_fixup(True)
            holes = torch.full((lay.dst_height, lay.dst_width), SENTINEL, dtype=torch.float32, device="cuda")
            gpu.resample_device(rq, src.data_ptr(), W, holes.data_ptr(), lay.dst_width, st)
            torch.cuda.synchronize()
            gpu.debug_skip_fixup(False)
            left = holes == SENTINEL
            assert int(left.sum().item()) == flagged, (kernel, W, H, sr, dr, ang, mode, int(left.sum().item()), flagged)
            assert torch.equal(holes[~left], full[~left]), (kernel, W, H, sr, dr, ang, mode)
            exercised.add(kernel)
        assert {"aai_cell_kernel", "aai_quad_kernel", "aai_quad_fast_kernel"} <= exercised, exercised
    finally:
        gpu.debug_skip_fixup(False)
