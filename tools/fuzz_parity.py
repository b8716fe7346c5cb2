#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (longer than the test-suite allows): random sizes, ratios, isocenters,
rotations (generic and structured), modes, policies, source types, batches with padded strides, and row bands.
usage: python tools/fuzz_parity.py [cases] [seed]      (FUZZ_MAX=<largest side>, FUZZ_CELL=1: the cell kernel for small outputs too)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import area_average_interpolation_amd as aai
from area_average_interpolation_amd import _lib as L
from oracle import pyoracle as po          # checker only (this is a test tool)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
REPORT = float(os.environ.get("FUZZ_REPORT", "1"))       # print the cases whose error exceeds this (default: none)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
aai.set_device(0)
if os.environ.get("FUZZ_CELL", "0") == "1":
    aai.debug_cell_min_waves(0)       # every plain rotated area request carries AAI_POLICY_PREFER_CELL: the cell kernel on the fuzz's small images too
st = torch.cuda.current_stream().cuda_stream
special = [0, 30, 45, 60, 90, 180, 270, math.degrees(math.atan(0.5)), 17.5, 1e-6, 89.999999]
worst, bad, tail = 0.0, 0, 0


def fp32_tail(e, kernel, run_double):
    """An error between 1e-5 and 3e-5 from an fp32 quad / cell kernel on a dst value far below its neighbours is the documented
    tail of the fp32 formulation (include/aai.h: AAI_POLICY_DOUBLE_PRECISION), provided the double-precision request
    of the same case is exact to 1e-6."""
    return e <= 3e-5 and ("quad" in kernel or "cell" in kernel) and run_double() <= 1e-6


for k in range(N):
    BIG = int(os.environ.get("FUZZ_MAX", "140"))
    W, H = int(rng.integers(1, BIG)), int(rng.integers(1, BIG))
    sr = float(rng.choice([1, 2, 3, 4, 5])) if k % 3 == 0 else float(rng.uniform(0.4, 8))
    dr = float(rng.choice([1, 2])) if k % 3 == 0 else float(rng.uniform(0.4, 3))
    if dr / sr > 2.2:
        dr = sr * 2.2
    ang = float(rng.choice(special)) + 90 * int(rng.integers(0, 4)) if k % 4 == 0 else float(rng.uniform(-400, 400))
    iso = ((W - 1) / 2, (H - 1) / 2) if k % 5 == 0 else (float(rng.uniform(-3, W + 3)), float(rng.uniform(-3, H + 3)))
    mode = int(rng.choice([1, 1, 2, 3, 4]))
    policy = int(rng.integers(0, 2)) if mode == 1 else 0
    dt = rng.choice(["f32", "u8", "u16"]) if mode in (1, 2) else "f32"
    if dt == "f32":
        src = rng.random((H, W)).astype(np.float32)
    else:
        src = rng.integers(0, 256 if dt == "u8" else 65536, size=(H, W)).astype(np.uint8 if dt == "u8" else np.uint16)
    scale = 1.0 if dt == "f32" else (256.0 if dt == "u8" else 65536.0)
    omode = {1: po.MODE_EXACT, 2: po.MODE_FAST, 3: 3, 4: 4}[mode]
    gold = po.oracle_run(omode, src.astype(np.float64), sr, dr, iso, ang, policy=policy)
    rc, msg, dst, giso, lay = aai.resample_host(src, sr, dr, iso, ang, mode=mode, policy=policy)
    if gold.dst.size == 0:
        assert rc == L.ERR_EMPTY_OUTPUT, (rc, msg)          # an extent that rounds to 0: the reference crashes, the library reports it
        continue
    assert rc == 0, msg
    tol = 2e-5 * scale if mode in (3, 4) else None
    if dst.size:
        if tol is not None:
            e = np.abs(dst - gold.dst).max() / tol * 1e-5
        else:
            e = (np.abs(dst - gold.dst) / np.maximum(np.abs(gold.dst), 1e-3 * scale)).max()
        zm = 0 if mode in (3, 4) else int(((gold.dst == 0) != (dst == 0)).sum())
        def again():
            rc2, _, d2, _, _ = aai.resample_host(src, sr, dr, iso, ang, mode=mode, policy=policy | L.POLICY_DOUBLE_PRECISION)
            return (np.abs(d2 - gold.dst) / np.maximum(np.abs(gold.dst), 1e-3 * scale)).max() if rc2 == 0 else 1.0
        if 1e-5 < e and not zm and mode in (1, 2) and fp32_tail(e, aai.last_kernel(), again):
            # counted separately AND as a failure: include/aai.h promises 1e-5, and a case past it is a miss whatever the cause
            tail += 1
            print("fp32 tail case (exact under AAI_POLICY_DOUBLE_PRECISION)", k, dict(W=W, H=H, sr=sr, dr=dr, ang=ang, mode=mode, policy=policy, dt=str(dt)), "err", e)
        worst = max(worst, e)
        if e > REPORT:
            print("near the bar: case", k, dict(W=W, H=H, sr=sr, dr=dr, ang=ang, iso=iso, mode=mode, policy=policy, dt=str(dt)), "err", e, aai.last_kernel())
        if e > 1e-5 or zm or dst.shape != gold.dst.shape or tuple(giso) != gold.dst_iso:
            bad += 1
            print("MISMATCH case", k, dict(W=W, H=H, sr=sr, dr=dr, ang=ang, iso=iso, mode=mode, policy=policy, dt=str(dt)), "err", e, "zero-mismatch", zm, aai.last_kernel())
    # interleaved channels: every channel against the oracle on that channel
    if k % 5 == 0 and mode in (1, 2):
        C = int(rng.integers(2, 5))
        if dt == "f32":
            isrc = rng.random((H, W, C)).astype(np.float32)
        else:
            isrc = rng.integers(0, 256 if dt == "u8" else 65536, size=(H, W, C)).astype(np.uint8 if dt == "u8" else np.uint16)
        rc, msg, idst, ilay = aai.resample_interleaved_host(isrc, sr, dr, iso, ang, mode=mode, policy=policy)
        assert rc == 0, msg
        for c in range(C):
            g = po.oracle_run(omode, isrc[:, :, c].astype(np.float64), sr, dr, iso, ang, policy=policy).dst
            if g.size:
                e = (np.abs(idst[:, :, c] - g) / np.maximum(np.abs(g), 1e-3 * scale)).max()
                def again():
                    rc2, _, d2, _ = aai.resample_interleaved_host(isrc, sr, dr, iso, ang, mode=mode, policy=policy | L.POLICY_DOUBLE_PRECISION)
                    return (np.abs(d2[:, :, c] - g) / np.maximum(np.abs(g), 1e-3 * scale)).max() if rc2 == 0 else 1.0
                if 1e-5 < e and fp32_tail(e, aai.last_kernel(), again):
                    tail += 1
                    print("fp32 tail case (exact under AAI_POLICY_DOUBLE_PRECISION)", k, dict(W=W, H=H, sr=sr, dr=dr, ang=ang, mode=mode, policy=policy, dt=str(dt), C=C, c=c), "err", e)
                worst = max(worst, e)
                if e > REPORT:
                    print("near the bar: case", k, dict(W=W, H=H, sr=sr, dr=dr, ang=ang, iso=iso, mode=mode, policy=policy, dt=str(dt), C=C, c=c), "err", e, aai.last_kernel())
                if e > 1e-5 or int(((g == 0) != (idst[:, :, c] == 0)).sum()):
                    bad += 1
                    print("CHANNEL MISMATCH case", k, dict(W=W, H=H, sr=sr, dr=dr, ang=ang, iso=iso, mode=mode, policy=policy, dt=str(dt), C=C, c=c), "err", e, aai.last_kernel())
    # row bands of the f32 cases must equal the full result bit for bit
    if dt == "f32" and lay.dst_height >= 32 and lay.dst_width > 0 and k % 2 == 0:
        rq = aai.make_request(W, H, sr, dr, iso, ang, mode=mode, policy=policy)
        t = torch.from_numpy(src).cuda()
        r0 = 16 * int(rng.integers(0, lay.dst_height // 16))
        r1 = int(rng.integers(r0 + 1, lay.dst_height + 1))
        a, b = aai.band_source_rows(rq, r0, r1)
        bs = t[a:b].clone()
        bd = torch.empty((r1 - r0, lay.dst_width), dtype=torch.float32, device="cuda")
        aai.resample_band_device(rq, r0, r1, bs.data_ptr(), W, bd.data_ptr(), lay.dst_width, st)
        torch.cuda.synchronize()
        if not np.array_equal(bd.cpu().numpy(), dst[r0:r1]):
            bad += 1
            print("BAND MISMATCH case", k, dict(W=W, H=H, sr=sr, dr=dr, ang=ang, iso=iso, mode=mode), r0, r1, a, b)
print("cases", N, "mismatching", bad, "worst relative error", worst, "| of the mismatches, fp32-tail cases (1e-5 < err <= 3e-5, exact under AAI_POLICY_DOUBLE_PRECISION):", tail)
sys.exit(1 if bad else 0)
