#!/usr/bin/env python3
"""Structured randomised sweeps of the CPU replay (tests/emulation/host_emulation.cpp: the kernels' arithmetic, K1 tables +
model scan + fix-up, fp32 quad formulation + scans + strict fix-up) against the CPU oracle -- the geometries where the
reference's DBL_EPSILON rules decide: integer / rational ratios, isocenters on pixel centres, corners, half and quarter
pixels, rotations at multiples of 90 degrees ("axis") or at atan(p/q), 15-degree steps and hair-breadth angles ("rotated").
Needs no GPU; with --gpu the same cases go through the device library instead (on a GPU box).
"wide": the rotated angles at ratios 6:1 ... 20:1 on images up to 220 x 220 (the wide-footprint kernels).
usage: python tools/replay_sweep.py axis|rotated|wide [cases] [seed] [--gpu]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                   # noqa: E402
import area_average_interpolation_amd as aai         # noqa: E402
import conftest                                      # noqa: E402
from oracle import pyoracle as po                    # noqa: E402  (checker only: this is a test tool)

AXIS_RATIOS = [(5, 1), (3, 1), (2, 1), (4, 1), (7, 1), (1, 1), (5, 2), (3, 2), (1, 2), (2, 3), (2.5, 1), (1, 3), (6, 1), (9, 2), (7, 3), (4, 3),
               (1.5, 1), (3, 4), (10, 3), (8, 1)]
ROT_RATIOS = [(5, 1), (3, 1), (2, 1), (4, 1), (1, 1), (5, 2), (3, 2), (1, 2), (2, 3), (2.5, 1), (6, 1), (7, 3), (4, 3), (1.5, 1), (2 ** 0.5, 1),
              (2 * 2 ** 0.5, 1), (5 ** 0.5, 1), (1, 2 ** 0.5), (2.236067977, 2), (1.25, 1)]
WIDE_RATIOS = [(6, 1), (8, 1), (10, 1), (12, 1), (16, 1), (7, 1), (9, 1), (15, 2), (20, 1), (6 * 2 ** 0.5, 1), (13, 2), (25, 3)]      # "wide": footprints beyond one 8 x 8 window
ROT_ANGLES = [45.0, 30.0, 60.0, math.degrees(math.atan(0.5)), math.degrees(math.atan(2)), math.degrees(math.atan(0.75)), math.degrees(math.atan(1 / 3)),
              22.5, 15.0, 75.0, math.degrees(math.atan(0.25)), 1e-7, 89.9999999, 0.001, 1.0]


def sweep(kind, cases, seed, hostemu, report=print, gpu=False):
    """gpu=True: the same cases through the device library (aai.resample_host) instead of the CPU replay"""
    rng = np.random.default_rng(seed)
    n = bad = fixups = 0
    worst = 0.0
    if not gpu:
        hostemu.aai_emu_use_quad(1)
    try:
        for _ in range(cases):
            big = 48 if kind == "axis" else (220 if kind == "wide" else 40)
            W, H = int(rng.integers(1 if kind != "wide" else 40, big)), int(rng.integers(1 if kind != "wide" else 40, big))
            ratios = AXIS_RATIOS if kind == "axis" else (WIDE_RATIOS if kind == "wide" else ROT_RATIOS)
            sr, dr = ratios[int(rng.integers(0, len(ratios)))]
            if dr / sr > 2.2:
                continue
            iso = [((W - 1) / 2, (H - 1) / 2), (0.0, 0.0), (float(rng.integers(0, W)), float(rng.integers(0, H))), (W / 2, H / 2),
                   (rng.integers(0, 4 * W) / 4.0, rng.integers(0, 4 * H) / 4.0), (rng.integers(-2, W + 2) + 0.5, float(rng.integers(-2, H + 2)))][int(rng.integers(0, 6))]
            iso = (float(iso[0]), float(iso[1]))
            if kind == "axis":
                ang = float(rng.choice([0.0, 90.0, 180.0, 270.0, 360.0, -90.0]))
            else:
                ang = float(ROT_ANGLES[int(rng.integers(0, len(ROT_ANGLES)))] + 90 * int(rng.integers(-1, 4)))
            mode = int(rng.choice([1, 1, 2, 3, 4] if gpu else [1, 1, 2]))      # the samplers (build-defined, oracle = this repository's) on the device only
            policy = int(rng.integers(0, 2)) if mode == 1 else 0
            floor = 1e-3
            form = int(rng.integers(0, 4)) if gpu and mode in (1, 2) else 0      # device only: 8- / 16-bit sources, interleaved channels
            if form == 1:
                src = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
                floor = 0.256
            elif form == 2:
                src = rng.integers(0, 65536, size=(H, W)).astype(np.uint16)
                floor = 65.536
            else:
                src = rng.random((H, W)).astype(np.float32)
            gold = po.oracle_run({1: po.MODE_EXACT, 2: po.MODE_FAST, 3: 3, 4: 4}[mode], src.astype(np.float64), float(sr), float(dr), iso, ang, policy=policy).dst
            if gold.size == 0:
                continue
            if gpu and form == 3:
                C = int(rng.integers(2, 5))
                inter = rng.random((H, W, C)).astype(np.float32)
                k = int(rng.integers(0, C))
                inter[:, :, k] = src
                rc, msg, iout, _ = aai.resample_interleaved_host(inter, float(sr), float(dr), iso, ang, mode=mode, policy=policy)
                assert rc == 0, msg
                out = iout[:, :, k]
            elif gpu:
                rc, msg, out, _, _ = aai.resample_host(src, float(sr), float(dr), iso, ang, mode=mode, policy=policy)
                assert rc == 0, msg
            else:
                out, axis = hostemu.resample(aai.make_request(W, H, float(sr), float(dr), iso, ang, mode=mode, policy=policy), src)
                fixups += int(axis and hostemu.aai_emu_axis_fixups() > 0)
            if mode in (3, 4):
                err = float(np.abs(out - gold).max()) / 2.0          # absolute, 2e-5 of the value range counts as 1e-5
            else:
                err = float((np.abs(out - gold) / np.maximum(np.abs(gold), floor)).max())
            n += 1
            worst = max(worst, err)
            if err > 1e-5 or (mode in (1, 2) and not np.array_equal(gold == 0, out == 0)):
                bad += 1
                report("MISMATCH", dict(W=W, H=H, sr=sr, dr=dr, iso=iso, ang=ang, mode=mode, policy=policy, form=form), "err", err)
    finally:
        if not gpu:
            hostemu.aai_emu_use_quad(0)
    return n, bad, worst, fixups


if __name__ == "__main__":
    on_gpu = "--gpu" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--gpu"]
    kind = args[0] if len(args) > 0 else "axis"
    cases = int(args[1]) if len(args) > 1 else 8000
    seed = int(args[2]) if len(args) > 2 else 1
    emu = None
    if on_gpu:
        aai.set_device(0)
    else:
        emu = conftest.hostemu.__wrapped__(aai)
    t0 = time.time()
    n, bad, worst, fixups = sweep(kind, cases, seed, emu, gpu=on_gpu)
    print("%s sweep seed %d (%s): cases %d mismatching %d worst relative error %.3g  K1 cases with fix-ups %s  (%.0f s)" % (
        kind, seed, "GPU" if on_gpu else "CPU replay", n, bad, worst, "n/a" if on_gpu else fixups, time.time() - t0))
    sys.exit(1 if bad else 0)
