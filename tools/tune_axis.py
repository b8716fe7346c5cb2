#!/usr/bin/env python3
"""Sweep the launch knobs of the axis-aligned kernel on the cfg2 workload (through the library's experiment hook
aai_debug_axis_tune, the run-time form of AAI_AXIS_TUNE).  Variants are interleaved in ONE process, several rounds each;
reports median and min per variant, then what the plan's own autotune picked on this box (aai_plan_info).
Needs the experiments build: make -C area_average_interpolation_amd/csrc exp; AAI_LIB=area_average_interpolation_amd/libaai_hip_exp.so"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes
import torch
import area_average_interpolation_amd as aai
from area_average_interpolation_amd import _lib as L

lib = L.load()
lib.aai_debug_axis_tune.restype = None
lib.aai_debug_axis_tune.argtypes = [ctypes.c_char_p]

W = H = int(os.environ.get("TUNE_SIZE", "8192"))
B = int(os.environ.get("TUNE_BATCH", "4"))
ROUNDS = int(os.environ.get("TUNE_ROUNDS", "7"))
SH = float(os.environ.get("TUNE_ISO_SHIFT", "0"))
rq = aai.make_request(W, H, float(os.environ.get("TUNE_SRCRES", "4")), float(os.environ.get("TUNE_DSTRES", "1")), ((W - 1) / 2 + SH, (H - 1) / 2 + SH), float(os.environ.get("TUNE_ANGLE", "0")))
rc, msg, lay = aai.query(rq)
dW, dH = lay.dst_width, lay.dst_height
aai.set_device(0)
src = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
dst = torch.empty((B, dH, dW), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for b in range(B):
    aai.synth_device(src[b].data_ptr(), W, H, W, b + 1, stream)
torch.cuda.synchronize()
aai.prepare(rq)            # the plan (and its own launch-shape measurement) before any override is active
alg = B * (4 * W * H + 4 * dW * dH)


def run():
    aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, stream, batch=B, src_image_stride=W * H, dst_image_stride=dW * dH)


variants = []
if len(sys.argv) > 1:
    variants = sys.argv[1:]
else:
    for nt, swap in itertools.product((1, 0), (0,)):
        for rows in (1, 2, 4, 8):
            variants.append("nt=%d,rows=%d,swap=%d" % (nt, rows, swap))
ref = None
times = {v: [] for v in variants}
for r in range(ROUNDS):
    for v in variants:
        lib.aai_debug_axis_tune(v.encode())
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 3)
        if ref is None:
            ref = dst.clone()
        else:
            assert torch.equal(ref, dst), "variant %s changed the result" % v
best = sorted(variants, key=lambda v: sorted(times[v])[len(times[v]) // 2])
for v in best:
    t = sorted(times[v])
    med, mn = t[len(t) // 2], t[0]
    print("%-46s median %.1f us (%.0f GB/s, %.1f%% of 8 TB/s)  min %.1f us (%.0f GB/s)" % (
        v, med * 1e3, alg / med / 1e6, alg / med / 1e6 / 80, mn * 1e3, alg / mn / 1e6))
lib.aai_debug_axis_tune(b"")
print("plan autotune on this box:", aai.plan_shape(rq))
t = []
for r in range(ROUNDS):
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record()
    torch.cuda.synchronize()
    t.append(e0.elapsed_time(e1) / 3)
t.sort()
print("as shipped (plan's shape)                      median %.1f us (%.0f GB/s, %.1f%% of 8 TB/s)" % (t[len(t) // 2] * 1e3, alg / t[len(t) // 2] / 1e6, alg / t[len(t) // 2] / 1e6 / 80))
