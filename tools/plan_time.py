#!/usr/bin/env python3
"""Time the one-off plan creation (tables, launch-shape measurement, model / knife / fp32-margin scans) per BASELINE geometry.
usage: [AAI_AXIS_AUTOTUNE=0] python tools/plan_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes
import torch
import area_average_interpolation_amd as aai
from area_average_interpolation_amd import _lib as L

lib = L.load()
lib.aai_debug_plan_shape.restype = ctypes.c_char_p
lib.aai_debug_plan_shape.argtypes = [ctypes.POINTER(L.Request)]

aai.set_device(0)
torch.zeros(1, device="cuda")
cases = [("cfg2 8192^2 -> 2048^2 @0", 8192, 8192, 4.0, 1.0, 0.0, 1), ("cfg4 4096^2 -> 1024^2 @0", 4096, 4096, 4.0, 1.0, 0.0, 1),
         ("8192x8191 3:1 @0 (odd size)", 8192, 8191, 3.0, 1.0, 0.0, 1), ("8192^2 5:1 @90", 8192, 8192, 5.0, 1.0, 90.0, 1),
         ("8191x8190 5:1 @0 (odd x even)", 8191, 8190, 5.0, 1.0, 0.0, 1), ("8192^2 6:1 @0 fast, integer iso", 8193, 8193, 6.0, 1.0, 0.0, 2),
         ("cfg3 8192^2 -> 3426^2 @17.5", 8192, 8192, 8192.0, 2731.0, 17.5, 1), ("cfg3 fast", 8192, 8192, 8192.0, 2731.0, 17.5, 2),
         ("cfg5 4096^2 x4 @45", 4096, 4096, 1.0, 4.0, 45.0, 1)]
for name, W, H, sr, dr, ang, mode in cases:
    rq = aai.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang, mode=mode)
    t0 = time.perf_counter()
    aai.prepare(rq)
    torch.cuda.synchronize()
    print("%-34s AAI_AXIS_AUTOTUNE=%s  prepare %.1f ms   %s" % (name, os.environ.get("AAI_AXIS_AUTOTUNE", "1"), 1e3 * (time.perf_counter() - t0),
                                                              lib.aai_debug_plan_shape(ctypes.byref(rq)).decode()))
