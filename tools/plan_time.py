#!/usr/bin/env python3
"""Time the one-off plan creation (tables, launch-shape measurement, model / knife / fp32-margin scans) per BASELINE geometry:
aai_prepare cold (first plan of its class in this process), then a second geometry of the same class (warm launch-shape
cache), then the same request again (cached plan).  usage: [AAI_AXIS_AUTOTUNE=0] [AAI_AXIS_CLASS_VERIFY=0] python tools/plan_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import area_average_interpolation_amd as aai

aai.set_device(0)
torch.zeros(1, device="cuda")


def prep(name, W, H, sr, dr, ang, mode, iso=None):
    rq = aai.make_request(W, H, sr, dr, iso or ((W - 1) / 2, (H - 1) / 2), ang, mode=mode)
    t0 = time.perf_counter()
    aai.prepare(rq)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    aai.prepare(rq)
    again = 1e3 * (time.perf_counter() - t0)
    print("%-46s prepare %8.2f ms   again %6.3f ms   %s" % (name, ms, again, aai.plan_shape(rq)), flush=True)


# first touch: the runtime loads each translation unit's code object on its first launch (~5-20 ms per unit, once per process)
t0 = time.perf_counter()
aai.debug_cell_min_waves(0)          # small images take the cell kernel too: its translation units are among the first touches
for (sr, ang, mode) in ((2.0, 0.0, 1), (2.0, 90.0, 1), (3.0, 17.5, 1), (3.0, 17.5, 2), (1.0, 33.0, 3), (8.0, 12.0, 1), (8.0, 12.0, 2)):
    img = torch.rand((96, 96), dtype=torch.float32, device="cuda")
    rq = aai.make_request(96, 96, sr, 1.0, (47.5, 47.5), ang, mode=mode)
    rc, msg, lay = aai.query(rq)
    out = torch.empty((lay.dst_height, lay.dst_width), dtype=torch.float32, device="cuda")
    aai.resample_device(rq, img.data_ptr(), 96, out.data_ptr(), lay.dst_width, torch.cuda.current_stream().cuda_stream)
    if mode == 1 and ang == 17.5:
        aai.debug_cell_min_waves(-1)     # ... and once more on the quad kernel
        rq2 = aai.make_request(96, 96, 2.5, 1.0, (47.5, 47.5), ang, mode=mode)
        rc, msg, lay2 = aai.query(rq2)
        out2 = torch.empty((lay2.dst_height, lay2.dst_width), dtype=torch.float32, device="cuda")
        aai.resample_device(rq2, img.data_ptr(), 96, out2.data_ptr(), lay2.dst_width, torch.cuda.current_stream().cuda_stream)
aai.debug_cell_min_waves(-1)
torch.cuda.synchronize()
print("first touch of every kernel family (code objects loaded): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
print("AAI_AXIS_AUTOTUNE=%s AAI_AXIS_CLASS_VERIFY=%s" % (os.environ.get("AAI_AXIS_AUTOTUNE", "1"), os.environ.get("AAI_AXIS_CLASS_VERIFY", "1")))
prep("cfg2 8192^2 -> 2048^2 @0 (cold class)", 8192, 8192, 4.0, 1.0, 0.0, 1)
prep("8192x8000 4:1 @0 (same class, warm tune cache)", 8192, 8000, 4.0, 1.0, 0.0, 1)
prep("cfg2 @180 (same class, warm)", 8192, 8192, 4.0, 1.0, 180.0, 1)
prep("cfg4 4096^2 -> 1024^2 @0 (new width class)", 4096, 4096, 4.0, 1.0, 0.0, 1)
prep("cfg1 512^2 -> 256^2 @0", 512, 512, 2.0, 1.0, 0.0, 1)
prep("8192x8191 3:1 @0 (inexact: device scan)", 8192, 8191, 3.0, 1.0, 0.0, 1)
prep("8192^2 5:1 @90", 8192, 8192, 5.0, 1.0, 90.0, 1)
prep("8193^2 6:1 @0 fast, integer isocenter", 8193, 8193, 6.0, 1.0, 0.0, 2)
prep("cfg3 8192^2 -> 3426^2 @17.5", 8192, 8192, 8192.0, 2731.0, 17.5, 1)
prep("cfg3 fast", 8192, 8192, 8192.0, 2731.0, 17.5, 2)
prep("cfg5 4096^2 x4 @45", 4096, 4096, 1.0, 4.0, 45.0, 1)
prep("cfg5 bilinear", 4096, 4096, 1.0, 4.0, 45.0, 3)
prep("refdefault 911^2 150->25.4 dpi @1.5 fast", 911, 911, 150.0, 25.4, 1.5, 2, (455.0, 455.0))
