#!/usr/bin/env python3
"""Throughput of the axis-aligned kernel across ratios / quadrants / modes (algorithmic GB/s)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import area_average_interpolation_amd as aai
aai.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
cases = []
for ang in (0, 90, 180, 270):
    cases.append((8192, 8192, 4.0, 1.0, ang, aai.MODE_AREA))
cases += [(8192, 8192, 8192.0, 2731.0, 0, aai.MODE_AREA), (8192, 8192, 2.0, 1.0, 0, aai.MODE_AREA), (8192, 8192, 3.0, 1.0, 0, aai.MODE_AREA),
          (8192, 8192, 8.0, 1.0, 0, aai.MODE_AREA), (8192, 8192, 16.0, 1.0, 0, aai.MODE_AREA), (8192, 8192, 1.0, 1.0, 0, aai.MODE_AREA),
          (4096, 4096, 1.0, 2.0, 0, aai.MODE_AREA), (4096, 4096, 1.0, 4.0, 0, aai.MODE_AREA), (8192, 8192, 4.0, 1.0, 0, aai.MODE_FAST),
          (8000, 6000, 5.0, 2.0, 90, aai.MODE_AREA), (8191, 8193, 4.0, 1.0, 0, aai.MODE_AREA),
          (8192, 8192, 8.0, 1.0, 90, aai.MODE_AREA), (8192, 8192, 3.0, 1.0, 270, aai.MODE_AREA), (8192, 8192, 2.0, 1.0, 90, aai.MODE_AREA),
          (8192, 8192, 1.0, 1.0, 270, aai.MODE_AREA), (4096, 4096, 1.0, 2.0, 90, aai.MODE_AREA), (8191, 8193, 4.0, 1.0, 90, aai.MODE_AREA)]
for (W, H, sr, dr, ang, mode) in cases:
    rq = aai.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang, mode=mode)
    rc, msg, lay = aai.query(rq)
    src = torch.empty((H, W), dtype=torch.float32, device="cuda")
    aai.synth_device(src.data_ptr(), W, H, W, 1, stream)
    dst = torch.empty((lay.dst_height, lay.dst_width), dtype=torch.float32, device="cuda")
    run = lambda: aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), lay.dst_width, stream)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); run(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    t = sorted(ts)[2]
    alg = 4 * W * H + 4 * lay.dst_width * lay.dst_height
    print("%5dx%-5d ratio %-9.5g angle %-3g mode %d -> %5dx%-5d  %-22s %8.1f us  %7.0f GB/s  %9.0f Mpix/s" % (
        W, H, sr / dr, ang, mode, lay.dst_width, lay.dst_height, aai.last_kernel(), t * 1e3, alg / t / 1e6, lay.dst_width * lay.dst_height / t / 1e3))
    del src, dst
