#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (aai_resample_f32: hipMalloc + H2D + kernel + D2H + hipFree
per call) on BASELINE config 2, for DESIGN.md section 5.  This is NOT bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import area_average_interpolation_amd as aai
aai.set_device(0)
W = H = 8192
rng = np.random.default_rng(0)
src = rng.random((H, W), dtype=np.float32)
for it in range(4):
    t0 = time.perf_counter()
    rc, msg, dst, iso, lay = aai.resample_host(src, 4, 1, ((W - 1) / 2, (H - 1) / 2), 0.0)
    dt = time.perf_counter() - t0
    assert rc == 0, msg
    print("call %d: %.2f ms  -> %.0f output Mpix/s, %.1f GB/s of source over PCIe (pageable host memory)" % (
        it, dt * 1e3, dst.size / dt / 1e6, src.nbytes / dt / 1e9))
