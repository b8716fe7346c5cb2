#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points on BASELINE config 2's shape, for DESIGN.md section 5:
  (a) aai_resample_f32 per image (hipMalloc + H2D + kernel + D2H + hipFree per call), pageable memory;
  (b) aai_resample_batch_host, 8 images through three device slots / streams, pageable memory;
  (c) the same from page-locked buffers (aai_host_alloc), where uploads, kernels and downloads overlap;
for fp32 and 8-bit sources.  None of this is bench.py's `value` (device-resident data)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import area_average_interpolation_amd as aai
aai.set_device(0)
W = H = 8192
B = 8
iso = ((W - 1) / 2, (H - 1) / 2)
rng = np.random.default_rng(0)
for dt in (np.float32, np.uint8):
    name = np.dtype(dt).name
    srcs = rng.random((B, H, W), dtype=np.float32) if dt == np.float32 else rng.integers(0, 256, size=(B, H, W), dtype=np.uint8)
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        rc, msg, dst, giso, lay = aai.resample_host(srcs[it], 4, 1, iso, 0.0)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0, msg
    print("%-7s single call, pageable : %7.2f ms per image -> %6.0f output Mpix/s, %5.1f GB/s of source over PCIe" % (
        name, best * 1e3, dst.size / best / 1e6, srcs[0].nbytes / best / 1e9))
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        rc, msg, dstb, lay = aai.resample_batch_host(srcs, 4, 1, iso, 0.0)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0, msg
    print("%-7s batch of %d, pageable  : %7.2f ms per image -> %6.0f output Mpix/s, %5.1f GB/s" % (
        name, B, best / B * 1e3, dstb.size / best / 1e6, srcs.nbytes / best / 1e9))
    with aai.PinnedArray(srcs.shape, srcs.dtype) as ps, aai.PinnedArray(dstb.shape, np.float32) as pd:
        ps.array[...] = srcs
        best = 1e9
        for it in range(3):
            t0 = time.perf_counter()
            rc, msg, dstp, lay = aai.resample_batch_host(ps.array, 4, 1, iso, 0.0, out=pd.array)
            best = min(best, time.perf_counter() - t0)
            assert rc == 0, msg
        assert np.array_equal(dstp, dstb)
        print("%-7s batch of %d, pinned    : %7.2f ms per image -> %6.0f output Mpix/s, %5.1f GB/s" % (
            name, B, best / B * 1e3, dstp.size / best / 1e6, srcs.nbytes / best / 1e9))
