#!/usr/bin/env python3
"""Interleaved channels (RGB) through the rotated area kernels: aai_resample_interleaved_device against the same channels as
planar images in one batched launch.  Device-resident, HIP events, median of 7.  usage: python tools/interleaved_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import area_average_interpolation_amd as aai

aai.set_device(0)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=7):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    return sorted(t)[n // 2]


for (W, sr, dr, ang, C, dt, mode) in ((8192, 8192.0, 2731.0, 17.5, 3, "f32", 1), (8192, 8192.0, 2731.0, 17.5, 3, "u8", 1), (8192, 8192.0, 2731.0, 17.5, 4, "u8", 1),
                                      (8192, 4.0, 1.0, 0.5, 3, "u8", 1), (8192, 1.0, 1.0, 1.0, 3, "u8", 1), (4096, 1.0, 2.0, 30.0, 3, "f32", 1),
                                      (4096, 1.0, 2.0, 30.0, 3, "u8", 1), (4096, 1.0, 3.0, 45.0, 4, "u8", 1), (8192, 2.0, 1.0, 45.0, 3, "f32", 1),
                                      # fast mode (the reference's default mode)
                                      (8192, 8192.0, 2731.0, 17.5, 1, "f32", 2), (8192, 8192.0, 2731.0, 17.5, 3, "f32", 2), (8192, 8192.0, 2731.0, 17.5, 3, "u8", 2),
                                      (8192, 8192.0, 2731.0, 17.5, 4, "u8", 2), (4096, 1.0, 2.0, 30.0, 3, "u8", 2)):
    H = W
    rq = aai.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang, mode=mode)
    rc, msg, lay = aai.query(rq)
    dW, dH = lay.dst_width, lay.dst_height
    planar = torch.rand((C, H, W), dtype=torch.float32, device="cuda")
    code = aai.DTYPE_F32
    if dt == "u8":
        planar = (planar * 255).to(torch.uint8); code = aai.DTYPE_U8
    inter = planar.permute(1, 2, 0).contiguous()
    outp = torch.empty((C, dH, dW), dtype=torch.float32, device="cuda")
    outi = torch.empty((dH, dW, C), dtype=torch.float32, device="cuda")
    ti = timed(lambda: aai.resample_interleaved_device(rq, C, inter.data_ptr(), W * C, outi.data_ptr(), dW * C, st, src_dtype=code))
    ki = aai.last_kernel()
    tp = timed(lambda: aai.resample_device(rq, planar.data_ptr(), W, outp.data_ptr(), dW, st, batch=C, src_image_stride=W * H, dst_image_stride=dW * dH, src_dtype=code))
    same = bool(torch.equal(outi.permute(2, 0, 1), outp))
    print("%5d^2 %g:%g angle %-5g mode %d C=%d %-3s  interleaved %.3f ms (%s)   %d planar %.3f ms (%s)   identical %s" % (
        W, sr, dr, ang, mode, C, dt, ti, ki, C, tp, aai.last_kernel(), same))
