"""The fix-up pass beside the production kernel: launch time with the pass (on the plan's side stream, beside the production
kernel) and without it (the diagnostic request flag AAI_POLICY_DIAG_NO_FIXUP / aai.debug_skip_fixup).
usage: python tools/fixup_ab.py [workload ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import area_average_interpolation_amd as aai
from area_average_interpolation_amd import _lib

lib = _lib.load()
aai.set_device(0)
stream = torch.cuda.current_stream().cuda_stream


def run(name, skip, n=30):
    W, H, sr, dr, ang, mode, desc = bench.WORKLOADS[name]
    lib.aai_shutdown()
    aai.debug_skip_fixup(skip)
    rq = aai.make_request(W, H, sr, dr, bench.isocenter(name, W, H), ang, mode=mode)
    rc, msg, lay = aai.query(rq)
    dW, dH = lay.dst_width, lay.dst_height
    src = torch.empty((H, W), dtype=torch.float32, device="cuda")
    dst = torch.empty((dH, dW), dtype=torch.float32, device="cuda")
    aai.synth_device(src.data_ptr(), W, H, W, 1, stream)
    aai.prepare(rq)
    for _ in range(3):
        aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, stream)
    e1.record()
    torch.cuda.synchronize()
    aai.debug_skip_fixup(False)
    return e0.elapsed_time(e1) / n * 1e3, aai.plan_shape(rq)


for name in (sys.argv[1:] or ["cfg3", "cfg5", "wide8", "cfg3fast"]):
    for rep in range(2):
        a, shape = run(name, False)
        c, _ = run(name, True)
        print("%-9s with the fix-up pass beside the kernel %9.1f us   without it %9.1f us   (%s)" % (name, a, c, shape), flush=True)
