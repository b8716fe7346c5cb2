#!/usr/bin/env python3
"""The cold path as the reference's user sees it (one call per process, Source.cpp:1565): a fresh process per request, aai_prepare
timed with AAI_TRACE_PLAN=1 stage lines on stderr (the first line of a rotated plan holds the load of its kernels' code objects),
then the first launch.  usage: python tools/plan_cold.py  (spawns one child per workload)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    import torch
    import bench
    import area_average_interpolation_amd as aai
    name = sys.argv[1]
    W, H, sr, dr, ang, mode, desc = bench.WORKLOADS[name]
    aai.set_device(0)
    torch.zeros(1, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rq = aai.make_request(W, H, sr, dr, bench.isocenter(name, W, H), ang, mode=mode)
    rc, msg, lay = aai.query(rq)
    src = torch.empty((H, W), dtype=torch.float32, device="cuda")
    dst = torch.empty((lay.dst_height, lay.dst_width), dtype=torch.float32, device="cuda")
    aai.synth_device(src.data_ptr(), W, H, W, 1, stream)      # (loads aai_util's code object: not part of the plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    aai.prepare(rq)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), lay.dst_width, stream)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), lay.dst_width, stream)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%-14s aai_prepare %8.2f ms   first launch %8.2f ms   second launch %7.3f ms   %s" % (name, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), aai.plan_shape(rq)), flush=True)
else:
    for name in ("cfg3", "cfg3fast", "cfg5", "refdefault", "refdefaultfast", "wide8", "cfg2", "cfg1"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), name], env=dict(os.environ, AAI_TRACE_PLAN="1"), capture_output=True, text=True)
        sys.stdout.write(p.stdout)
        for l in p.stderr.splitlines():
            if l.startswith("[aai plan]"):
                print("    " + l)
        sys.stdout.flush()
