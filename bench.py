#!/usr/bin/env python3
"""bench.py -- headline benchmark: output Mpixels/s and achieved HBM GB/s of the area-average hot path.

Workload (BASELINE.json metric / configs[1]): 8192x8192 fp32 -> 2048x2048, rotation 0, area-average with
the reference weight policy, isocenter = image centre, synthetic input from the SURVEY Appendix C.1 hash.
A "step" is one pass of the hot path over one batch of `--batch` device-resident images (one batched
launch through the C ABI).  Images of a batch are distinct buffers (default 4 x 256 MiB = 1 GiB per GPU,
four times the 256 MiB Infinity Cache), so every step streams its source from HBM.

Multi-GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`): one process per
GPU, weak scaling -- every rank resamples its own shard of the batch (no data-path collective: the images
are independent; the request block is broadcast once over RCCL before timing).  value = images of all
ranks x output pixels / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                      # device memory, streams, torch.distributed: plumbing only
import torch.distributed as dist

import area_average_interpolation_amd as aai
from area_average_interpolation_amd import distributed as D

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s copy)

# name -> (W, H, srcRes, dstRes, angle, mode, description)
WORKLOADS = {
    "cfg1": (512, 512, 2.0, 1.0, 0.0, aai.MODE_AREA, "512x512 fp32 -> 256x256, rotation 0"),
    "cfg2": (8192, 8192, 4.0, 1.0, 0.0, aai.MODE_AREA, "8192x8192 fp32 -> 2048x2048, rotation 0"),
    "cfg3": (8192, 8192, 8192.0, 2731.0, 17.5, aai.MODE_AREA, "8192x8192 fp32 -> 3426x3426 (ratio 2731/8192), rotation 17.5"),
    "cfg3fast": (8192, 8192, 8192.0, 2731.0, 17.5, aai.MODE_FAST, "8192x8192 fp32 -> 3426x3426, rotation 17.5, fast mode"),
    "cfg4": (4096, 4096, 4.0, 1.0, 0.0, aai.MODE_AREA, "4096x4096 fp32 -> 1024x1024, rotation 0"),
    "cfg5s": (512, 512, 1.0, 4.0, 45.0, aai.MODE_AREA, "512x512 fp32 -> 2896x2896 (x4 up), rotation 45 (1/8 linear scale of cfg5)"),
    "cfg5": (4096, 4096, 1.0, 4.0, 45.0, aai.MODE_AREA, "4096x4096 fp32 -> 23170x23170 (x4 up), rotation 45"),
    "cfg5bilinear": (4096, 4096, 1.0, 4.0, 45.0, aai.MODE_BILINEAR, "4096x4096 fp32 -> 23170x23170, rotation 45, bilinear"),
    "cfg5bicubic": (4096, 4096, 1.0, 4.0, 45.0, aai.MODE_BICUBIC, "4096x4096 fp32 -> 23170x23170, rotation 45, bicubic"),
}


def cpu_baseline(name, budget_s=15.0):
    """CPU baseline on this box's host cores (1 thread: the reference is single-threaded), on a BOUNDED
    sample of the same workload sized from a short probe to take about `budget_s` seconds.

    kind "reference": the unmodified reference (oracle/_ref/libaai_ref.so, built from Source.cpp by
    oracle/Makefile) when that build travelled with the repo; the sample is a full-width band of the
    image for rotation 0 (same ratio, same per-pixel work: the reference's cost per output pixel does not
    depend on the image height) or a square crop for rotated workloads.
    kind "port": otherwise, the CPU oracle (oracle/aai_oracle.c) on a band of output rows of the full image.
    """
    from oracle import pyoracle as po          # checker / baseline only; never on the product path
    import numpy as np
    W, H, sr, dr, ang, mode, _ = WORKLOADS[name]
    omode = {aai.MODE_AREA: po.MODE_EXACT, aai.MODE_FAST: po.MODE_FAST}.get(mode, mode)
    cpu_model = ""
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass

    if po.have_ref():
        def run_crop(w, h):
            src = po.synth_image(w, h, 1).astype(np.float64)
            t0 = time.perf_counter()
            r = po.ref_run(omode, src, sr, dr, ((w - 1) / 2, (h - 1) / 2), ang)
            dt = time.perf_counter() - t0
            assert r.ok, r.msg
            return r.dst.size, dt

        L = sr / dr
        if ang == 0.0:
            unit = max(1, int(round(8 * L)))                 # source rows for 8 output rows
            n, t = run_crop(W, unit)
            rows_src = int(min(H, max(unit, unit * (budget_s / max(t, 1e-6)))))
            n, t = run_crop(W, rows_src)
            sample = "unmodified reference (Source.cpp via oracle/_ref) on a full-width band: %dx%d source rows -> %d output pixels, %.1f s" % (W, rows_src, n, t)
        else:
            n, t = run_crop(256, 256)
            side = int(min(W, max(256, 256 * (budget_s / max(t, 1e-6)) ** 0.5)))
            n, t = run_crop(side, side)
            sample = "unmodified reference (Source.cpp via oracle/_ref) on a %dx%d crop with the same ratio/rotation: %d output pixels, %.1f s" % (side, side, n, t)
        return {"value": n / t / 1e6, "unit": "Mpixels/s (output)", "cores": 1, "kind": "reference",
                "sample": sample, "host_cpus": os.cpu_count(), "cpu_model": cpu_model}

    if not po.have_oracle():
        po.build()
    lib = po._load_oracle()
    lib.aai_oracle_rows.restype = ctypes.c_int
    lib.aai_oracle_rows.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
        [ctypes.c_double] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    src = np.empty((H, W), np.float32)
    lib.aai_oracle_synth_f32(src.ctypes.data, W, H, 1)
    rc, _, lay = aai.query(aai.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang, mode=mode))
    dW, dH = lay.dst_width, lay.dst_height
    err = ctypes.create_string_buffer(256)

    def run(r0, r1):
        out = np.empty((r1 - r0, dW), np.float64)
        t0 = time.perf_counter()
        ok = lib.aai_oracle_rows(omode, 0, src.ctypes.data, 1, W, H, sr, sr, dr, dr, (W - 1) / 2, (H - 1) / 2, ang,
                                 r0, r1, out.ctypes.data, err, 256)
        assert ok, err.value
        return time.perf_counter() - t0

    mid = dH // 2
    probe = run(mid, mid + 2)
    rows = int(max(2, min(dH - mid, budget_s / max(probe / 2, 1e-6))))
    t = run(mid, mid + rows)
    return {"value": rows * dW / t / 1e6, "unit": "Mpixels/s (output)", "cores": 1, "kind": "port",
            "sample": "oracle/aai_oracle.c (double precision, 1 thread) on %d of %d output rows of the same "
                      "workload (rows %d..%d), %.1f s" % (rows, dH, mid, mid + rows - 1, t),
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--custom", default=None, metavar="W,H,srcRes,dstRes,angle[,fast]",
                    help="explore another geometry (not a BASELINE configuration); e.g. 8192,8192,4,1,0.5")
    ap.add_argument("--policy", default="reference", choices=["reference", "exact"])
    ap.add_argument("--src-dtype", default="f32", choices=["f32", "u8", "u16"],
                    help="source element type (the headline metric is f32; u8/u16 exercise the typed entry points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; nccl (= RCCL) is the real one, gloo lets two ranks rehearse on one GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even with one rank (rehearses the RCCL code path on a single GPU)")
    ap.add_argument("--gather", action="store_true", help="also time an RCCL gather of the outputs to rank 0 (reported separately)")
    args = ap.parse_args()

    saved_stdout = None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1 or args.force_dist:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL printf()s its version banner (NCCL_DEBUG=VERSION on these boxes) to stdout when the communicator is created:
        # park fd 1 on stderr until the ONE JSON line is due
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    else:
        local = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)
    cdev = dev if (world == 1 or args.backend == "nccl") else torch.device("cpu")     # where collective payloads live
    aai.set_device(local)

    if args.custom:
        f = args.custom.split(",")
        WORKLOADS["custom"] = (int(f[0]), int(f[1]), float(f[2]), float(f[3]), float(f[4]),
                               aai.MODE_FAST if len(f) > 5 and f[5] == "fast" else aai.MODE_AREA,
                               "custom %sx%s fp32, resolution %s -> %s, rotation %s" % tuple(f[:5]))
        args.workload = "custom"
    W, H, sr, dr, ang, mode, desc = WORKLOADS[args.workload]
    policy = aai.POLICY_REFERENCE if args.policy == "reference" else aai.POLICY_EXACT
    rq = aai.make_request(W, H, sr, dr, ((W - 1) / 2, (H - 1) / 2), ang, mode=mode, policy=policy)
    rq = D.broadcast_request(rq, src=0, device=cdev, force=args.force_dist)           # the only collective the path needs
    rc, msg, lay = aai.query(rq)
    assert rc == 0, msg
    dW, dH = lay.dst_width, lay.dst_height

    B = args.batch
    total_images = B * world
    first, last = D.shard_bounds(total_images, rank, world)
    assert last - first == B
    src = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    dst = torch.empty((B, dH, dW), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for b in range(B):
        aai.synth_device(src[b].data_ptr(), W, H, W, first + b + 1, stream)       # image g uses seed g+1
    esz, dcode = {"f32": (4, aai.DTYPE_F32), "u8": (1, aai.DTYPE_U8), "u16": (2, aai.DTYPE_U16)}[args.src_dtype]
    if args.src_dtype == "u8":
        src = (src * 256.0).to(torch.uint8)
    elif args.src_dtype == "u16":
        src = (src * 65536.0).to(torch.int32).to(torch.int16)                       # bit pattern of the uint16 value

    def step():
        aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, stream, batch=B,
                            src_image_stride=W * H, dst_image_stride=dW * dH, src_dtype=dcode)

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps          # HIP events on the launch stream
    kernel_name = aai.last_kernel()

    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    gather_ms = None
    if args.gather and dist.is_initialized():
        fence()
        g0 = time.perf_counter()
        payload = dst if cdev == dev else dst.cpu()
        parts = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
        dist.gather(payload, gather_list=parts, dst=0)
        fence()
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        out_pix = total_images * dW * dH * args.steps
        alg_bytes = B * (esz * W * H + 4 * dW * dH)                           # per launch, per GPU
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None                                                        # measured HBM bytes per launch (rocprofv3 PMC)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload)
            if pmc and pmc["batch"] == B and args.src_dtype == "f32":
                traffic = pmc["bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "Mpixels/s (output) and achieved HBM GB/s, 8192^2->2048^2 fp32, 1/2/4/8 GPU",
            "value": out_pix / elapsed / 1e6,
            "unit": "Mpixels/s (output)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if args.src_dtype == "f32" else "synthetic (%s source)" % args.src_dtype,
            "config": {"workload": "%s: %s, area-average, %s weight policy, isocenter = image centre" % (args.workload, desc, args.policy)
                       if mode == aai.MODE_AREA else "%s: %s" % (args.workload, desc),
                       "images_per_gpu_per_step": B, "src_bytes_per_gpu": esz * W * H * B, "parallelism": "batch-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms_per_launch": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "src_gbps_input_rate": esz * W * H * B / (kernel_ms * 1e-3) / 1e9,
        }
        if gather_ms is not None:
            line["gather_ms"] = gather_ms
        if world == 1 and not args.no_cpu_baseline and mode in (aai.MODE_AREA, aai.MODE_FAST):
            line["cpu_baseline"] = cpu_baseline(args.workload)
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        if saved_stdout is not None:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)          # whatever C stdio still holds goes to stderr, not after the JSON line
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
