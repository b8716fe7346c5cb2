#!/usr/bin/env python3
"""bench.py -- headline benchmark: output Mpixels/s and achieved HBM GB/s of the area-average hot path.

Workload (BASELINE.json metric / configs[1]): 8192x8192 fp32 -> 2048x2048, rotation 0, area-average with
the reference weight policy, isocenter = image centre, synthetic input from the SURVEY Appendix C.1 hash.
A "step" is one pass of the hot path over one batch of `--batch` device-resident images (one batched
launch through the C ABI).  Images of a batch are distinct buffers (default 4 x 256 MiB = 1 GiB per GPU,
four times the 256 MiB Infinity Cache), so every step streams its source from HBM.

Timing: W warm-up steps, then the block of K steps is run `repeats` times back to back between two fences
(barrier + device synchronize); `repeats` is chosen from the warm-up so that the timed GPU leg lasts about
`--min-seconds` (6 s by default: a 5-second telemetry sampler then sees the GPU busy); ms_per_step = elapsed /
(repeats * K), taken as the MAX over ranks.  Nothing but the launches runs inside that wall-clock window: ONE pair of
HIP events on the launch stream brackets the whole timed region, and its elapsed time / launches is the average launch
duration behind `roofline` (rocprofv3 --kernel-trace of the same command agrees: profiles/).  Per-step min / median /
max (`timed.step_ms`) come from a separate pass of at most 512 steps with an event pair each.

Default invocation (one GPU, config 2): `roofline.traffic` is measured by this run -- two short child passes under
rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE; MI355X_MICROARCH.md HBM section, FETCH_SIZE x2 on gfx950) started before
this process touches the GPU; where rocprofv3 is missing the figure committed under profiles/ is quoted and says so --
and a `configs` block follows the headline: every BASELINE.json configuration (plus the reference's own default call),
kernel-only, with the plan's launch shape, its cold `prepare_ms` and the unmodified reference timed on this host's cores.

Multi-GPU: `python bench.py --gpus N` starts N worker processes itself (one per GPU; the parent makes no GPU call
and relays rank 0's single JSON line), or run it under `python -m torch.distributed.run --nproc-per-node N`.
Weak scaling over the batch (`--shard batch`, default): every rank resamples its own shard of the batch, no
data-path collective -- the request block is broadcast once over RCCL before timing; `--gather` adds an RCCL
gather of the outputs to rank 0, reported separately.  `--shard rows`: ONE image per step split into dst row
bands over the ranks (strong scaling; each rank generates only the source rows its band reads).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s copy)
MODE_AREA, MODE_FAST, MODE_BILINEAR, MODE_BICUBIC = 1, 2, 3, 4      # include/aai.h

# name -> (W, H, srcRes, dstRes, angle, mode, description)
WORKLOADS = {
    "cfg1": (512, 512, 2.0, 1.0, 0.0, MODE_AREA, "512x512 fp32 -> 256x256, rotation 0"),
    "cfg2": (8192, 8192, 4.0, 1.0, 0.0, MODE_AREA, "8192x8192 fp32 -> 2048x2048, rotation 0"),
    "cfg3": (8192, 8192, 8192.0, 2731.0, 17.5, MODE_AREA, "8192x8192 fp32 -> 3426x3426 (ratio 2731/8192), rotation 17.5"),
    "cfg3fast": (8192, 8192, 8192.0, 2731.0, 17.5, MODE_FAST, "8192x8192 fp32 -> 3426x3426, rotation 17.5, fast mode"),
    "cfg4": (4096, 4096, 4.0, 1.0, 0.0, MODE_AREA, "4096x4096 fp32 -> 1024x1024, rotation 0"),
    "cfg5s": (512, 512, 1.0, 4.0, 45.0, MODE_AREA, "512x512 fp32 -> 2896x2896 (x4 up), rotation 45 (1/8 linear scale of cfg5)"),
    "cfg5": (4096, 4096, 1.0, 4.0, 45.0, MODE_AREA, "4096x4096 fp32 -> 23170x23170 (x4 up), rotation 45"),
    "cfg5fast": (4096, 4096, 1.0, 4.0, 45.0, MODE_FAST, "4096x4096 fp32 -> 23170x23170 (x4 up), rotation 45, fast mode"),
    "cfg5bilinear": (4096, 4096, 1.0, 4.0, 45.0, MODE_BILINEAR, "4096x4096 fp32 -> 23170x23170, rotation 45, bilinear"),
    "cfg5bicubic": (4096, 4096, 1.0, 4.0, 45.0, MODE_BICUBIC, "4096x4096 fp32 -> 23170x23170, rotation 45, bicubic"),
    # the reference's own example call, Source.cpp:1528-1534 (isocenter (455, 455), mode 2 = fast is its default)
    "refdefault": (911, 911, 150.0, 25.4, 1.5, MODE_AREA, "911x911 fp32 at 150 dpi -> 25.4 dpi (158x158), rotation 1.5, isocenter (455,455)"),
    "refdefaultfast": (911, 911, 150.0, 25.4, 1.5, MODE_FAST, "911x911 fp32 at 150 dpi -> 25.4 dpi (158x158), rotation 1.5, isocenter (455,455), fast mode"),
    # not a BASELINE configuration: a wide footprint (the reference's default ratio is 5.9:1; film at 200 dpi -> 25.4 dpi is 8:1) at config 3's angle
    "wide8": (8192, 8192, 8.0, 1.0, 17.5, MODE_AREA, "8192x8192 fp32 -> 1284x1284 (8:1), rotation 17.5"),
    "wide8fast": (8192, 8192, 8.0, 1.0, 17.5, MODE_FAST, "8192x8192 fp32 -> 1284x1284 (8:1), rotation 17.5, fast mode"),
}
ISOCENTER = {"refdefault": (455.0, 455.0), "refdefaultfast": (455.0, 455.0)}      # default: the image centre


def isocenter(name, W, H):
    return ISOCENTER.get(name, ((W - 1) / 2, (H - 1) / 2))


# the `configs` block of the default invocation: (workload, images per launch)
CONFIG_SET = [("cfg1", 4), ("cfg2", 1), ("cfg3", 1), ("cfg3fast", 1), ("cfg4", 64), ("cfg5", 1), ("cfg5fast", 1), ("cfg5bilinear", 1), ("cfg5bicubic", 1),
              ("refdefault", 1), ("refdefaultfast", 1), ("wide8", 1), ("wide8fast", 1)]


# ---- CPU baseline -----------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        return [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return ""


def _ref_band(args):
    """One worker of the multi-process CPU baseline: the unmodified reference on a full-width band of image `seed`."""
    W, rows, sr, dr, ang, omode, seed = args
    import numpy as np
    from oracle import pyoracle as po
    src = po.synth_image(W, rows, seed).astype(np.float64)
    t0 = time.perf_counter()
    r = po.ref_run(omode, src, sr, dr, ((W - 1) / 2, (rows - 1) / 2), ang)
    dt = time.perf_counter() - t0
    assert r.ok, r.msg
    return r.dst.size, dt


def cpu_baseline(name, budget_s=15.0, procs=0, pool=None):
    """CPU baseline on this box's host cores, on a BOUNDED sample of the same workload sized from a short probe to
    take about `budget_s` seconds.

    kind "reference": the unmodified reference (oracle/_ref/libaai_ref.so, built from Source.cpp by
    oracle/Makefile) when that build travelled with the repo; the sample is a full-width band of the
    image for rotation 0 (same ratio, same per-pixel work: the reference's cost per output pixel does not
    depend on the image height) or a square crop for rotated workloads.  1 thread: the reference has none.
    kind "port": otherwise, the CPU oracle (oracle/aai_oracle.c) on a band of output rows of the full image.
    procs > 0 ("N processes over images -- state N"): additionally N processes,
    each running the reference on a band of its own image, reported as `nproc`.
    """
    import ctypes
    import numpy as np
    from oracle import pyoracle as po          # checker / baseline only; never on the product path
    W, H, sr, dr, ang, mode, _ = WORKLOADS[name]
    omode = {MODE_AREA: po.MODE_EXACT, MODE_FAST: po.MODE_FAST}.get(mode, mode)
    cpu_model = _cpu_model()

    if po.have_ref():
        def run_crop(w, h):
            src = po.synth_image(w, h, 1).astype(np.float64)
            t0 = time.perf_counter()
            r = po.ref_run(omode, src, sr, dr, ((w - 1) / 2, (h - 1) / 2), ang)
            dt = time.perf_counter() - t0
            assert r.ok, r.msg
            return r.dst.size, dt

        L = sr / dr
        if name in ISOCENTER:
            src = po.synth_image(W, H, 1).astype(np.float64)
            t0 = time.perf_counter()
            r = po.ref_run(omode, src, sr, dr, ISOCENTER[name], ang)
            t = time.perf_counter() - t0
            assert r.ok, r.msg
            n, rows_src = r.dst.size, H
            sample = "unmodified reference (Source.cpp via oracle/_ref) on the whole %dx%d image: %d output pixels, %.2f s" % (W, H, n, t)
        elif ang == 0.0:
            unit = max(1, int(round(8 * L)))                 # source rows for 8 output rows
            n, t = run_crop(W, unit)
            rows_src = int(min(H, max(unit, unit * (budget_s / max(t, 1e-6)))))
            n, t = run_crop(W, rows_src)
            sample = "unmodified reference (Source.cpp via oracle/_ref) on a full-width band: %dx%d source rows -> %d output pixels, %.1f s" % (W, rows_src, n, t)
        else:
            n, t = run_crop(256, 256)
            side = int(min(W, max(256, 256 * (budget_s / max(t, 1e-6)) ** 0.5)))
            n, t = run_crop(side, side)
            rows_src = side
            sample = "unmodified reference (Source.cpp via oracle/_ref) on a %dx%d crop with the same ratio/rotation: %d output pixels, %.1f s" % (side, side, n, t)
        out = {"value": n / t / 1e6, "unit": "Mpixels/s (output)", "cores": 1, "kind": "reference",
               "sample": sample, "pixels": n, "seconds": t, "host_cpus": os.cpu_count(), "cpu_model": cpu_model}
        if procs > 1 and ang == 0.0 and pool is not None:
            jobs = [(W, rows_src, sr, dr, ang, omode, b + 1) for b in range(procs)]
            t0 = time.perf_counter()
            parts = pool.map(_ref_band, jobs)
            wall = time.perf_counter() - t0
            pix = sum(p[0] for p in parts)
            out["nproc"] = {"processes": procs, "value": pix / max(p[1] for p in parts) / 1e6, "unit": "Mpixels/s (output)",
                            "sample": "%d processes, each the unmodified reference on a %dx%d band of its own image (seed b+1); "
                                      "slowest worker %.1f s, wall %.1f s" % (procs, W, rows_src, max(p[1] for p in parts), wall)}
        return out

    if not po.have_oracle():
        po.build()
    import area_average_interpolation_amd as aai
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
    return {"value": rows * dW / t / 1e6, "unit": "Mpixels/s (output)", "cores": 1, "kind": "port", "pixels": rows * dW, "seconds": t,
            "sample": "oracle/aai_oracle.c (double precision, 1 thread) on %d of %d output rows of the same "
                      "workload (rows %d..%d), %.1f s" % (rows, dH, mid, mid + rows - 1, t),
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model}


# ---- self-launch --------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def under_profiler():
    """rocprofv3 (and rocprof) preload a tool library that initialises the GPU in THIS process before main() runs."""
    env = os.environ
    return any(k in env for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or "rocprof" in env.get("LD_PRELOAD", "")


def spawn_ranks(n, argv, deadline_s):
    """`python bench.py --gpus N` without a launcher: start one fresh worker process per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, like torch.distributed.run sets them) and relay rank 0's single JSON
    line.  This parent never touches the GPU (children are new processes, never an exec of one that has) -- which is why
    it refuses to run under a profiler's preload, where the GPU is already initialised here.  The workers get a session
    of their own: SIGTERM / SIGINT / the overall deadline kill exactly those process groups, so no rank outlives us
    holding a GPU, and a rank stuck in a rendezvous cannot block us for ever."""
    import signal
    if under_profiler():
        sys.stderr.write("bench.py: refusing to self-launch %d ranks under a profiler preload (the GPU is already initialised in this "
                         "process); profile a single rank, or start the ranks with torch.distributed.run\n" % n)
        sys.exit(2)
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)          # exactly the sessions started below
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, frame):
        kill_all()
        sys.exit(128 + signum)

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, AAI_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # wait for all of them; a rank that dies takes the others (blocked in a rendezvous or a collective) with it
    t_end = time.monotonic() + deadline_s
    timed_out = False
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            kill_all()
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > t_end:
            timed_out = True
            kill_all()
            break
        time.sleep(0.05)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    codes = [p.wait() for p in procs]
    for sig, h in old.items():
        signal.signal(sig, h)
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if timed_out or any(codes) or len(lines) != 1:
        sys.stderr.write("bench.py: %sworker exit codes %r, %d JSON line(s) from rank 0\n" %
                         ("deadline of %.0f s passed; " % deadline_s if timed_out else "", codes, len(lines)))
        sys.exit(1)
    print(lines[0], flush=True)
    sys.exit(0)


# ---- HBM traffic of the dominant kernel, measured by this run ---------------------------------------------------------------
def measure_traffic(argv_child, steady, timeout_s=240.0):
    """Two child passes of the same workload under rocprofv3, one counter each (FETCH_SIZE takes 3 of the 4 TCC slots,
    WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), started BEFORE this process touches the GPU.  Returns
    (bytes per launch of the kernel with the most launches, description) or (None, reason).  FETCH_SIZE / WRITE_SIZE are in
    KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced stream (same guide, HBM section): bytes =
    (2 FETCH_SIZE + WRITE_SIZE) * 1024."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if under_profiler():
        return None, "already under a profiler"
    got = {}
    tmp = tempfile.mkdtemp(prefix="aai_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__)] + argv_child
            try:
                r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp", AAI_BENCH_CHILD="1"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                   text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s pass timed out" % counter
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s pass failed (%d): %s" % (counter, r.returncode, (r.stderr or "")[-200:].replace("\n", " "))
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and "aai_" in row.get("Kernel_Name", "") and "synth" not in row["Kernel_Name"]:
                        rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], row.get("Grid_Size"), float(row["Counter_Value"])))
            if not rows:
                return None, "no %s rows in the rocprofv3 output" % counter
            # the steady state: the LAST dispatches of the pass (the child runs `steady` timed launches after its warm-up; a
            # fresh plan's launch-shape measurement comes first and uses other shapes and fewer images)
            rows.sort()
            tail = [r for r in rows[-steady:] if r[1] == rows[-1][1] and r[2] == rows[-1][2]]
            if len(tail) < max(2, steady // 2):
                return None, "the last dispatches of the %s pass do not share one launch shape" % counter
            got[counter] = (tail[-1][1], len(tail), sum(r[3] for r in tail) / len(tail))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    (kf, nf, fetch), (kw, nw, write) = got["FETCH_SIZE"], got["WRITE_SIZE"]
    if kf != kw:
        return None, "the two passes disagree on the dominant kernel"
    nbytes = (2.0 * fetch + write) * 1024.0
    return nbytes, ("measured by this run: rocprofv3 --pmc FETCH_SIZE (mean of %d launches: %.0f KiB, x2 on gfx950) and --pmc WRITE_SIZE (%d launches: %.0f KiB) "
                    "in two child passes of the same workload, kernel %s" % (nf, fetch, nw, write, kf.split("(")[0][-60:]))


# ---- the other BASELINE configurations, kernel-only ------------------------------------------------------------------------
def short_kernel(name):
    """aai_cell_kernel<float, 4, false, false, 64>(...) -> aai_cell_kernel<float,4,false,false,64>"""
    name = name.split("(")[0].strip()
    return name.split("::")[-1].replace(" ", "")[:64]


def compact_shape(shape):
    """the plan's "kernel=K rows=R nt=N swap=S flagged=F dense=D" without the kernel (named beside it)"""
    return " ".join(t for t in shape.split() if not t.startswith("kernel="))[:80]


def config_block(aai, torch, policy, cfg2_cpu, cpu_procs, cpu_pool, with_cpu, cpu_scale=1.0):
    """Every BASELINE.json configuration (CONFIG_SET) on this GPU: device-resident synthetic images, `prepare_ms` = wall
    clock of aai_prepare on a geometry this process has not seen (config 2: the headline already built its plan -- its
    cold figure is `plan.first_call_ms` of the headline), then a block of launches between two HIP events on the launch
    stream.  Algorithmic bytes as in SURVEY.md 8(d): source read once + output written once.  CPU: the unmodified reference
    on a bounded sample of the same workload on this host (cpu_baseline); config 4 additionally as N processes."""
    stream = torch.cuda.current_stream().cuda_stream
    out = []
    for name, B in CONFIG_SET:
        W, H, sr, dr, ang, mode, desc = WORKLOADS[name]
        rq = aai.make_request(W, H, sr, dr, isocenter(name, W, H), ang, mode=mode, policy=policy)
        rc, msg, lay = aai.query(rq)
        assert rc == 0, msg
        dW, dH = lay.dst_width, lay.dst_height
        src = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
        dst = torch.empty((B, dH, dW), dtype=torch.float32, device="cuda")
        for b in range(B):
            aai.synth_device(src[b].data_ptr(), W, H, W, b + 1, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        aai.prepare(rq)
        torch.cuda.synchronize()
        prepare_ms = (time.perf_counter() - t0) * 1e3

        def launch():
            aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, stream, batch=B, src_image_stride=W * H, dst_image_stride=dW * dH)

        launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        launch()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - t0, 1e-5)
        n = int(max(5, min(4000, 0.5 / one)))                      # about 0.5 s of launches per configuration ...
        # ... after as many untimed ones: the CPU leg of the previous configuration left the GPU idle for seconds, and the kernels
        # that are bound by instruction issue run ~12 % slower until its clocks are back up (config 3: 0.189 ms in a cold 0.4 s
        # block, 0.167 ms in the 6 s leg of `--workload cfg3` on the same box)
        for _ in range(n):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        alg = B * (4 * W * H + 4 * dW * dH)
        # compact on purpose: the driver keeps the last 8 KB of stdout, the whole JSON line has to fit (tests/test_host_logic.py);
        # the workload's description is WORKLOADS[name], the host CPU is named once at the top level of the line
        entry = {"name": name, "images": B, "dst": [dW, dH], "kernel": short_kernel(aai.last_kernel()),
                 "ms": round(ms, 5), "launches": n, "mpix_s": round(B * dW * dH / (ms * 1e-3) / 1e6, 1),
                 "gbps": round(alg / (ms * 1e-3) / 1e9, 1), "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                 "prepare_ms": round(prepare_ms, 2), "plan": compact_shape(aai.plan_shape(rq))}
        if name == "cfg3":
            # the same request as a batch of 8 distinct images in ONE launch (2.1 GB of sources: the 256 MB Infinity Cache, which holds a
            # single image across the launches above, does not): ms per launch
            try:
                B8 = 8
                src8 = torch.empty((B8, H, W), dtype=torch.float32, device="cuda")
                dst8 = torch.empty((B8, dH, dW), dtype=torch.float32, device="cuda")
                for b in range(B8):
                    aai.synth_device(src8[b].data_ptr(), W, H, W, b + 1, stream)

                def launch8():
                    aai.resample_device(rq, src8.data_ptr(), W, dst8.data_ptr(), dW, stream, batch=B8, src_image_stride=W * H, dst_image_stride=dW * dH)

                n8 = max(5, n // B8)
                for _ in range(n8):
                    launch8()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(n8):
                    launch8()
                e1.record()
                torch.cuda.synchronize()
                entry["x8_ms"] = round(e0.elapsed_time(e1) / n8, 5)
                del src8, dst8
            except Exception as exc:
                entry["x8_ms"] = "failed: %s" % (str(exc)[:60],)
        if ms < 0.03:
            # a launch-bound configuration: the same launches captured ONCE into a HIP graph (the plan exists: a capture records
            # launches only) and replayed -- what a caller with a fixed pipeline pays per launch
            try:
                per = 64
                g, cs = torch.cuda.CUDAGraph(), torch.cuda.Stream()
                with torch.cuda.graph(g, stream=cs):
                    for _ in range(per):
                        aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, torch.cuda.current_stream().cuda_stream, batch=B,
                                            src_image_stride=W * H, dst_image_stride=dW * dH)
                reps = max(4, n // per)
                for _ in range(reps):
                    g.replay()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                entry["graph_ms"] = round(e0.elapsed_time(e1) / (reps * per), 5)
                del g
            except Exception as exc:        # (reported, not fatal: the launch figures above stand)
                entry["graph_ms"] = "failed: %s" % (str(exc)[:60],)
        del src, dst
        torch.cuda.empty_cache()
        if with_cpu and mode in (MODE_AREA, MODE_FAST):
            if name == "cfg2" and cfg2_cpu:
                cpu = cfg2_cpu
            else:
                cpu = cpu_baseline(name, budget_s=6.0 * cpu_scale, procs=cpu_procs if name == "cfg4" else 0, pool=cpu_pool)
            # [Mpixels/s of the unmodified reference on 1 core, output pixels of the sample, seconds] (+ config 4: N processes)
            entry["cpu"] = [round(cpu["value"], 5), cpu.get("pixels"), round(cpu.get("seconds", 0.0), 2)]
            if "nproc" in cpu:
                entry["cpu_nproc"] = [cpu["nproc"]["processes"], round(cpu["nproc"]["value"], 5)]
            entry["gpu_over_cpu"] = round(B * dW * dH / (ms * 1e-3) / 1e6 / cpu["value"], 0)
            entry["cpu_kind"] = cpu["kind"]
        out.append(entry)
    return out


def cfg4_sharded(aai, torch, dist, D, world, rank, cdev, policy, fence, mock):
    """BASELINE config 4 as BASELINE.json words it -- a batch of 64 independent 4096x4096 -> 1024x1024 images sharded over the
    ranks -- on every rank of a multi-rank run: rank r resamples images [r 64/N, (r + 1) 64/N) (seed = global index + 1), one
    batched launch per step, no data-path collective; kernel-only ms per rank from a HIP event pair, aggregate = all 64 images
    over the slowest rank's time.  Returns the entry on rank 0 (None elsewhere)."""
    W, H, sr, dr, ang, mode, desc = WORKLOADS["cfg4"]
    total = 64
    first, last = D.shard_bounds(total, rank, world)
    B = last - first
    rq = aai.make_request(W, H, sr, dr, isocenter("cfg4", W, H), ang, mode=mode, policy=policy)
    rc, msg, lay = aai.query(rq)
    assert rc == 0, msg
    dW, dH = lay.dst_width, lay.dst_height
    n = 20
    if mock:
        ms = 2.0
    else:
        stream = torch.cuda.current_stream().cuda_stream
        src = torch.empty((max(B, 1), H, W), dtype=torch.float32, device="cuda")
        dst = torch.empty((max(B, 1), dH, dW), dtype=torch.float32, device="cuda")
        for b in range(B):
            aai.synth_device(src[b].data_ptr(), W, H, W, first + b + 1, stream)

        def launch():
            if B:
                aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, stream, batch=B, src_image_stride=W * H, dst_image_stride=dW * dH)

        launch()
        launch()
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            launch()
        e1.record()
        fence()
        ms = e0.elapsed_time(e1) / n
        del src, dst
        torch.cuda.empty_cache()
    mine = torch.tensor([ms], dtype=torch.float64, device=cdev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    rank_ms = [round(float(x.item()), 5) for x in every]
    if rank != 0:
        return None
    alg = (total // world) * (4 * W * H + 4 * dW * dH)
    worst = max(rank_ms)
    return {"workload": "cfg4: 64 x (%s), %d images per rank" % (desc, total // world), "ms_per_rank": rank_ms, "launches": n,
            "mpix_s": round(total * dW * dH / (worst * 1e-3) / 1e6, 1), "frac_per_gpu": round(alg / (worst * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "kernel-only, every rank's images device-resident, no collective on the data path"}


# ---- one rank -----------------------------------------------------------------------------------------------------
def worker(args):
    import ctypes
    import torch                      # device memory, streams, torch.distributed: plumbing only
    import torch.distributed as dist
    import area_average_interpolation_amd as aai
    from area_average_interpolation_amd import distributed as D

    saved_stdout = None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    mock = args.mock_device          # CPU rehearsal of the launch / collective / reporting path (tests): no GPU call at all
    if args.mock_fail_rank == rank:
        sys.exit(3)
    # the worker processes of the multi-process CPU baseline (config 4) are started BEFORE this process touches the GPU
    want_configs = args.configs == "on" or (args.configs == "auto" and world == 1 and args.workload == "cfg2" and not args.custom and
                                            not mock and not args.traffic_child and args.shard == "batch" and args.src_dtype == "f32")
    cpu_procs = args.cpu_procs if args.cpu_procs >= 0 else (min(16, os.cpu_count() or 1) if (args.workload == "cfg4" or want_configs) else 0)
    cpu_pool = None
    cpu_note = None
    if under_profiler() and not args.no_cpu_baseline and not mock:
        # A profiler's preload has initialised the GPU in THIS process: starting worker interpreters from it (each inheriting the
        # preload) is the hop spawn_ranks() and measure_traffic() refuse too.  The CPU legs are skipped and the line says so.
        args.no_cpu_baseline = True
        cpu_note = "skipped: running under a profiler preload (no child processes from a GPU-initialised process)"
    if cpu_procs > 1 and world == 1 and not args.no_cpu_baseline and not mock:
        from multiprocessing import get_context
        cpu_pool = get_context("spawn").Pool(cpu_procs)
    if args.gpus > 1 or world > 1 or args.force_dist:
        assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                   # --force-dist without a launcher: a one-rank group
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        # RCCL printf()s its version banner (NCCL_DEBUG=VERSION on these boxes) to stdout when the communicator is created:
        # park fd 1 on stderr until the ONE JSON line is due
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if not mock:
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        if args.backend == "nccl" and not mock:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    elif not mock:
        local = 0
        torch.cuda.set_device(0)
    backend = dist.get_backend() if dist.is_initialized() else None
    world_seen = dist.get_world_size() if dist.is_initialized() else 1
    dev = torch.device("cpu") if mock else torch.device("cuda", local)
    cdev = dev if (world == 1 or backend == "nccl") else torch.device("cpu")     # where collective payloads live
    if not mock:
        aai.set_device(local)

    if args.custom:
        f = args.custom.split(",")
        WORKLOADS["custom"] = (int(f[0]), int(f[1]), float(f[2]), float(f[3]), float(f[4]),
                               MODE_FAST if len(f) > 5 and f[5] == "fast" else MODE_AREA,
                               "custom %sx%s fp32, resolution %s -> %s, rotation %s" % tuple(f[:5]))
        args.workload = "custom"
    W, H, sr, dr, ang, mode, desc = WORKLOADS[args.workload]
    policy = aai.POLICY_REFERENCE if args.policy == "reference" else aai.POLICY_EXACT
    rq = aai.make_request(W, H, sr, dr, isocenter(args.workload, W, H), ang, mode=mode, policy=policy)
    rq = D.broadcast_request(rq, src=0, device=cdev, force=args.force_dist)           # the only collective the path needs
    rc, msg, lay = aai.query(rq)
    assert rc == 0, msg
    dW, dH = lay.dst_width, lay.dst_height
    esz, dcode = {"f32": (4, aai.DTYPE_F32), "u8": (1, aai.DTYPE_U8), "u16": (2, aai.DTYPE_U16)}[args.src_dtype]
    rows_mode = args.shard == "rows"

    B = 1 if rows_mode else args.batch
    if rows_mode:
        assert args.src_dtype == "f32", "--shard rows runs on fp32 sources"
        # ONE image per step: this rank's band of dst rows and the source rows that band reads (no collective, no halo
        # exchange: the footprint is generated locally from the same stateless hash)
        align = 1 if lay.kernel in (aai._lib.KERNEL_AXIS, aai._lib.KERNEL_AXIS_WIDE) else 16
        r0, r1 = D.shard_rows(dH, rank, world, align=align)
        s0, s1 = aai.band_source_rows(rq, r0, r1) if r1 > r0 else (0, 0)
        total_images = 1
        out_rows = r1 - r0
        src_rows = s1 - s0
    else:
        total_images = B * world
        first, last = D.shard_bounds(total_images, rank, world)
        assert last - first == B
        out_rows, src_rows = dH, H

    if mock:
        stream = 0

        def step():
            time.sleep(0.002)
    else:
        stream = torch.cuda.current_stream().cuda_stream
        if rows_mode:
            src = torch.empty((max(src_rows, 1), W), dtype=torch.float32, device=dev)
            dst = torch.empty((max(out_rows, 1), dW), dtype=torch.float32, device=dev)
            if src_rows:
                aai.synth_rows_device(src.data_ptr(), W, H, s0, s1, W, 1, stream)
                aai.prepare(rq)

            def step():
                if out_rows:
                    aai.resample_band_device(rq, r0, r1, src.data_ptr(), W, dst.data_ptr(), dW, stream)
        else:
            src = torch.empty((B, H, W), dtype=torch.float32, device=dev)
            dst = torch.empty((B, dH, dW), dtype=torch.float32, device=dev)
            for b in range(B):
                aai.synth_device(src[b].data_ptr(), W, H, W, first + b + 1, stream)       # image g uses seed g+1
            if args.src_dtype == "u8":
                src = (src * 256.0).to(torch.uint8)
            elif args.src_dtype == "u16":
                src = (src * 65536.0).to(torch.int32).to(torch.int16)                       # bit pattern of the uint16 value

            def step():
                aai.resample_device(rq, src.data_ptr(), W, dst.data_ptr(), dW, stream, batch=B,
                                    src_image_stride=W * H, dst_image_stride=dW * dH, src_dtype=dcode)

    def fence():
        if not mock:
            torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        if not mock:
            torch.cuda.synchronize()

    # warm-up (also builds the plan), then a short probe that sizes `repeats`
    t_prep = time.perf_counter()
    step()
    fence()
    first_call_ms = (time.perf_counter() - t_prep) * 1e3         # plan creation (scans / tables / launch-shape measurement) + one launch
    for _ in range(max(args.warmup, 1) - 1):
        step()
    fence()
    if args.traffic_child:
        # a counter pass of measure_traffic(): a few launches of the steady shape, nothing to report
        for _ in range(args.steps):
            step()
        fence()
        return
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    probe = max(time.perf_counter() - t0, 1e-6)
    repeats = args.repeats if args.repeats > 0 else max(1, min(1000000, int(args.min_seconds / probe + 0.999)))
    if dist.is_initialized():
        t = torch.tensor([repeats], dtype=torch.int64, device=cdev)
        dist.broadcast(t, src=0)
        repeats = int(t.item())

    # the timed region: launches only; ONE event pair on the launch stream brackets it
    n_timed = repeats * args.steps
    e0 = e1 = None
    if not mock:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    if e0:
        e0.record()
    for _ in range(n_timed):
        step()
    if e1:
        e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / n_timed if e0 else elapsed / n_timed * 1e3        # average launch duration on the stream
    # per-step spread, outside the timed region: at most 512 steps with an event pair each
    n_stat = min(n_timed, 512)
    if mock:
        per_step = [elapsed / n_timed * 1e3] * n_stat
    else:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_stat)]
        for a, b in evs:
            a.record()
            step()
            b.record()
        fence()
        per_step = sorted(a.elapsed_time(b) for a, b in evs)
        del evs
    kernel_name = "mock" if mock else aai.last_kernel()

    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    rank_ms = [kernel_ms]
    if dist.is_initialized():
        mine = torch.tensor([kernel_ms], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) for x in every]

    gather_ms = None
    if args.gather and dist.is_initialized() and not mock:
        fence()
        g0 = time.perf_counter()
        payload = dst if cdev == dev else dst.cpu()
        if rows_mode:
            # bands differ in height: pad to the tallest
            tallest = max(D.shard_rows(dH, r, world, align=align)[1] - D.shard_rows(dH, r, world, align=align)[0] for r in range(world))
            pad = torch.zeros((max(tallest, 1), dW), dtype=payload.dtype, device=payload.device)
            pad[: payload.shape[0]] = payload
            payload = pad
        parts = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
        dist.gather(payload, gather_list=parts, dst=0)
        fence()
        gather_ms = (time.perf_counter() - g0) * 1e3

    # BASELINE config 4 (64 images over the ranks) rides along on every multi-rank batch run of the headline
    cfg4_rank = None
    if world > 1 and dist.is_initialized() and args.configs != "off" and not rows_mode and args.workload == "cfg2" and 64 % world == 0:
        if not mock:
            del src, dst
            torch.cuda.empty_cache()
        cfg4_rank = cfg4_sharded(aai, torch, dist, D, world, rank, cdev, policy, fence, mock)

    if rank == 0:
        out_pix = total_images * dW * dH * n_timed
        alg_bytes = B * (esz * W * src_rows + 4 * dW * out_rows)              # per launch on this GPU
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        plan_shape = "" if mock else aai.plan_shape(rq)                         # "kernel=K rows=R nt=N swap=S flagged=F dense=D"
        traffic, traffic_source = args.traffic_bytes, args.traffic_note          # HBM bytes per launch: this run's rocprofv3 PMC passes
        if traffic is None:
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload)
                if pmc and pmc["batch"] == B and args.src_dtype == "f32" and not rows_mode:
                    traffic = pmc["bytes_per_launch"]
                    traffic_source = "profiles/pmc_traffic.json: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE) committed with the repository, " \
                                     "NOT collected by this run (%s; this run: %s)" % (pmc.get("measured", "see profiles/README.md"), args.traffic_note or "not attempted")
            except Exception:
                pass
        mid = len(per_step) // 2
        line = {
            "metric": "Mpixels/s (output) and achieved HBM GB/s, 8192^2->2048^2 fp32, 1/2/4/8 GPU",
            "value": out_pix / elapsed / 1e6,
            "unit": "Mpixels/s (output)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / n_timed * 1e3,
            "higher_is_better": True, "scaling": "strong" if rows_mode else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if args.src_dtype == "f32" else "synthetic (%s source)" % args.src_dtype,
            "config": {"workload": "%s: %s, area-average, %s weight policy, isocenter = image centre" % (args.workload, desc, args.policy)
                       if mode == MODE_AREA else "%s: %s" % (args.workload, desc),
                       "images_per_gpu_per_step": B if not rows_mode else 1.0 / world, "src_bytes_per_gpu": esz * W * src_rows * B,
                       "parallelism": ("one image in %d row bands" if rows_mode else "batch-sharded x%d") % world},
            "timed": {"repeats": repeats, "steps_timed": n_timed, "seconds": elapsed,
                      "step_ms": {"min": per_step[0], "median": per_step[mid], "max": per_step[-1], "steps": len(per_step),
                                  "note": "separate pass with one HIP event pair per step; the timed region itself holds launches only"}},
            "plan": {"shape": plan_shape, "first_call_ms": first_call_ms,
                     "note": "first_call_ms = plan creation (tables, one-off scans, K1 launch-shape measurement) + one launch, wall clock"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "kernel_ms_per_launch": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "src_gbps_input_rate": esz * W * src_rows * B / (kernel_ms * 1e-3) / 1e9,
            "distributed": {"backend": backend, "world_size_seen": world_seen, "kernel_ms_per_rank": rank_ms,
                            "value_compute_only": out_pix / elapsed / 1e6},
        }
        line["host"] = {"cpu_model": _cpu_model(), "host_cpus": os.cpu_count()}
        if mock:
            line["mock"] = True
        if gather_ms is not None:
            line["distributed"]["gather_ms"] = gather_ms
            line["distributed"]["value_with_gather"] = total_images * dW * dH / (elapsed / n_timed + gather_ms * 1e-3) / 1e6
        if world == 1 and not args.no_cpu_baseline and not mock and mode in (MODE_AREA, MODE_FAST):
            line["cpu_baseline"] = cpu_baseline(args.workload, budget_s=15.0 * args.cpu_scale, procs=cpu_procs if args.workload == "cfg4" else 0, pool=cpu_pool)
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        if cpu_note:
            line["cpu_baseline"] = {"value": None, "unit": "Mpixels/s (output)", "cores": 0, "kind": "reference", "sample": cpu_note}
        if want_configs:
            del src, dst
            torch.cuda.empty_cache()
            line["configs"] = config_block(aai, torch, policy, line.get("cpu_baseline"), cpu_procs, cpu_pool, not args.no_cpu_baseline, args.cpu_scale)
            line["configs_note"] = ("per BASELINE configuration, kernel-only: ms per launch, output Mpixels/s, algorithmic GB/s and fraction of the 8 TB/s HBM peak, "
                                    "cold aai_prepare ms, graph_ms = the same launches replayed from a HIP graph (launch-bound configurations only), x8_ms = 8 distinct images in one launch; cpu = [Mpixels/s of the unmodified reference on 1 core of this host, output pixels of its sample, seconds]")
        if cfg4_rank is not None:
            line["cfg4"] = cfg4_rank
        if saved_stdout is not None:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)          # whatever C stdio still holds goes to stderr, not after the JSON line
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None
        print(json.dumps(line), flush=True)
    if cpu_pool is not None:
        cpu_pool.close()
        cpu_pool.join()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--shard", default="batch", choices=["batch", "rows"],
                    help="batch: every rank resamples its own images (weak scaling); rows: ONE image split into dst row bands (strong scaling)")
    ap.add_argument("--min-seconds", type=float, default=6.0, help="the timed leg repeats the block of --steps until it lasts about this long "
                    "(default 6 s: longer than a 5-second telemetry sampling period)")
    ap.add_argument("--configs", default="auto", choices=["auto", "on", "off"],
                    help="append the per-configuration block (every BASELINE config, kernel-only, with CPU baselines): auto = the default one-GPU config-2 run")
    ap.add_argument("--traffic", default="auto", choices=["auto", "off"],
                    help="measure roofline.traffic in this run (two rocprofv3 --pmc child passes before the GPU is touched): auto = one GPU, no launcher")
    ap.add_argument("--deadline", type=float, default=1500.0, help="--gpus N self-launch: kill the ranks and fail after this many seconds")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)     # a counter pass of measure_traffic()
    ap.add_argument("--repeats", type=int, default=0, help="fix the number of repeats of the --steps block (0 = from --min-seconds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scale", type=float, default=1.0, help="scale the CPU baseline's time budgets (15 s headline, 6 s per configuration); tests use a small one")
    ap.add_argument("--cpu-procs", type=int, default=-1, help="processes of the extra multi-process CPU baseline (default: min(16, cores) for cfg4, else none)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; nccl (= RCCL) is the real one, gloo lets two ranks rehearse on one GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even with one rank (rehearses the RCCL code path on a single GPU)")
    ap.add_argument("--gather", action="store_true", help="also time an RCCL gather of the outputs to rank 0 (reported separately)")
    ap.add_argument("--mock-device", action="store_true", help=argparse.SUPPRESS)      # tests: launch / collective / report path on CPU
    ap.add_argument("--mock-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()

    # No launcher around us and more than one GPU asked for: become the launcher (and never touch the GPU ourselves).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:], args.deadline)
    args.traffic_bytes, args.traffic_note = None, None
    if (args.traffic == "auto" and args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.mock_device and not args.traffic_child and
            not args.force_dist and args.shard == "batch" and os.environ.get("AAI_BENCH_TRAFFIC", "1") != "0"):
        child = ["--traffic-child", "--traffic", "off", "--configs", "off", "--no-cpu-baseline", "--workload", args.workload, "--batch", str(args.batch),
                 "--steps", "8", "--warmup", "2", "--policy", args.policy, "--src-dtype", args.src_dtype] + (["--custom", args.custom] if args.custom else [])
        args.traffic_bytes, args.traffic_note = measure_traffic(child, steady=8)
    worker(args)


if __name__ == "__main__":
    main()
