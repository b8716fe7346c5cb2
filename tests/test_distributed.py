"""CPU, world_size 2 over gloo: the batch-sharding driver (parameter broadcast, shard bounds, optional
gather).  The per-rank compute is injected; here it is the CPU oracle standing in for the GPU kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import area_average_interpolation_amd as aai
    from area_average_interpolation_amd import distributed as D
    from oracle import pyoracle as po

    good = aai.make_request(40, 32, 4, 1, (19.5, 15.5), 0.0)
    mine = good if rank == 0 else aai.make_request(1, 1, 1, 1, (0, 0), 99.0)     # non-root holds junk

    def compute(rq, first, last):
        outs = []
        for b in range(first, last):
            src = po.synth_image(rq.src_width, rq.src_height, b + 1).astype(np.float64)
            r = po.oracle_run(rq.mode, src, rq.src_res_x, rq.dst_res_x, (rq.src_iso_x, rq.src_iso_y), rq.rotation_deg)
            outs.append(torch.from_numpy(r.dst.astype(np.float32)))
        return torch.stack(outs) if outs else torch.zeros((0, 8, 10))

    local, full = D.run_sharded(mine, batch, compute, gather=True)
    a, b = D.shard_bounds(batch, rank, world)
    assert local.shape[0] == b - a
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 5])
def test_two_rank_sharding_equals_single_process(po, aai, batch):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert full.shape == (batch, 8, 10)
    for b in range(batch):
        src = po.synth_image(40, 32, b + 1).astype(np.float64)
        gold = po.oracle_run(po.MODE_EXACT, src, 4, 1, (19.5, 15.5), 0.0).dst.astype(np.float32)
        assert np.array_equal(full[b], gold), b       # concatenation of shards == single-process result, bit-exact


def test_shard_bounds_partition():
    from area_average_interpolation_amd.distributed import shard_bounds
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert [shard_bounds(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]


def test_shard_rows_partition_and_band_footprints(aai):
    """Row-band sharding of one image: aligned contiguous bands that cover every row,
    and source footprints (aai_band_source_rows, host only) that are in range and -- for an un-rotated
    down-sampling -- about 1/world of the image each."""
    from area_average_interpolation_amd.distributed import shard_rows
    for n in (1, 15, 16, 17, 2048, 3426):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert all(a % 16 == 0 for a, b in spans if a < b)
    rq = aai.make_request(8192, 8192, 4, 1, (4095.5, 4095.5), 0.0)
    for r in range(8):
        r0, r1 = shard_rows(2048, r, 8)
        a, b = aai.band_source_rows(rq, r0, r1)
        assert 0 <= a < b <= 8192 and 1020 <= b - a <= 1030        # 256 dst rows x 4 source rows; the last band loses the 2 rows past the image
    rq = aai.make_request(8192, 8192, 8192, 2731, (4095.5, 4095.5), 17.5)
    for r in range(8):
        r0, r1 = shard_rows(3426, r, 8)
        a, b = aai.band_source_rows(rq, r0, r1)
        assert 0 <= a < b <= 8192 and b - a < 8192 * 0.6            # a rotated band still reads a bounded strip


# ---- bench.py launches its own ranks ------------------------------------------------------------------------------
def _run_bench(*extra, timeout=240):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mock-device", "--no-cpu-baseline", "--steps", "3", "--warmup", "1",
                        "--min-seconds", "0.05", *extra], capture_output=True, text=True, env=env, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stdout, p.stderr


def test_bench_self_launches_two_ranks_and_relays_one_line():
    """`python bench.py --gpus 2` without torch.distributed.run: the parent starts the ranks itself (fresh processes with
    RANK / WORLD_SIZE / MASTER_* set), rank 0's single JSON line comes back through it and carries what the world looked
    like from inside the collective backend.  The device step is mocked: this rehearses spawn, rendezvous (gloo),
    request broadcast, barrier-fenced timing, max-over-ranks and per-rank reporting on the CPU."""
    rc, lines, out, err = _run_bench("--gpus", "2", "--backend", "gloo")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and out.strip().count("\n") == 0, out          # ONE line on stdout, nothing else
    d = lines[0]
    assert d["n_gpus"] == 2 and d["mock"] is True and d["scaling"] == "weak"
    assert d["distributed"]["backend"] == "gloo" and d["distributed"]["world_size_seen"] == 2
    assert len(d["distributed"]["kernel_ms_per_rank"]) == 2
    assert d["timed"]["steps_timed"] == d["timed"]["repeats"] * d["steps"] and d["timed"]["repeats"] >= 1
    assert d["timed"]["step_ms"]["min"] <= d["timed"]["step_ms"]["median"] <= d["timed"]["step_ms"]["max"]
    assert abs(d["ms_per_step"] * d["timed"]["steps_timed"] * 1e-3 - d["timed"]["seconds"]) < 1e-6
    # BASELINE config 4 (64 images over the ranks) rides along on a multi-rank headline run: 32 images per rank here
    assert len(d["cfg4"]["ms_per_rank"]) == 2 and "32 images per rank" in d["cfg4"]["workload"] and d["cfg4"]["mpix_s"] > 0
    assert len(out.strip()) < 8000                                         # the driver keeps an 8 KB tail of stdout
    # row bands of one image: same launch path, strong scaling, half an image per rank
    rc, lines, out, err = _run_bench("--gpus", "2", "--backend", "gloo", "--shard", "rows")
    assert rc == 0 and len(lines) == 1, err[-2000:]
    assert lines[0]["scaling"] == "strong" and lines[0]["config"]["images_per_gpu_per_step"] == 0.5
    # one rank keeps the old shape (no process group, no spawn)
    rc, lines, out, err = _run_bench()
    assert rc == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["distributed"]["backend"] is None


def test_bench_parent_fails_when_a_rank_fails():
    rc, lines, out, err = _run_bench("--gpus", "2", "--backend", "gloo", "--mock-fail-rank", "1", timeout=400)
    assert rc != 0 and not lines


# ---- on the GPU box: two ranks drive the HIP path ------------------------------------------------------------------
@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_drive_the_hip_path():
    """Two self-launched ranks share the one GPU of the test box (gloo for the collectives, the real HIP kernels for the
    step): batch sharding and row-band sharding both produce their single line, with the kernel that served them."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    for extra, kernel, scaling in ((["--workload", "cfg1"], "aai_axis_kernel", "weak"),
                                   (["--workload", "cfg1", "--shard", "rows"], "aai_axis_kernel", "strong"),
                                   (["--custom", "4096,4096,3,1,17.5", "--shard", "rows", "--gather"], "aai_cell_kernel", "strong")):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu-baseline",
                            "--steps", "3", "--warmup", "1", "--min-seconds", "0.05"] + extra, capture_output=True, text=True, env=env, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert p.returncode == 0 and len(lines) == 1, (extra, p.stderr[-2000:])
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["distributed"]["world_size_seen"] == 2 and d["scaling"] == scaling
        assert kernel in d["roofline"]["kernel"], d["roofline"]["kernel"]
        assert all(ms > 0 for ms in d["distributed"]["kernel_ms_per_rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg2", "cfg3"])
def test_row_bands_of_two_ranks_concatenate_to_the_reference_rows(aai, name):
    """SURVEY.md section 8(f) N2 end to end: the image is split into the dst row bands that `world` ranks would take
    (distributed.shard_rows), every band is computed from a buffer that holds ONLY the source rows it reads (generated
    from the stateless hash, aai_synth_rows_device_f32), and the concatenation is compared with the unmodified
    reference's known answers at full BASELINE size (complete rows, sample grid, zero count, sum)."""
    import numpy as np
    import torch
    from area_average_interpolation_amd.distributed import shard_rows
    from area_average_interpolation_amd import _lib as L
    from conftest import TOL, load_full, rel_err
    assert torch.cuda.is_available()
    aai.set_device(0)
    z, meta = load_full(name)
    W, H = meta["W"], meta["H"]
    rq = aai.make_request(W, H, meta["src_res"], meta["dst_res"], meta["iso"], meta["angle"], mode=1)
    rc, msg, lay = aai.query(rq)
    assert rc == 0, msg
    dW, dH = lay.dst_width, lay.dst_height
    stream = torch.cuda.current_stream().cuda_stream
    align = 1 if lay.kernel in (L.KERNEL_AXIS, L.KERNEL_AXIS_WIDE) else 16
    for world in (2, 8):
        out = torch.full((dH, dW), -1.0, dtype=torch.float32, device="cuda")
        read = 0
        for rank in range(world):
            r0, r1 = shard_rows(dH, rank, world, align=align)
            s0, s1 = aai.band_source_rows(rq, r0, r1)
            band_src = torch.empty((s1 - s0, W), dtype=torch.float32, device="cuda")
            aai.synth_rows_device(band_src.data_ptr(), W, H, s0, s1, W, 1, stream)
            aai.resample_band_device(rq, r0, r1, band_src.data_ptr(), W, out[r0:r1].data_ptr(), dW, stream)
            torch.cuda.synchronize()
            read += s1 - s0
            del band_src
        m = meta["exact"]
        assert rel_err(out[::m["step"], ::m["step"]].cpu().numpy(), z["exact_grid"]).max() <= TOL, (name, world)
        assert rel_err(out[m["rows"], :].cpu().numpy(), z["exact_rows"]).max() <= TOL, (name, world)
        assert int((out == 0).sum().item()) == m["zeros"], (name, world)
        total = float(out.sum(dtype=torch.float64).item())
        assert abs(total - float(m["sum"])) <= 2e-7 * float(m["sum"])
        # bands read their own footprint, not the image: exactly the image once without rotation; a rotated band also
        # reads the rows its slanted ends reach (W sin(theta) of them)
        assert read < (H + 64 * world if name == "cfg2" else H + world * (0.36 * H + 64)), (name, world, read)
