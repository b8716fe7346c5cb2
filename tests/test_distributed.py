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
