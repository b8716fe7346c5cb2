"""Batch sharding across the GPUs of one node (BASELINE config 4).

A batch of independent images has no exchange step: rank r resamples images
[r*B/N, (r+1)*B/N) on its own GPU.  The only collectives are
  * one broadcast of the ~80-byte parameter block (struct aai_request) from rank 0, so that every rank
    resamples with identical geometry, and
  * an OPTIONAL gather of the outputs to rank 0 (reported separately by bench.py: over xGMI it costs more
    than the compute it gathers).
One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib as L


def shard_bounds(n_items, rank, world):
    """Contiguous block partition: rank r owns [r*n//world, (r+1)*n//world)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (rank * n_items) // world, ((rank + 1) * n_items) // world


def shard_rows(n_rows, rank, world, align=16):
    """Row band of ONE image for rank `rank`: contiguous, starts aligned to `align`
    rows (rotated requests need multiples of 16), the last band takes the remainder.  Returns (row0, row1),
    possibly empty when there are more ranks than aligned bands."""
    blocks = (n_rows + align - 1) // align
    b0, b1 = shard_bounds(blocks, rank, world)
    return min(b0 * align, n_rows), min(b1 * align, n_rows)


def request_to_tensor(request, device="cpu"):
    raw = bytes(ctypes.string_at(ctypes.byref(request), ctypes.sizeof(request)))
    return torch.tensor(list(raw), dtype=torch.uint8, device=device)


def tensor_to_request(t):
    raw = bytes(t.cpu().tolist())
    rq = L.Request()
    ctypes.memmove(ctypes.byref(rq), raw, ctypes.sizeof(rq))
    return rq


def broadcast_request(request, src=0, device="cpu", force=False):
    """Every rank returns rank `src`'s request (pass any placeholder Request on the other ranks).
    force: run the collective even in a one-rank group (rehearsal of the RCCL path on a single GPU)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return request
    t = request_to_tensor(request if dist.get_rank() == src else L.Request(), device=device)
    dist.broadcast(t, src=src)
    return tensor_to_request(t)


def run_sharded(request, batch, compute_shard, gather=False, device="cpu"):
    """Resample `batch` images split over the ranks.

    compute_shard(request, first, last) -> tensor [last-first, dH, dW] on `device` (this rank's images).
    Returns (local outputs, gathered [batch, dH, dW] on rank 0 if gather else None)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    request = broadcast_request(request, device=device)
    first, last = shard_bounds(batch, rank, world)
    local = compute_shard(request, first, last)
    if not gather or world == 1:
        return local, (local if gather else None)
    # shards may differ by one image when world does not divide batch: exchange sizes via all_gather of
    # padded shards
    per = [shard_bounds(batch, r, world) for r in range(world)]
    biggest = max(b - a for a, b in per)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if rank == 0:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, gather_list=parts, dst=0)
        out = torch.cat([parts[r][: per[r][1] - per[r][0]] for r in range(world)], dim=0)
        return local, out
    dist.gather(pad, dst=0)
    return local, None
