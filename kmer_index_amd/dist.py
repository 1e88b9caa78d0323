"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The batch search shards naturally: queries are independent, the index is replicated in
every GPU's HBM, rank r searches the contiguous query range [nq*r/W, nq*(r+1)/W).  Concatenating the
shards' outputs in rank order reproduces the single-GPU output byte for byte, so the only exchange
step is collecting results:
  * all_gather_totals : per-rank (n_queries, n_hits) — 16 bytes per rank, what a consumer needs to
                        address the sharded hit lists where they lie (the zero-copy mode bench.py times);
  * gather_hit_lists  : the full gatherv of hit_off + positions to one rank (point-to-point
                        send/recv with the displacements from the totals; over xGMI each shard
                        crosses one direct link).
Replicating the index is either "every rank builds it" (0.1 s on the GPU for 1e8 bp — what bench.py
does) or build once + broadcast_index: the flat image of kmx_index_save as one broadcast payload.
"""
import os
import tempfile

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(nq: int, rank: int, world: int):
    """Contiguous query range of `rank`."""
    return nq * rank // world, nq * (rank + 1) // world


def shard_queries(qranks: np.ndarray, qoff: np.ndarray, rank: int, world: int):
    """This rank's queries with offsets rebased to 0."""
    b, e = shard_bounds(len(qoff) - 1, rank, world)
    lo, hi = int(qoff[b]), int(qoff[e])
    return qranks[lo:hi], (qoff[b:e + 1] - qoff[b]).astype(np.uint64)


def all_gather_totals(n_queries: int, n_hits: int, device=None, group=None):
    """Every rank learns every shard's (n_queries, n_hits).  Returns an int64 array [world, 2]."""
    world = dist.get_world_size(group)
    mine = torch.tensor([n_queries, n_hits], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return torch.stack(out).cpu().numpy()


def gather_hit_lists(hit_off: torch.Tensor, positions: torch.Tensor, dst: int = 0, group=None):
    """gatherv of the shards' results to rank `dst`.

    hit_off   : int64 [nq_local + 1], local offsets (hit_off[0] == 0)
    positions : int32/uint32-as-int32 [n_hits_local]
    Returns (hit_off_global int64 [nq_total + 1], positions_global) on `dst`, (None, None) elsewhere.
    """
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    device = hit_off.device
    totals = all_gather_totals(hit_off.numel() - 1, positions.numel(), device=device, group=group)
    if rank != dst:
        if hit_off.numel() > 1:
            dist.send(hit_off[1:].contiguous(), dst=dst, group=group)
        if positions.numel():
            dist.send(positions.contiguous(), dst=dst, group=group)
        return None, None
    nq_total = int(totals[:, 0].sum())
    hits_total = int(totals[:, 1].sum())
    g_off = torch.zeros(nq_total + 1, dtype=torch.int64, device=device)
    g_pos = torch.empty(hits_total, dtype=positions.dtype, device=device)
    q0, h0 = 0, 0
    for r in range(world):
        nq_r, nh_r = int(totals[r, 0]), int(totals[r, 1])
        if r == dst:
            g_off[q0 + 1:q0 + 1 + nq_r] = hit_off[1:] + h0
            g_pos[h0:h0 + nh_r] = positions
        else:
            tmp = torch.empty(nq_r, dtype=torch.int64, device=device)
            if nq_r:
                dist.recv(tmp, src=r, group=group)
            g_off[q0 + 1:q0 + 1 + nq_r] = tmp + h0
            if nh_r:
                dist.recv(g_pos[h0:h0 + nh_r], src=r, group=group)
        q0 += nq_r
        h0 += nh_r
    return g_off, g_pos


def broadcast_bytes(data, src: int = 0, device=None, group=None) -> np.ndarray:
    """One byte string from rank `src` to every rank (length first, then the payload as one broadcast).
    `data` is bytes / a uint8 array on `src` and ignored elsewhere; every rank returns a uint8 array."""
    rank = dist.get_rank(group)
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        payload = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy() if not isinstance(data, np.ndarray) else data)
        n[0] = payload.numel()
    dist.broadcast(n, src=src, group=group)
    if rank != src:
        payload = torch.empty(int(n.item()), dtype=torch.uint8)
    payload = payload.to(device) if device is not None else payload
    if payload.numel():
        dist.broadcast(payload, src=src, group=group)
    return payload.cpu().numpy()


def broadcast_index(index, src: int = 0, device_index: int = -1, device=None, group=None, scratch_dir=None):
    """Build once, replicate everywhere: rank `src` passes its engine.Index, every other
    rank passes None and gets an index loaded from the broadcast image on GPU `device_index`.  `device` is the
    torch device the collective runs on (a cuda device for the nccl backend, None for gloo)."""
    from . import engine
    rank = dist.get_rank(group)
    scratch_dir = scratch_dir or ("/dev/shm" if os.path.isdir("/dev/shm") else None)
    fd, path = tempfile.mkstemp(prefix=f"kmx_bcast_r{rank}_", suffix=".img", dir=scratch_dir)
    os.close(fd)
    try:
        image = None
        if rank == src:
            index.save(path)
            image = np.fromfile(path, dtype=np.uint8)
        image = broadcast_bytes(image, src=src, device=device, group=group)
        if rank == src:
            return index
        image.tofile(path)
        return engine.Index.load(path, device=device_index)
    finally:
        if os.path.exists(path):
            os.remove(path)
