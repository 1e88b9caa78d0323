"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The batch search shards naturally: queries are independent, the index is replicated in
every GPU's HBM, rank r searches the contiguous query range [nq*r/W, nq*(r+1)/W).  Concatenating the
shards' outputs in rank order reproduces the single-GPU output byte for byte, so the only exchange
step is collecting results:
  * all_gather_totals : per-rank (n_queries, n_hits) — 16 bytes per rank, what a consumer needs to
                        address the sharded hit lists where they lie (the zero-copy mode bench.py times);
  * HitGather.gather   : the full gatherv of hit_off + positions to one rank — one grouped point-to-point
                        exchange with the displacements from the totals (over xGMI each shard crosses its
                        own direct link, all W-1 links at once), into buffers kept between batches.
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


class HitGather:
    """gatherv of the shards' results to rank `dst`, reusable across batches.

    One call = one small all_gather (per-shard totals: the displacements) + ONE grouped point-to-point
    exchange (`dist.batch_isend_irecv`: ncclGroupStart .. ncclGroupEnd on the nccl backend), in which the
    root posts the receives from all W-1 peers at once, straight into their slices of the gathered
    arrays — over xGMI every peer's shard then crosses its own direct link concurrently (W-1 links
    busy, not one after the other).  The root's buffers are grow-only and kept between calls, so a
    steady-state batch allocates nothing.
    """

    def __init__(self, dst: int = 0, group=None):
        self.dst, self.group = dst, group
        self.g_off = None     # int64 [>= nq_total + 1]
        self.g_pos = None     # positions' dtype [>= hits_total]
        self.last_bytes_per_peer = []   # root only: bytes received from each peer in the last call (0 for itself)

    def _ensure(self, nq_total, hits_total, device, pos_dtype):
        if self.g_off is None or self.g_off.numel() < nq_total + 1 or self.g_off.device != device:
            self.g_off = torch.empty(nq_total + 1 + (nq_total >> 4), dtype=torch.int64, device=device)
        if self.g_pos is None or self.g_pos.numel() < hits_total or self.g_pos.device != device or self.g_pos.dtype != pos_dtype:
            self.g_pos = torch.empty(hits_total + (hits_total >> 4), dtype=pos_dtype, device=device)

    def gather(self, hit_off: torch.Tensor, positions: torch.Tensor, totals=None):
        """hit_off int64 [nq_local + 1] (hit_off[0] == 0), positions [n_hits_local] (n_hits_local == hit_off[-1]).
        `totals` ([world, 2] int64 numpy, from all_gather_totals) may be passed by a caller that already has them.
        Returns (hit_off_global int64 [nq_total + 1], positions_global) on `dst` — views of the kept buffers, valid
        until the next call — and (None, None) elsewhere."""
        group, dst = self.group, self.dst
        rank = dist.get_rank(group)
        world = dist.get_world_size(group)
        device = hit_off.device
        nq_local = hit_off.numel() - 1
        if totals is None:
            totals = all_gather_totals(nq_local, positions.numel(), device=device if device.type == "cuda" else None, group=group)
        if rank != dst:
            ops = []
            peer = dist.get_global_rank(group, dst) if group is not None else dst
            if nq_local:
                ops.append(dist.P2POp(dist.isend, hit_off[1:], peer, group))
            if positions.numel():
                ops.append(dist.P2POp(dist.isend, positions, peer, group))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            return None, None
        nq_total = int(totals[:, 0].sum())
        hits_total = int(totals[:, 1].sum())
        self._ensure(nq_total, hits_total, device, positions.dtype)
        g_off, g_pos = self.g_off[:nq_total + 1], self.g_pos[:hits_total]
        g_off[0] = 0
        ops, rebase = [], []
        self.last_bytes_per_peer = [0] * world
        q0, h0 = 0, 0
        for r in range(world):
            nq_r, nh_r = int(totals[r, 0]), int(totals[r, 1])
            if r == dst:
                torch.add(hit_off[1:], h0, out=g_off[q0 + 1:q0 + 1 + nq_r])
                g_pos[h0:h0 + nh_r].copy_(positions)
            else:
                peer = dist.get_global_rank(group, r) if group is not None else r
                if nq_r:
                    ops.append(dist.P2POp(dist.irecv, g_off[q0 + 1:q0 + 1 + nq_r], peer, group))
                    if h0:
                        rebase.append((q0 + 1, q0 + 1 + nq_r, h0))
                if nh_r:
                    ops.append(dist.P2POp(dist.irecv, g_pos[h0:h0 + nh_r], peer, group))
                self.last_bytes_per_peer[r] = nq_r * 8 + nh_r * g_pos.element_size()
            q0 += nq_r
            h0 += nh_r
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for a, b, h in rebase:                      # the peers sent shard-local offsets
            g_off[a:b] += h
        return g_off, g_pos


def gather_hit_lists(hit_off: torch.Tensor, positions: torch.Tensor, dst: int = 0, group=None):
    """One-shot form of HitGather.gather (fresh buffers)."""
    return HitGather(dst, group).gather(hit_off, positions)


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
