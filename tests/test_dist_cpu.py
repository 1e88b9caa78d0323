"""N > 1 path on CPU: world_size-2 gloo.  The shards' results (produced here by the CPU oracle, which is
only the data source for the transport test) gathered in rank order must equal the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmer_index_amd import dist as kdist
        from kmer_index_amd import synth
        from oracle import orc
        text = synth.ranks(1004, 60_000, 5)
        ks = [6, 9]
        qranks, qoff = synth.mixed_queries(2004, text, 1001, [4, 6, 9, 12, 15, 18], 5)
        oidx = orc.Index(text, 5, ks)
        my_q, my_off = kdist.shard_queries(qranks, qoff, rank, world)
        h_off, pos, status, _ = oidx.search_batch(my_q, my_off)
        t_off = torch.from_numpy(h_off.astype(np.int64))
        t_pos = torch.from_numpy(pos.view(np.int32).copy())
        totals = kdist.all_gather_totals(len(my_off) - 1, pos.size)
        g_off, g_pos = kdist.gather_hit_lists(t_off, t_pos, dst=0)
        # image transport (broadcast_index's carrier): an odd-sized payload from the last rank, and an empty one
        blob = np.arange(1_000_003, dtype=np.uint64).astype(np.uint8)
        got = kdist.broadcast_bytes(blob.tobytes() if rank == world - 1 else None, src=world - 1)
        assert np.array_equal(got, blob), "broadcast_bytes payload differs"
        assert kdist.broadcast_bytes(b"" if rank == 0 else None, src=0).size == 0
        # a rank without queries takes part in the gather
        e_off, e_pos = kdist.gather_hit_lists(torch.zeros(1 if rank == 1 else 3, dtype=torch.int64) + 0,
                                              torch.zeros(0, dtype=torch.int32), dst=0)
        if rank == 0:
            assert e_off.numel() == 1 + 2 * (world - 1) and e_pos.numel() == 0
        # one HitGather over several batches (bench.py's gather leg rotates two result handles through it): the kept
        # buffers serve a smaller and then a larger batch, every gathered array is exact each time
        gat = kdist.HitGather(dst=0)
        for lo, hi in ((0, 400), (400, 1001), (100, 130)):
            sq, so = qranks[int(qoff[lo]):int(qoff[hi])], (qoff[lo:hi + 1] - qoff[lo]).astype(np.uint64)
            mq, mo = kdist.shard_queries(sq, so, rank, world)
            ho, po, _, _ = oidx.search_batch(mq, mo)
            go, gp = gat.gather(torch.from_numpy(ho.astype(np.int64)), torch.from_numpy(po.view(np.int32).copy()))
            if rank == 0:
                fo, fp, _, _ = oidx.search_batch(sq, so)
                assert np.array_equal(go.numpy().astype(np.uint64), fo) and np.array_equal(gp.numpy().view(np.uint32), fp), (lo, hi)
                assert sum(gat.last_bytes_per_peer) == 8 * (hi - lo - (len(mo) - 1)) + 4 * (fp.size - po.size)
        if rank == 0:
            f_off, f_pos, _, _ = oidx.search_batch(qranks, qoff)
            ok = (np.array_equal(g_off.numpy().astype(np.uint64), f_off) and np.array_equal(g_pos.numpy().view(np.uint32), f_pos)
                  and int(totals[:, 0].sum()) == len(qoff) - 1 and int(totals[:, 1].sum()) == f_pos.size)
            q.put(("ok" if ok else "mismatch", int(f_pos.size)))
    except Exception as e:  # pragma: no cover
        q.put(("error", repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_gathers_to_the_unsharded_result(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, info = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
    assert status == "ok", info
    assert info > 1000


def test_shard_bounds_cover_everything():
    from kmer_index_amd import dist as kdist
    for nq in (0, 1, 7, 1000, 12345):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                b, e = kdist.shard_bounds(nq, r, world)
                assert b == prev and e >= b
                prev = e
            assert prev == nq
