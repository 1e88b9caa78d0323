"""N > 1 path with the real engine: two processes share the one GPU of the test box (gloo carries the exchange;
on an N-GPU node the same code runs one rank per GPU over RCCL).  Rank 0 builds the index, the image is
broadcast, every rank searches its contiguous query shard on the device, and the gathered result must equal
the unsharded oracle result byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest

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
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmer_index_amd import dist as kdist
        from kmer_index_amd import engine, synth
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        text = synth.ranks(1003, 400_000, 4)
        ks = [8, 10, 12]
        qranks, qoff = synth.mixed_queries(2003, text, 20_001, [6, 8, 10, 12, 20, 22, 24, 31], 4)
        idx = kdist.broadcast_index(engine.Index(text, 4, ks, device=0) if rank == 0 else None, src=0, device_index=0)
        assert idx.info()["ks"] == ks and idx.info()["n"] == text.size
        my_q, my_off = kdist.shard_queries(qranks, qoff, rank, world)
        d_q = torch.from_numpy(my_q.copy()).to(dev)
        d_off = torch.from_numpy(my_off.view(np.int64).copy()).to(dev)
        res = idx.search_device(d_q.data_ptr(), d_off.data_ptr(), len(my_off) - 1)
        t_off, t_pos = res.device_tensors(dev)
        torch.cuda.synchronize()
        g_off, g_pos = kdist.gather_hit_lists(t_off.cpu(), t_pos.cpu(), dst=0)
        totals = kdist.all_gather_totals(len(my_off) - 1, int(t_pos.numel()))
        if rank == 0:
            from oracle import orc
            f_off, f_pos, _, _ = orc.Index(text, 4, ks).search_batch(qranks, qoff, mode=orc.MODE_INTENDED, n_threads=4)
            ok = (np.array_equal(g_off.numpy().astype(np.uint64), f_off) and np.array_equal(g_pos.numpy().view(np.uint32), f_pos)
                  and int(totals[:, 1].sum()) == f_pos.size)
            q.put(("ok" if ok else "mismatch", int(f_pos.size)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("error", repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_broadcast_image_search_shards_and_gather():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, info = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
    assert status == "ok", info
    assert info > 10_000
