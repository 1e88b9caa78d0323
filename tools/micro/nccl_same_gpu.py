"""Can RCCL run two ranks on ONE GPU?  (It refuses on this stack: 'Duplicate GPU detected' — which is why the 2-rank rehearsals on
the 1-GPU test box run over gloo.)  Usage: python tools/micro/nccl_same_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda:0") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce over RCCL with two ranks on one GPU -> {t.tolist()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:300]}", flush=True)
        sys.exit(0)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)
