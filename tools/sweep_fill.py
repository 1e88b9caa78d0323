#!/usr/bin/env python3
"""Times the batch pipeline for several k_fill variants on one GPU (same process, interleaved rounds).

    python tools/sweep_fill.py [--n 100000000] [--nq 10000000] [--variants 8,8v,16,...] [--rounds 3]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--nq", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--sigma", type=int, default=4)
    ap.add_argument("--variants", default="8,8v,8vn,16,16v,4v")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--table", default="open")
    ap.add_argument("--query-order", default="random", choices=["random", "sorted"],
                    help="sorted: queries pre-sorted by rank-hash on the host (experiment: upper bound of key-ordered processing)")
    args = ap.parse_args()
    import torch
    from kmer_index_amd import engine, synth
    dev = torch.device("cuda", 0)
    text = synth.ranks(1002, args.n, args.sigma)
    q, off = synth.uniform_queries(2002, args.nq, args.k, args.sigma)
    if args.query_order == "sorted":
        w = (args.sigma ** np.arange(args.k - 1, -1, -1)).astype(np.uint64)
        h = (q.reshape(-1, args.k).astype(np.uint64) * w).sum(axis=1)
        q = q.reshape(-1, args.k)[np.argsort(h, kind="stable")].reshape(-1).copy()
    d_q = torch.from_numpy(q).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    table = {"open": engine.TABLE_OPEN, "dense": engine.TABLE_DENSE}[args.table]
    variants = args.variants.split(",")
    idxs = {}
    for v in variants:
        os.environ["KMX_FILL_VARIANT"] = v
        idxs[v] = engine.Index(text, args.sigma, [args.k], table=table)
    stream = torch.cuda.current_stream().cuda_stream
    res = engine.Result()
    best = {v: [] for v in variants}
    digest = {}
    for r in range(args.rounds):
        for v in variants:
            ix = idxs[v]
            ix.search_device(d_q.data_ptr(), d_off.data_ptr(), args.nq, stream=stream, result=res)   # warm
            torch.cuda.synchronize()
            ix.stats_enable(True)
            ix.stats_reset()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ix.search_device(d_q.data_ptr(), d_off.data_ptr(), args.nq, stream=stream, result=res)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            st = ix.stats()
            ix.stats_enable(False)
            fill = st["k_fill"]["total_ms"] / max(st["k_fill"]["launches"], 1)
            best[v].append((dt, fill))
            if r == 0:
                c = res.counts()
                _, pos_ptr, _ = res.device_ptrs()
                digest[v] = c["n_hits"]
    hits = next(iter(digest.values()))
    print(f"hits/step {hits}  (algorithmic fill bytes {8 * hits / 1e9:.2f} GB)")
    for v in variants:
        dts = sorted(x[0] for x in best[v])
        fl = sorted(x[1] for x in best[v])
        kms = {k: round(x["total_ms"] / max(x["launches"], 1), 4) for k, x in idxs[v].stats().items() if x["launches"]}
        print(f"variant {v:>5}: step ms min {dts[0]:.3f} med {dts[len(dts)//2]:.3f} | k_fill ms min {fl[0]:.3f} med {fl[len(fl)//2]:.3f} "
              f"-> {8 * hits / fl[0] / 1e6:.0f} GB/s algorithmic ({8 * hits / fl[0] / 1e6 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)
    assert len(set(digest.values())) == 1


if __name__ == "__main__":
    main()
