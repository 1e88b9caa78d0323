#!/usr/bin/env python3
"""A coarse throughput sweep over alphabets, ks and query lengths — looking for cliffs, not for records: every case is one
index and a batch of random/planted queries of one length; the line says queries/s, hits and which kernels took the time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kmer_index_amd import engine, synth  # noqa: E402

INDEXES = [
    # (name, sigma, n, ks)
    ("dna4 k=10", 4, 20_000_000, [10]),
    ("dna4 k=16 (open)", 4, 20_000_000, [16]),
    ("dna4 k=8,10,12", 4, 20_000_000, [8, 10, 12]),
    ("dna5 k=10", 5, 20_000_000, [10]),
    ("dna15 k=5", 15, 5_000_000, [5]),
    ("aa20 k=5", 20, 10_000_000, [5]),
    ("aa27 k=4,6", 27, 5_000_000, [4, 6]),
    ("binary k=20", 2, 5_000_000, [20]),
]
if os.environ.get("KMX_SWEEP_N"):          # the same indexes over a longer text (buckets grow with it)
    INDEXES = [(a, b, int(os.environ["KMX_SWEEP_N"]), d) for a, b, c, d in INDEXES]
only = sys.argv[1:] or [v for v in os.environ.get("KMX_SWEEP_ONLY", "").split(";") if v]   # (the variable: names with blanks under rocprofv3)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for name, sigma, n, ks in INDEXES:
    if only and not any(o in name for o in only):
        continue
    text = synth.ranks(7, n, sigma)
    idx = engine.Index(text, sigma, ks)
    k = ks[-1]
    lens = sorted({max(1, ks[0] - 2), ks[0] - 1, ks[0], k, k + 1, k + 3, 2 * k, 2 * k + 3, 3 * k + 1, 5 * k + 2, 64, 200})
    if os.environ.get("KMX_SWEEP_LENS"):
        lens = [int(v) for v in os.environ["KMX_SWEEP_LENS"].split(",")]
    for m in lens:
        nq = 500_000 if m <= 64 else 200_000
        q, off = synth.mixed_queries(100 + m, text, nq, [m], sigma)
        d_q = torch.from_numpy(q).to(dev)
        d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        res = engine.Result()
        try:
            idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
            torch.cuda.synchronize()
            idx.stats_enable(True)
            idx.stats_reset()
            t0 = time.perf_counter()
            for _ in range(3):
                idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            st = {k_: round(v["total_ms"] / max(v["launches"], 1), 3) for k_, v in idx.stats().items() if v["launches"]}
            idx.stats_enable(False)
            c = res.counts()
            top = sorted(st.items(), key=lambda kv: -kv[1])[:3]
            flag = "  <<< slow" if nq / dt < 20e6 and c["n_hits"] / dt < 5e9 else ""
            print(f"{name:18s} m={m:3d}: {nq/dt/1e6:9.1f} M q/s  {c['n_hits']/dt/1e9:7.2f} G hits/s  err={c['n_error']}  {top}{flag}", flush=True)
        except engine.KmxError as e:
            print(f"{name:18s} m={m:3d}: {e}", flush=True)
        res.close()
    idx.close()
