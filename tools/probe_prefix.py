#!/usr/bin/env python3
"""Times PREFIX (m < k) and STITCH-with-rest batches on the full-size index (informational)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kmer_index_amd import engine, synth  # noqa: E402

n, sigma, k = 100_000_000, 4, 10
CASES = ((9, 2_000_000), (8, 1_000_000), (7, 200_000), (6, 50_000), (5, 10_000), (3, 300), (13, 2_000_000), (25, 2_000_000),
         (20, 2_000_000), (30, 2_000_000), (100, 2_000_000), (150, 1_000_000), (1000, 100_000), (5000, 20_000))
if os.environ.get("KMX_PROBE") == "aa20":       # BASELINE configs[4]'s index: 20 letters, k = 5, buckets of about 3
    n, sigma, k = 10_000_000, 20, 5
    CASES = ((4, 1_000_000), (3, 100_000), (2, 5_000), (7, 2_000_000), (12, 2_000_000))
text = synth.ranks(1002, n, sigma)
idx = engine.Index(text, sigma, [k])
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
only = [int(a) for a in sys.argv[1:]]
for m, nq in CASES:
    if only and m not in only:
        continue
    q, off = synth.uniform_queries(77 + m, nq, m, sigma)
    if m > k:   # plant half so that stitches survive
        q, off = synth.mixed_queries(77 + m, text, nq, [m], sigma)
    d_q = torch.from_numpy(q).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    res = engine.Result()
    idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
    torch.cuda.synchronize()
    idx.stats_enable(True)
    idx.stats_reset()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = {k_: round(v["total_ms"] / max(v["launches"], 1), 3) for k_, v in idx.stats().items() if v["launches"]}
    c = res.counts()
    idx.stats_enable(False)
    print(f"m={m:2d} nq={nq}: {dt*1e3:8.3f} ms/step  {nq/dt/1e6:8.1f} M q/s  hits {c['n_hits']}  "
          f"({8*c['n_hits']/dt/1e9:7.1f} GB/s alg.)  kinds prefix={c['n_prefix']} stitch={c['n_stitch']}  {st}", flush=True)
    res.close()
