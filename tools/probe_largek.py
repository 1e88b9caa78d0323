#!/usr/bin/env python3
"""Large k (open-addressing table, buckets of about one position): exact and multi-part batches, all planted."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kmer_index_amd import engine, synth  # noqa: E402

n, sigma = 100_000_000, 4
text = synth.ranks(1002, n, sigma)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for ks in ([20], [16, 24, 31]):
    idx = engine.Index(text, sigma, ks)
    for m, nq in ((ks[0], 10_000_000), (2 * ks[0], 5_000_000), (100, 2_000_000), (150, 2_000_000), (ks[0] - 1, 2_000_000)):
        q, off = synth.mixed_queries(77 + m, text, nq, [m], sigma, planted_frac=0.9)
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
        print(f"ks={ks} m={m:3d} nq={nq}: {dt * 1e3:8.3f} ms/step {nq / dt / 1e6:9.1f} M q/s  hits {c['n_hits']} exact={c['n_exact']} "
              f"stitch={c['n_stitch']} prefix={c['n_prefix']} err={c['n_error']}  {st}", flush=True)
        res.close()
    idx.close()
