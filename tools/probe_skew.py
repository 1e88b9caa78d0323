#!/usr/bin/env python3
"""Repetitive text: a 300-letter unit repeated over 10 % of a 1e8-letter text (mutated at 2 %), reads sampled
uniformly.  Shows what giant candidate buckets cost the STITCH path (query-centric validation) — with the bytes: for the long
reads the ALGORITHMIC bytes of SURVEY 8d (4 bytes x the sum of the buckets of a query's parts, estimated from a sample of the
queries through kmx_index_bucket_host) over the time of the validation kernels, as a fraction of the 8 TB/s peak."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kmer_index_amd import engine, synth  # noqa: E402

n, sigma = 100_000_000, 4
text = synth.ranks(1002, n, sigma)
rng = np.random.default_rng(3)
unit = rng.integers(0, 4, 300).astype(np.uint8)
rep = np.tile(unit, n // 10 // 300)
mut = rng.integers(0, rep.size, rep.size // 50)
rep[mut] = rng.integers(0, 4, mut.size)
text[n // 2:n // 2 + rep.size] = rep
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for ks in ([10], [20]):
    t0 = time.perf_counter()
    idx = engine.Index(text, sigma, ks, keep_host_arena=True)
    print(f"ks={ks}: built in {time.perf_counter() - t0:.2f} s {idx.info()}", flush=True)
    for m, nq in ((ks[0], 2_000_000), (100, 200_000)):
        q, off = synth.mixed_queries(77 + m, text, nq, [m], sigma, planted_frac=1.0)
        d_q = torch.from_numpy(q).to(dev)
        d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        res = engine.Result()
        idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
        torch.cuda.synchronize()
        idx.stats_enable(True)
        idx.stats_reset()
        t0 = time.perf_counter()
        steps = 3
        for _ in range(steps):
            idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = {k_: round(v["total_ms"] / max(v["launches"], 1), 3) for k_, v in idx.stats().items() if v["launches"]}
        c = res.counts()
        idx.stats_enable(False)
        extra = ""
        if m > ks[0] and "k_validate" in st:
            # parts of a single-k query: the full k-parts and, with a rest, the k-mer that ends the query
            k = ks[0]
            starts = [j * k for j in range(m // k)] + ([m - k] if m % k else [])
            sample = range(0, nq, max(1, nq // 4000))
            tot = 0
            for i in sample:
                qi = q[int(off[i]):int(off[i + 1])]
                for s0 in starts:
                    b = idx.bucket_host(k, qi[s0:s0 + k])
                    tot += 0 if b is None else b.size
            bytes_alg = 4.0 * tot / len(sample) * nq
            gbps = bytes_alg / (st["k_validate"] * 1e-3) / 1e9
            extra = f"  validation: {bytes_alg / 1e9:.2f} GB of part buckets (sampled) / {st['k_validate']} ms = {gbps:.0f} GB/s = {gbps / 8000:.3f} of peak"
        print(f"ks={ks} m={m:3d} nq={nq}: {dt * 1e3:9.3f} ms/step {nq / dt / 1e6:9.2f} M q/s  hits {c['n_hits']}  {st}{extra}", flush=True)
        res.close()
    idx.close()
