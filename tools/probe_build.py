#!/usr/bin/env python3
"""Index construction time, device vs host flatten, for a few (sigma, ks) on a 1e8-letter text."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_index_amd import engine, synth  # noqa: E402

n = int(float(os.environ.get("N", "1e8")))
for sigma, ks in ((4, [10]), (4, [14]), (4, [15]), (4, [16]), (4, [20]), (4, [31]), (5, [14]), (20, [8])):
    text = synth.ranks(1002, n, sigma)
    row = []
    for host in (False, True):
        if host and os.environ.get("SKIP_HOST"):
            continue
        t0 = time.perf_counter()
        idx = engine.Index(text, sigma, ks, host_flatten=host)
        dt = time.perf_counter() - t0
        row.append(f"{'host' if host else 'device'} {dt:7.2f} s ({idx.info()['device_bytes'] / 1e9:.2f} GB, tables {idx.info()['tables']})")
        idx.close()
    print(f"sigma={sigma} ks={ks} n={n}: " + " | ".join(row), flush=True)
