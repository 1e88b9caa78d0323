#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (kmx_search_batch + kmx_result_view) on BASELINE configs[1]:
queries start in host memory, the sorted hit lists end in host memory.  Never `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_index_amd import engine, synth  # noqa: E402


def main():
    n, nq, k = 100_000_000, int(os.environ.get("NQ", 10_000_000)), 10
    text = synth.ranks(1002, n, 4)
    idx = engine.Index(text, 4, [k])
    z = synth.u64_stream(2002, nq * k, 0)
    qr = (((z >> np.uint64(32)) * np.uint64(4)) >> np.uint64(32)).astype(np.uint8)
    qoff = np.arange(nq + 1, dtype=np.uint64) * np.uint64(k)
    res = engine.Result()
    for rep in range(4):
        t0 = time.perf_counter()
        idx.search(qr, qoff, result=res)
        t1 = time.perf_counter()
        off, pos, st, kinds = res.host(copy=False)
        t2 = time.perf_counter()
        hits = int(off[-1])
        print(f"rep {rep}: H2D + search {1e3 * (t1 - t0):8.1f} ms, D2H view {1e3 * (t2 - t1):8.1f} ms "
              f"({hits * 4 / (t2 - t1) / 1e9:5.1f} GB/s), end to end {nq / (t2 - t0) / 1e6:7.1f} M queries/s, hits {hits}", flush=True)
    t0 = time.perf_counter()
    idx.search(qr, qoff, flags=engine.SEARCH_COUNT_ONLY, result=res)
    off = res.host()[0]
    print(f"count-only: {nq / (time.perf_counter() - t0) / 1e6:7.1f} M queries/s end to end")


if __name__ == "__main__":
    main()
