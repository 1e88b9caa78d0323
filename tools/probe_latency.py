#!/usr/bin/env python3
"""Small-batch latency of the host-buffer entry point (kmx_search_batch + kmx_result_view), fresh result per call
(the shape of kmer_index::search(query)) and with a reused result."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_index_amd import engine, synth  # noqa: E402


def main():
    n, k = 10_000_000, 10
    text = synth.ranks(1002, n, 4)
    idx = engine.Index(text, 4, [8, k, 12])
    for nq, lens in [(n_, [8, 10, 12, 20, 9]) for n_ in (1, 16, 256, 4096, 65536)] + [(n_, [8, 10, 12]) for n_ in (256, 1024, 4096, 8192, 16384)]:
        qr, qoff = synth.mixed_queries(77 + nq, text, nq, lens, 4)
        print(f"lengths {lens}:", end=" ")
        for mode in ("fresh", "reused"):
            res = engine.Result()
            idx.search(qr, qoff, result=res).host(copy=False)
            ts = []
            for rep in range(30):
                t0 = time.perf_counter()
                if mode == "fresh":
                    r = idx.search(qr, qoff)
                    r.host(copy=False)
                    r.close()
                else:
                    idx.search(qr, qoff, result=res).host(copy=False)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            print(f"nq={nq:6d} {mode:7s}: median {1e6 * ts[len(ts) // 2]:9.1f} us   min {1e6 * ts[0]:9.1f} us   "
                  f"({nq / ts[len(ts) // 2] / 1e6:8.3f} M queries/s)", flush=True)


if __name__ == "__main__":
    main()
