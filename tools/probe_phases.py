#!/usr/bin/env python3
"""Where a block of k_prefix_sort_block (1024-thread shape) spends its cycles, phase by phase: a MEASUREMENT build of the library
(KMX_PHASE_TIMING=1: thread 0 of every block adds the shader-clock cycles between marks to words of the index's debug block).
Usage on the GPU box: KMX_PHASE_TIMING=1 python tools/probe_phases.py [m]   (rebuilds libkmx.so with the marks, then without)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KMX_PHASE_TIMING"] = "1"
from kmer_index_amd import build  # noqa: E402

build.build(force=True)
import torch  # noqa: E402
from kmer_index_amd import engine, synth  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nq = {6: 50_000, 5: 10_000, 7: 200_000}.get(m, 20_000)
n, sigma, k = 100_000_000, 4, 10
text = synth.ranks(1002, n, sigma)
idx = engine.Index(text, sigma, [k])
q, off = synth.uniform_queries(77 + m, nq, m, sigma)
dev = torch.device("cuda", 0)
d_q = torch.from_numpy(q).to(dev)
d_off = torch.from_numpy(off.view(np.int64)).to(dev)
res = engine.Result()
stream = torch.cuda.current_stream().cuda_stream
idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
res.counts()
torch.cuda.synchronize()
w0 = idx.debug_words().astype(np.int64)
steps = 3
for _ in range(steps):
    idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, stream=stream, result=res)
    res.counts()
torch.cuda.synchronize()
w = (idx.debug_words().astype(np.int64) - w0) / steps
names = {3: "slice header (list, descriptor, run boundaries)", 4: "staging global -> LDS + barrier", 5: "copy-out LDS -> global + barrier", 6: "pair tables",
         7: "pair lookup + merge-path searches", 8: "merge steps", 9: "barrier behind the merge", 10: "write back + sentinels + barrier"}
tot = sum(w[i] for i in names)
print(f"m = {m}, {nq} slices per launch; cycles of thread 0 summed over blocks, per launch: {tot:.3e}")
for i, nm in names.items():
    print(f"  {nm:52s} {w[i]:14.3e}  {100 * w[i] / tot:5.1f} %   {w[i] / nq:9.0f} cycles per slice")
os.environ.pop("KMX_PHASE_TIMING")
build.build(force=True)
