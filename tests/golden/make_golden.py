#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the authoring container:

    python tests/golden/make_golden.py

Sources of the expected values
  fast_pow.json, bitset.json : the REAL reference headers (fast_pow.hpp, compressed_bitset.hpp) compiled from
                               /root/reference into oracle/_ref/libref.so (oracle/ref_shim.cpp) — reference outputs.
  planner.json               : the thesis' known-answer table (thesis/content/03_measuring_performance.tex:109-128),
                               typed in by hand, plus the restated planner's tables for four k-sets (cross-checked
                               against the thesis rows at generation time).
  search_*.npz               : the CPU restatement (oracle/oracle.cpp, intended mode) — every list is additionally
                               checked against the naive text scan here, so a fixture is never written from an
                               oracle that disagrees with ground truth.  The reference itself cannot produce
                               these (kmer_index.hpp needs seqan3 + robin_hood, absent here): "parity unpinned"
                               against reference outputs, pinned against exact occurrences.
Fixtures are data only: seeds, sizes, expected counts / digests / position lists.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from kmer_index_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402
from tests.helpers import digest, make_queries  # noqa: E402


def gen_fast_pow(R):
    bases = [0, 1, 2, 3, 4, 5, 7, 15, 16, 20, 27, 255, 256, 65537, (1 << 32) + 1]
    exps = list(range(0, 70)) + [100, 127, 128, 200, 255]
    rows = [[b, e, int(R.ref_fast_pow(b, e))] for b in bases for e in exps]
    json.dump({"source": "reference fast_pow.hpp via oracle/_ref", "rows": rows}, open(os.path.join(HERE, "fast_pow.json"), "w"))
    return len(rows)


def gen_bitset():
    cases = []
    z = synth.u64_stream(4242, 4000)
    zi = 0
    for n_bits in [0, 1, 5, 63, 64, 65, 127, 128, 129, 200, 1000]:
        for fill in (0, 1):
            ops = []
            for _ in range(min(40, 3 * n_bits)):
                ops.append((int(z[zi] % np.uint64(max(n_bits, 1))), int(z[zi + 1] & np.uint64(1))))
                zi += 2
            got = orc.bitset_words(n_bits, fill, ops, which="ref") if n_bits else orc.bitset_words(n_bits, fill, [], which="ref")
            words, ones = got
            cases.append({"n_bits": n_bits, "fill": fill, "ops": ops if n_bits else [], "words": [int(w) for w in words], "ones": ones})
    # out-of-range behaviour (std::out_of_range, compressed_bitset.hpp:46,56,66)
    for n_bits, idx in [(0, 0), (10, 10), (64, 64), (64, 1000)]:
        got = orc.bitset_words(n_bits, 1, [(idx, 0)], which="ref")
        cases.append({"n_bits": n_bits, "fill": 1, "ops": [(idx, 0)], "words": None, "ones": None, "out_of_range": got is None})
    json.dump({"source": "reference compressed_bitset.hpp via oracle/_ref", "cases": cases}, open(os.path.join(HERE, "bitset.json"), "w"))
    return len(cases)


def gen_planner():
    thesis = {"ks": [9, 11, 13, 17], "rows": {"29": [9, 9, 11], "30": [13, 17], "31": [9, 9, 13], "32": None, "33": [9, 11, 13]}}
    tables = {}
    for ks in ([5], [10], [8, 10, 12], [9, 11, 13, 17], [3, 4, 5], [31]):
        multi, nk = orc.plan(ks)
        tables[",".join(map(str, ks))] = {"multi_true": np.nonzero(multi)[0][:64].tolist(), "n_multi": int(multi.sum()),
                                          "nk_first_64": nk[:64], "digest": int(sum((q + 1) * (i + 1) * k for q in range(len(nk)) for i, k in enumerate(nk[q])) % (1 << 61))}
    multi, nk = orc.plan(thesis["ks"])
    for q, want in thesis["rows"].items():
        q = int(q)
        if want is None:
            assert not multi[q] and len(nk[q]) == 1
        else:
            assert multi[q] and nk[q] == want, (q, nk[q], want)
    json.dump({"thesis": thesis, "tables": tables}, open(os.path.join(HERE, "planner.json"), "w"))


CONFIGS = {
    # name: (sigma, n, ks, text_seed, query_seed, kind, arg)
    "cfg1_dna4_k5": (4, 100_000, [5], 1001, 2001, "uniform", (10_000, 5)),          # BASELINE configs[0], full size
    "cfg2_dna4_k10_small": (4, 400_000, [10], 1002, 2002, "uniform", (20_000, 10)),
    "cfg3_dna4_multi_small": (4, 400_000, [8, 10, 12], 1003, 2003, "mixed", (6_000, [8, 10, 12, 20, 22, 24])),
    "cfg4_dna5_k10_small": (5, 400_000, [10], 1004, 2004, "uniform", (20_000, 10)),
    "cfg5_aa20_k5_small": (20, 300_000, [5], 1005, 2005, "mixed", (10_000, [5])),
    "envelope_dna4_multi": (4, 200_000, [8, 10, 12], 1006, 2006, "lengths", (list(range(1, 30)) + [33, 35], 12)),
    "envelope_dna4_k5": (4, 100_000, [5], 1007, 2007, "lengths", (list(range(1, 16)) + [20], 12)),
}


def make_inputs(cfg):
    sigma, n, ks, ts, qs, kind, arg = cfg
    text = synth.ranks(ts, n, sigma)
    if kind == "uniform":
        q, off = synth.uniform_queries(qs, arg[0], arg[1], sigma)
    elif kind == "mixed":
        q, off = synth.mixed_queries(qs, text, arg[0], arg[1], sigma)
    else:
        q, off = make_queries(text, sigma, arg[0], arg[1], seed=qs)
    return text, q, off


def gen_search():
    for name, cfg in CONFIGS.items():
        sigma, n, ks = cfg[0], cfg[1], cfg[2]
        text, q, off = make_inputs(cfg)
        oidx = orc.Index(text, sigma, ks)
        h_off, pos, status, _ = oidx.search_batch(q, off, n_threads=8)
        n_off, n_pos = orc.naive_batch(text, q, off)
        ok = status == 0
        # ground truth check of every accepted query
        cnt = np.diff(h_off)
        assert np.array_equal(cnt[ok], np.diff(n_off)[ok]), name
        for i in np.nonzero(ok)[0]:
            assert np.array_equal(pos[int(h_off[i]):int(h_off[i + 1])], n_pos[int(n_off[i]):int(n_off[i + 1])]), (name, i)
        nfull = min(64, off.size - 1)
        while nfull > 1 and h_off[nfull] > 20000:   # keep the committed lists small
            nfull -= 1
        np.savez_compressed(os.path.join(HERE, f"search_{name}.npz"),
                            counts=cnt.astype(np.uint32), status=status.astype(np.uint8), digest=np.array([digest(h_off, pos)], np.uint64),
                            first_lists=pos[:int(h_off[nfull])], first_off=h_off[:nfull + 1],
                            input_digest=np.array([int(np.sum(q.astype(np.uint64) * (np.arange(q.size, dtype=np.uint64) % np.uint64(251) + np.uint64(1))))], np.uint64))
        print(name, "queries", off.size - 1, "hits", int(h_off[-1]), "errors", int((~ok).sum()))


if __name__ == "__main__":
    orc.build(ref=True)
    R = orc.ref_lib()
    assert R is not None, "oracle/_ref not built (needs /root/reference)"
    print("fast_pow rows", gen_fast_pow(R))
    print("bitset cases", gen_bitset())
    gen_planner()
    gen_search()
