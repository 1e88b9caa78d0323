"""Paths that only run when a buffer cannot be had or a knob says so, exercised in a child process (the knobs are
read once per process): without the survivor buffer k_validate<true> checks further parts in line and k_compact
decodes the mask words (KMX_NO_STITCH_HITS); arenas beyond 4 GiB use 64-bit LDS records in k_fill (KMX_FORCE_REC64);
no line-aligned copy (KMX_ALIGNED=0)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from kmer_index_amd import engine, synth
from oracle import orc
from tests.helpers import pack

rng = np.random.default_rng(7)
motif = rng.integers(0, 4, 53).astype(np.uint8)
text = np.tile(motif, 4000)
mut = rng.integers(0, text.size, text.size // 19)
text[mut] = rng.integers(0, 4, mut.size)
text = np.ascontiguousarray(text)
for ks in ([7], [6, 9, 11], [16]):
    idx = engine.Index(text, 4, ks)
    oidx = orc.Index(text, 4, ks)
    qs = []
    for m in (5, 7, 9, 16, 18, 21, 28, 33, 45, 66, 100):
        for s0 in (0, 13, 53 * 40 + 7, 53 * 2000 + 30, text.size - m):
            qs.append(text[s0:s0 + m].copy())
            qs.append(np.tile(motif, 3)[s0 %% 53:s0 %% 53 + m].copy())
    qranks, qoff = pack(qs)
    o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, mode=orc.MODE_INTENDED, n_threads=4)
    for flags in (engine.SEARCH_DEFAULT, engine.SEARCH_KEEP_MASKS):
        res = engine.Result()
        for rep in range(2):
            ho, pos, st, kd = idx.search(qranks, qoff, flags=flags, result=res).host()
            assert np.array_equal(st, o_st.astype(np.uint8)), (ks, flags, rep)
            assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos), (ks, flags, rep)
    for i in range(0, len(qs), 5):
        if o_st[i] == 0:
            assert np.array_equal(o_pos[int(o_off[i]):int(o_off[i + 1])], orc.naive_scan(text, qs[i]))
print("fallback child ok", int(o_off[-1]))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"KMX_NO_STITCH_HITS": "1"}, {"KMX_FORCE_REC64": "1"}, {"KMX_ALIGNED": "0"},
                                 {"KMX_NO_STITCH_HITS": "1", "KMX_FORCE_REC64": "1"}])
def test_fallback_paths_in_a_child_process(env):
    e = dict(os.environ)
    e.update(env)
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=e)
    assert res.returncode == 0 and "fallback child ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


OOM_CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from kmer_index_amd import engine, synth
from tests.helpers import make_queries

text = synth.ranks(1003, 300_000, 4)
idx = engine.Index(text, 4, [8, 10, 12])
q, off = make_queries(text, 4, [8, 10, 12, 13, 20, 22], 2000, seed=5)
nq = off.size - 1
res = idx.search(q, off)                       # KMX_HOST_CHUNK makes this a chunk-streamed search; one chunk search reports out of memory
got = res.host()
n_parts = res.n_parts()
res.close()
print("oom child ok", n_parts, int(got[0][nq]), int(np.bitwise_xor.reduce(got[1].astype(np.uint64) * np.arange(1, got[1].size + 1, dtype=np.uint64))))
"""


@pytest.mark.gpu
def test_chunked_search_survives_an_out_of_memory_with_one_worker():
    """after an out-of-memory in a chunk-streamed host batch the retry runs with ONE set of device buffers at the
    SAME chunk size (it used to come back to the freed second worker and halve down to failure).  The allocation failure is
    injected (KMX_TEST_INJECT_CHUNK_OOM = the n-th chunk search); the result must equal the uninjected one's, chunk for chunk."""
    outs = []
    for inject in (None, "2", "3"):
        e = dict(os.environ)
        e["KMX_HOST_CHUNK"] = "3000"
        if inject:
            e["KMX_TEST_INJECT_CHUNK_OOM"] = inject
        res = subprocess.run([sys.executable, "-c", OOM_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=e)
        assert res.returncode == 0 and "oom child ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
        outs.append(res.stdout.strip().splitlines()[-1].split()[3:])
    assert outs[0] == outs[1] == outs[2], outs           # same number of chunks (no halving), same hits


SUBK_CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from kmer_index_amd import engine, synth
from oracle import orc
from tests.helpers import pack

text = synth.ranks(606, 900_000, 4)
for ks in ([10], [6]):
    idx = engine.Index(text, 4, ks, prefix_levels=-1)
    oidx = orc.Index(text, 4, ks)
    k = ks[0]
    # k = 10: m = 2 .. 5 -> 56 K .. 879 positions in 65536 .. 1024 runs; k = 6: m = 1 .. 3 -> 225 K .. 14 K positions in 1024 .. 64 runs
    lens = (2, 3, 4, 5) if k == 10 else (1, 2, 3)
    qs = [text[s0:s0 + m].copy() for m in lens for s0 in (1000, 5003, 70001)]
    qs += [text[text.size - m:].copy() for m in lens]
    qranks, qoff = pack(qs)
    idx.stats_enable(True)
    res = engine.Result()
    for rep in range(2):
        ho, pos, st, kd = idx.search(qranks, qoff, result=res).host()
        o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=4)
        assert np.array_equal(st, o_st.astype(np.uint8)), (ks, rep)
        assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos), (ks, rep)
    st = idx.stats()
    print("stats", ks, {n: v["launches"] for n, v in st.items() if n.startswith("k_prefix") and v["launches"]})
    idx.close()
print("subk child ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env,ran,not_ran", [({}, ("k_prefix_split", "k_prefix_bands"), ()),
                                             ({"KMX_NO_SPLIT": "1"}, ("k_prefix_bands", "k_prefix_merge_pass"), ("k_prefix_split",)),
                                             ({"KMX_NO_BANDS": "1"}, ("k_prefix_merge_pass",), ("k_prefix_split", "k_prefix_bands"))])
def test_subk_slices_beyond_a_chunk_without_bands_or_splits(env, ran, not_ran):
    """The sub-k slices beyond one chunk go through value bands (few runs) or are spread by value (many runs); chunks + merge passes are
    what remains for positions that crowd — and for KMX_NO_SPLIT / KMX_NO_BANDS (read once per process: a child each), which keep that
    path alive under the same oracle comparison."""
    e = dict(os.environ)
    e.update(env)
    res = subprocess.run([sys.executable, "-c", SUBK_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=e)
    assert res.returncode == 0 and "subk child ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
    stats = " ".join(line for line in res.stdout.splitlines() if line.startswith("stats"))
    for name in ran:
        assert f"'{name}'" in stats, (name, stats)
    for name in not_ran:
        assert f"'{name}'" not in stats, (name, stats)
