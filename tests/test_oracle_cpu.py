"""CPU suite, part 1: the oracle itself is pinned — against the real reference where it builds
(oracle/_ref), against the committed golden vectors, against the thesis' planner table and against
the naive scan.  No GPU, no product code."""
import json
import os

import numpy as np
import pytest

from kmer_index_amd import synth
from tests.golden.make_golden import CONFIGS, make_inputs
from tests.helpers import digest, make_queries

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fast_pow_golden(orc):
    rows = json.load(open(os.path.join(GOLD, "fast_pow.json")))["rows"]
    assert len(rows) > 1000
    for b, e, want in rows:
        assert orc.fast_pow(b, e) == want, (b, e)
    # the known answers of SURVEY §4.4
    assert orc.fast_pow(4, 10) == 1048576 and orc.fast_pow(5, 10) == 9765625 and orc.fast_pow(20, 5) == 3200000
    assert orc.fast_pow(7, 0) == 1 and orc.fast_pow(1, 200) == 1 and orc.fast_pow(3, 64) == 0
    assert orc.fast_pow(2, 62) == 1 << 62 and orc.fast_pow(2, 63) == 0      # LUT entry 63 is "overflow" (fast_pow.hpp:19)


def test_fast_pow_vs_real_reference(orc):
    R = orc.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (reference checkout absent on this machine)")
    for b in list(range(0, 40)) + [255, 1 << 20, (1 << 32) + 3]:
        for e in range(256):
            assert R.ref_fast_pow(b, e) == orc.fast_pow(b, e), (b, e)


def test_bitset_golden(orc):
    cases = json.load(open(os.path.join(GOLD, "bitset.json")))["cases"]
    for c in cases:
        got = orc.bitset_words(c["n_bits"], c["fill"], [tuple(o) for o in c["ops"]])
        if c.get("out_of_range"):
            assert got is None
            continue
        words, ones = got
        assert [int(w) for w in words] == c["words"], c
        assert ones == c["ones"]
        assert len(words) == c["n_bits"] // 64 + 1      # compressed_bitset.hpp:23


def test_bitset_vs_real_reference(orc):
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    z = synth.u64_stream(99, 2000)
    zi = 0
    for n_bits in (1, 63, 64, 65, 300, 4097):
        for fill in (0, 1):
            ops = [(int(z[zi + 2 * i] % np.uint64(n_bits)), int(z[zi + 2 * i + 1] & np.uint64(1))) for i in range(60)]
            zi += 120
            a = orc.bitset_words(n_bits, fill, ops, which="orc")
            b = orc.bitset_words(n_bits, fill, ops, which="ref")
            assert np.array_equal(a[0], b[0]) and a[1] == b[1]


def test_thread_pool_of_real_reference(orc):
    R = orc.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built")
    assert R.ref_pool_sum(4, 1000) == 500500
    assert R.ref_pool_sum(1, 10) == 55


def test_batch_search_carried_by_the_reference_thread_pool(orc):
    """The CPU baseline's batch (bench.py cpu_baseline) runs its chunk tasks on the REFERENCE's own thread_pool
    (thread_pool.{hpp,cpp} from /root/reference in oracle/_ref): same result as on the restated pool."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    text = synth.ranks(3, 150_000, 4)
    q, off = synth.mixed_queries(4, text, 30_000, [5, 8, 9, 17], 4)
    idx = orc.Index(text, 4, [8])
    a = idx.search_batch(q, off, n_threads=6)
    b = idx.search_batch(q, off, n_threads=6, reference_pool=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_planner_thesis_kat(orc):
    gold = json.load(open(os.path.join(GOLD, "planner.json")))
    multi, nk = orc.plan(gold["thesis"]["ks"])
    for q, want in gold["thesis"]["rows"].items():
        q = int(q)
        if want is None:                       # "n/a": falls back on one element
            assert not multi[q] and len(nk[q]) == 1
        else:
            assert multi[q] and nk[q] == want
    for key, t in gold["tables"].items():
        ks = [int(x) for x in key.split(",")]
        multi, nk = orc.plan(ks)
        assert nk[:64] == t["nk_first_64"] and int(multi.sum()) == t["n_multi"]
        assert int(sum((q + 1) * (i + 1) * k for q in range(len(nk)) for i, k in enumerate(nk[q])) % (1 << 61)) == t["digest"]


def test_planner_survey_table(orc):
    """ks = {8,10,12}."""
    multi, nk = orc.plan([8, 10, 12])
    want = {8: [8], 9: [10], 10: [10], 11: [12], 12: [12], 13: [8], 16: [8], 17: [10], 19: [10], 20: [10, 10], 21: [12],
            22: [10, 12], 23: [12], 24: [12, 12], 25: [10], 29: [10], 30: [10, 10, 10], 32: [10, 10, 12], 34: [10, 12, 12], 36: [12, 12, 12]}
    for q, w in want.items():
        assert nk[q] == w, (q, nk[q])
    assert multi[10] and multi[12] and multi[20] and multi[22] and not multi[8] and not multi[21]


@pytest.mark.parametrize("name", list(CONFIGS))
def test_search_golden(orc, name):
    """The restatement reproduces the committed vectors (which were ground-truthed at generation time)."""
    cfg = CONFIGS[name]
    text, q, off = make_inputs(cfg)
    g = np.load(os.path.join(GOLD, f"search_{name}.npz"))
    assert int(np.sum(q.astype(np.uint64) * (np.arange(q.size, dtype=np.uint64) % np.uint64(251) + np.uint64(1)))) == int(g["input_digest"][0])
    oidx = orc.Index(text, cfg[0], cfg[2])
    h_off, pos, status, _ = oidx.search_batch(q, off, n_threads=4)
    assert np.array_equal(np.diff(h_off).astype(np.uint32), g["counts"])
    assert np.array_equal(status.astype(np.uint8), g["status"])
    assert digest(h_off, pos) == int(g["digest"][0])
    nf = g["first_off"].size - 1
    assert np.array_equal(pos[:int(h_off[nf])], g["first_lists"])


def test_faithful_vs_intended_envelope(orc):
    """Inside the envelope (<= 2 parts with a rest, <= 2 summands) the line-by-line restatement, the repaired
    restatement and the naive scan agree; with >= 3 parts + rest / >= 3 summands only the repaired one matches."""
    text = synth.ranks(31, 60_000, 4)
    single = orc.Index(text, 4, [5])
    multi = orc.Index(text, 4, [9, 10])
    inside, outside = 0, 0
    for m in (4, 5, 7, 10, 13, 14, 15, 20):           # k=5: <= 2 parts + rest, or exact multiples
        for t in range(6):
            s = (t * 7919 + m * 104729) % (text.size - m)
            q = text[s:s + m]
            nv = orc.naive_scan(text, q)
            assert np.array_equal(single.search(q, orc.MODE_FAITHFUL)[1], nv)
            assert np.array_equal(single.search(q, orc.MODE_INTENDED)[1], nv)
            inside += 1
    for m in (16, 17, 18, 19, 22):                   # k=5: 3-4 parts + rest (kmer_index.hpp:314 defect)
        for t in range(6):
            s = (t * 7919 + m * 104729) % (text.size - m)
            q = text[s:s + m]
            nv = orc.naive_scan(text, q)
            assert np.array_equal(single.search(q, orc.MODE_INTENDED)[1], nv)
            outside += int(not np.array_equal(single.search(q, orc.MODE_FAITHFUL)[1], nv))
    for m in (27, 28, 29, 30):                       # {9,10}: 3 summands (kmer_index.hpp:526,535 defects)
        for t in range(6):
            s = (t * 7919 + m * 104729) % (text.size - m)
            q = text[s:s + m]
            nv = orc.naive_scan(text, q)
            assert 0 < nv.size
            assert np.array_equal(multi.search(q, orc.MODE_INTENDED)[1], nv)
            outside += int(not np.array_equal(multi.search(q, orc.MODE_FAITHFUL)[1], nv))
    assert inside == 48 and outside > 20


def test_result_object_semantics(orc):
    """kmer_index_result quirks kept by the restatement: size() counts mask bits, so it is 0 for bypass results."""
    text = synth.ranks(3, 20_000, 4)
    idx = orc.Index(text, 4, [6])
    st, pos, m = idx.search(text[100:106], want_mask=True)
    assert m["bypass"] and m["bits"] == 0 and pos.size >= 1 and m["candidates"] == pos.size
    st, pos, m = idx.search(text[100:110], want_mask=True)             # 1 part + rest
    assert not m["bypass"] and m["bits"] == m["candidates"] and m["words"].size == m["bits"] // 64 + 1
    bits = np.unpackbits(m["words"].view(np.uint8), bitorder="little")[:m["bits"]]
    assert bits.sum() == pos.size


def test_errors(orc):
    text = synth.ranks(3, 20_000, 4)
    idx = orc.Index(text, 4, [13])
    assert idx.search(np.zeros(0, np.uint8))[0] == orc.ST_EMPTY_QUERY
    assert idx.search(np.zeros(10000, np.uint8))[0] == orc.ST_TOO_LONG
    assert idx.search(text[0:1])[0] == orc.ST_FANOUT          # 4^12 > 1e7
    assert idx.search(text[0:2])[0] == orc.ST_OK              # 4^11 = 4194304 <= 1e7
    assert idx.search(text[5:5 + 14])[0] == orc.ST_FANOUT     # rest of 1 letter -> 4^12 prefix buckets
    assert idx.search(synth.ranks(8, 14, 4))[0] == orc.ST_OK  # first part misses -> early empty result, no throw


def test_batch_on_thread_pool_matches_serial(orc):
    text = synth.ranks(12, 80_000, 5)
    idx = orc.Index(text, 5, [6, 9], n_threads=2)
    q, off = make_queries(text, 5, [3, 6, 9, 12, 15, 18], 20, seed=3)
    a = idx.search_batch(q, off, n_threads=1)
    b = idx.search_batch(q, off, n_threads=7)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_last_hash_int_truncation_is_restated(orc):
    """Defect 4 (kmer_index.hpp:214-226): `int last_hash` keeps 32 bits of the previous part's hash and is sign-extended in
    the comparison at :219.  With sigma^k > 2^32 (DNA4, k >= 17) a part whose hash is h >= 2^32 with bit 31 clear, followed
    by the part whose hash is h & 0xFFFFFFFF, makes the reference reuse the FIRST part's bucket for the second.
    FAITHFUL restates that; INTENDED == naive scan is what the engine answers (tests/test_search_gpu.py)."""
    sigma, k = 4, 17
    # part1: first letter != 0 (hash >= 2^32), second letter in {0, 1} (bit 31 clear); part2 = 'A' + part1[1:]  (hash = low 32 bits)
    part1 = np.array([2, 1] + [3, 0, 2, 1, 1, 3, 2, 0, 3, 1, 2, 2, 0, 1, 3], np.uint8)
    part2 = part1.copy()
    part2[0] = 0
    h1 = sum(int(r) << (2 * (k - 1 - i)) for i, r in enumerate(part1))
    h2 = sum(int(r) << (2 * (k - 1 - i)) for i, r in enumerate(part2))
    assert h1 >= 1 << 32 and not (h1 >> 31) & 1 and h2 == h1 & 0xFFFFFFFF
    text = synth.ranks(4242, 6000, sigma)
    text[1000:1000 + k] = part1; text[1000 + k:1000 + 2 * k] = part2          # a true occurrence of part1 + part2
    text[3000:3000 + k] = part1; text[3000 + k:3000 + 2 * k] = part1          # part1 twice in a row: NOT an occurrence
    idx = orc.Index(text, sigma, [k])
    q = np.concatenate([part1, part2])
    truth = orc.naive_scan(text, q)
    assert truth.tolist() == [1000]
    st_i, got_i = idx.search(q, mode=orc.MODE_INTENDED)
    st_f, got_f = idx.search(q, mode=orc.MODE_FAITHFUL)
    assert st_i == orc.ST_OK and st_f == orc.ST_OK
    assert got_i.tolist() == [1000]                                             # INTENDED == naive
    assert got_f.tolist() == [3000]                                             # the reference: first part's bucket used twice
    # the false equality needs the truncation: the same construction one letter shorter (4^16 = 2^32: every hash < 2^32
    # equals its own low word, and hashes in [2^31, 2^32) compare unequal after sign extension) is answered correctly
    k2 = 16
    p1, p2 = part1[:k2].copy(), part2[:k2].copy()
    t2 = synth.ranks(4243, 6000, sigma)
    t2[1000:1000 + k2] = p1; t2[1000 + k2:1000 + 2 * k2] = p2
    t2[3000:3000 + k2] = p1; t2[3000 + k2:3000 + 2 * k2] = p1
    idx2 = orc.Index(t2, sigma, [k2])
    q2 = np.concatenate([p1, p2])
    for mode in (orc.MODE_INTENDED, orc.MODE_FAITHFUL):
        assert idx2.search(q2, mode=mode)[1].tolist() == orc.naive_scan(t2, q2).tolist() == [1000]
    # a previous hash in [2^31, 2^32): never "equal" after sign extension, even when the next part IS the same k-mer
    p3 = np.array([2] + [1] * 15, np.uint8)                                     # hash = 2 << 30 | ... >= 2^31
    t3 = synth.ranks(4244, 6000, sigma)
    t3[500:500 + k2] = p3; t3[500 + k2:500 + 2 * k2] = p3
    idx3 = orc.Index(t3, sigma, [k2])
    q3 = np.concatenate([p3, p3])
    for mode in (orc.MODE_INTENDED, orc.MODE_FAITHFUL):
        assert idx3.search(q3, mode=mode)[1].tolist() == orc.naive_scan(t3, q3).tolist()
    idx.close(); idx2.close(); idx3.close()
