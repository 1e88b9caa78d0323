"""GPU parity at BASELINE.json's full sizes, through size-independent properties.

At n = 1e8 the CPU oracle would take minutes per config, so the full batches are checked by properties that
pin the result completely for exact queries and strongly for the others:
  * per-query hit COUNT == number of text k-mers with the query's rank-hash (numpy bincount over the text,
    independent of the engine) — with ascending order and the re-hash property below this determines the list;
  * every list strictly ascending;
  * every reported position re-reads to the query (text[pos + j] == q[j]) on a large sample;
  * a sample of queries equals the oracle / naive scan exactly.
"""
import numpy as np
import pytest

from kmer_index_amd import synth

pytestmark = pytest.mark.gpu


def kmer_hashes(text, k, sigma):
    """Rank-hash of every k-mer of the text (rolling, vectorised)."""
    n = text.size - k + 1
    h = np.zeros(n, np.uint64)
    for j in range(k):
        h *= np.uint64(sigma)
        h += text[j:j + n]
    return h


def query_hashes(q, k, sigma):
    w = (np.uint64(sigma) ** np.arange(k - 1, -1, -1, dtype=np.uint64))
    return (q.reshape(-1, k).astype(np.uint64) * w).sum(axis=1)


def check_lists(text, qranks_2d, hit_off, positions, sample_queries):
    """Ascending everywhere; re-hash property on a sample of queries (all their hits)."""
    # ascending inside every list: diff > 0 except at list boundaries
    cnt = np.diff(hit_off).astype(np.int64)
    for s in range(0, positions.size - 1, 1 << 26):
        e = min(positions.size - 1, s + (1 << 26))
        d = positions[s + 1:e + 1].astype(np.int64) - positions[s:e].astype(np.int64)
        bad = np.nonzero(d <= 0)[0] + s + 1          # indices that must be list starts
        assert np.isin(bad, hit_off[:-1].astype(np.int64)).all(), "a hit list is not strictly ascending"
    m = qranks_2d.shape[1]
    for qi in sample_queries:
        p = positions[int(hit_off[qi]):int(hit_off[qi + 1])].astype(np.int64)
        for j in range(m):
            assert (text[p + j] == qranks_2d[qi, j]).all()


def test_cfg2_dna4_k10_full(engine, orc):
    """BASELINE configs[1]: DNA4 text 1e8, k=10, 1e7 random 10-mers on one GPU."""
    n, k, sigma, nq = 100_000_000, 10, 4, 10_000_000
    text = synth.ranks(1002, n, sigma)
    q, off = synth.uniform_queries(2002, nq, k, sigma)
    idx = engine.Index(text, sigma, [k], table=engine.TABLE_OPEN)
    res = idx.search(q, off)
    hit_off, positions, status, kinds = res.host()
    assert (status == 0).all()
    per_key = np.bincount(kmer_hashes(text, k, sigma).astype(np.int64), minlength=sigma ** k)
    want = per_key[query_hashes(q, k, sigma).astype(np.int64)]
    assert np.array_equal(np.diff(hit_off).astype(np.int64), want)
    assert positions.size == int(want.sum()) == res.counts()["n_hits"]
    rng = np.random.default_rng(5)
    check_lists(text, q.reshape(-1, k), hit_off, positions, rng.integers(0, nq, 20000))
    # exact equality with the naive scan for a handful
    for qi in rng.integers(0, nq, 5):
        assert np.array_equal(positions[int(hit_off[qi]):int(hit_off[qi + 1])], orc.naive_scan(text, q[qi * k:(qi + 1) * k]))
    # dense table gives the identical result
    idx2 = engine.Index(text, sigma, [k], table=engine.TABLE_DENSE)
    r2 = idx2.search(q, off)
    h2, p2, _, _ = r2.host()
    assert np.array_equal(h2, hit_off) and np.array_equal(p2, positions)


def test_cfg3_dna4_multi_k_mixed_full(engine, orc):
    """configs[2]: multi-k {8,10,12} index over 1e8 bp, 1e7 mixed-length queries on the stitch path (BASELINE's size)."""
    n, sigma, ks, nq = 100_000_000, 4, [8, 10, 12], 10_000_000
    text = synth.ranks(1003, n, sigma)
    q, off = synth.mixed_queries(2003, text, nq, [8, 10, 12, 20, 22, 24], sigma)
    idx = engine.Index(text, sigma, ks)
    res = idx.search(q, off)
    hit_off, positions, status, kinds = res.host()
    assert (status == 0).all()
    lens = np.diff(off).astype(np.int64)
    cnt = np.diff(hit_off).astype(np.int64)
    # exact lengths: counts from the text's k-mer spectrum
    for k in (8, 10, 12):
        sel = np.nonzero(lens == k)[0]
        per_key = np.bincount(kmer_hashes(text, k, sigma).astype(np.int64), minlength=sigma ** k)
        starts = off[sel].astype(np.int64)
        qk = np.stack([q[starts + j] for j in range(k)], axis=1)
        assert np.array_equal(cnt[sel], per_key[query_hashes(qk.reshape(-1), k, sigma).astype(np.int64)])
        assert (kinds[sel][cnt[sel] > 0] == engine.KIND_EXACT).all()
    # stitched lengths: every hit re-reads to the query; planted queries are found; random ones almost never hit
    long_sel = np.nonzero(lens >= 20)[0]
    assert (kinds[long_sel] != engine.KIND_PREFIX).all() and (kinds[long_sel] == engine.KIND_STITCH).any()
    assert (cnt[long_sel] >= 1).sum() > 0.45 * long_sel.size
    pos = np.concatenate([positions[int(hit_off[i]):int(hit_off[i + 1])] for i in long_sel[cnt[long_sel] > 0][:200000]]).astype(np.int64)
    qi = np.repeat(long_sel[cnt[long_sel] > 0][:200000], cnt[long_sel[cnt[long_sel] > 0][:200000]])
    for j in range(20):
        assert (text[pos + j] == q[off[qi].astype(np.int64) + j]).all()
    rng = np.random.default_rng(7)
    for i in rng.choice(long_sel, 6, replace=False):
        assert np.array_equal(positions[int(hit_off[i]):int(hit_off[i + 1])], orc.naive_scan(text, q[int(off[i]):int(off[i + 1])]))


def test_cfg4_dna5_k10_one_shard(engine, orc):
    """configs[3]: DNA5 (with N) text 1e8, k=10; one of the 8 shards of 1.25e7 queries."""
    n, k, sigma, nq = 100_000_000, 10, 5, 12_500_000
    text = synth.ranks(1004, n, sigma)
    shard = 3
    # the shard's letters are items [shard*nq*k, (shard+1)*nq*k) of query stream 2004
    z = np.empty(nq * k, np.uint8)
    for s in range(0, nq * k, 1 << 24):
        e = min(nq * k, s + (1 << 24))
        u = synth.u64_stream(2004, e - s, shard * nq * k + s)
        z[s:e] = (((u >> np.uint64(32)) * np.uint64(sigma)) >> np.uint64(32)).astype(np.uint8)
    q = z
    off = np.arange(nq + 1, dtype=np.uint64) * np.uint64(k)
    idx = engine.Index(text, sigma, [k], table=engine.TABLE_OPEN)
    res = idx.search(q, off)
    hit_off, positions, status, kinds = res.host()
    per_key = np.bincount(kmer_hashes(text, k, sigma).astype(np.int64), minlength=sigma ** k)
    want = per_key[query_hashes(q, k, sigma).astype(np.int64)]
    assert np.array_equal(np.diff(hit_off).astype(np.int64), want)
    rng = np.random.default_rng(11)
    check_lists(text, q.reshape(-1, k), hit_off, positions, rng.integers(0, nq, 20000))
    assert (kinds[want == 0] == engine.KIND_NONE).all()


def test_cfg5_aa20_k5_full(engine, orc):
    """configs[4]: AA20 protein text 1e7 residues, k=5, 1e7 queries (half planted)."""
    n, k, sigma, nq = 10_000_000, 5, 20, 10_000_000
    text = synth.ranks(1005, n, sigma)
    q, off = synth.mixed_queries(2005, text, nq, [5], sigma)
    idx = engine.Index(text, sigma, [k], table=engine.TABLE_OPEN)
    res = idx.search(q, off)
    hit_off, positions, status, kinds = res.host()
    per_key = np.bincount(kmer_hashes(text, k, sigma).astype(np.int64), minlength=sigma ** k)
    want = per_key[query_hashes(q, k, sigma).astype(np.int64)]
    assert np.array_equal(np.diff(hit_off).astype(np.int64), want)
    rng = np.random.default_rng(13)
    check_lists(text, q.reshape(-1, k), hit_off, positions, rng.integers(0, nq, 50000))
    oidx = orc.Index(text, sigma, [k])
    o_off, o_pos, _, _ = oidx.search_batch(q[:200_000 * k], off[:200_001], n_threads=8)
    assert np.array_equal(o_off, hit_off[:200_001]) and np.array_equal(o_pos, positions[:int(hit_off[200_000])])


def test_more_than_2_to_32_hits(engine, orc):
    """A batch whose hit total exceeds 2^32: 64-bit offsets end to end (3.2e6 8-mers on the 1e8-bp text, ~4.9e9 hits)."""
    n, k, sigma, nq = 100_000_000, 8, 4, 3_200_000
    text = synth.ranks(1003, n, sigma)
    q, off = synth.uniform_queries(4242, nq, k, sigma)
    idx = engine.Index(text, sigma, [k])
    per_key = np.bincount(kmer_hashes(text, k, sigma).astype(np.int64), minlength=sigma ** k)
    want = per_key[query_hashes(q, k, sigma).astype(np.int64)].astype(np.uint64)
    assert int(want.sum()) > (1 << 32)
    res = idx.search(q, off, flags=engine.SEARCH_COUNT_ONLY)
    ho = res.host()[0]
    assert np.array_equal(np.diff(ho), want) and int(ho[-1]) == int(want.sum())
    res.close()
    res = idx.search(q, off)
    hit_off, positions, status, kinds = res.host()
    assert np.array_equal(hit_off, ho) and positions.size == int(want.sum())
    rng = np.random.default_rng(17)
    sample = np.concatenate([rng.integers(0, nq, 3000), np.arange(nq - 200, nq)])     # incl. lists beyond offset 2^32
    check_lists(text, q.reshape(-1, k), hit_off, positions, sample)
