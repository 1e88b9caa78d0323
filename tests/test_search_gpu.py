"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the naive scan."""
import numpy as np
import pytest

from kmer_index_amd import synth
from tests.helpers import inside_envelope, make_queries, pack

pytestmark = pytest.mark.gpu

CASES = [
    # (name, sigma, n, ks, lengths)  — lengths stay inside the reference's correct envelope
    ("dna4_k5", 4, 100_000, [5], list(range(1, 16)) + [20]),
    ("dna4_k10", 4, 400_000, [10], list(range(4, 31)) + [40]),
    ("dna5_k10", 5, 400_000, [10], list(range(5, 31))),
    ("aa20_k5", 20, 300_000, [5], list(range(1, 16))),
    ("dna4_multi", 4, 400_000, [8, 10, 12], list(range(2, 30)) + [33, 35]),
    ("dna15_k3_multi", 15, 200_000, [3, 4, 5], list(range(1, 10))),
]


def _compare(engine, orc, text, sigma, ks, qranks, qoff, table):
    idx = engine.Index(text, sigma, ks, table=table)
    res = idx.search(qranks, qoff)
    hit_off, positions, status, kinds = res.host()
    oidx = orc.Index(text, sigma, ks)
    o_off, o_pos, o_status, _ = oidx.search_batch(qranks, qoff, mode=orc.MODE_INTENDED, n_threads=4)
    n_off, n_pos = orc.naive_batch(text, qranks, qoff)
    ok_mask = o_status == 0
    assert np.array_equal(status, o_status.astype(np.uint8)), "per-query status differs from the oracle"
    assert np.array_equal(hit_off, o_off), "hit_off differs from the oracle"
    assert np.array_equal(positions, o_pos), "positions differ from the oracle"
    # the line-by-line restatement of the reference (its defects included): inside the envelope of SURVEY 4.3 the HIP
    # result IS the reference's result, query by query — asserted on the GPU side, not by transitivity
    f_off, f_pos, f_status, _ = oidx.search_batch(qranks, qoff, mode=orc.MODE_FAITHFUL, n_threads=4)
    plan = orc.plan(ks)
    n_inside = 0
    for i in range(qoff.size - 1):
        if not inside_envelope(plan, ks, int(qoff[i + 1] - qoff[i])):
            continue
        n_inside += 1
        assert status[i] == f_status[i], f"query {i}: status differs from the faithful restatement"
        assert np.array_equal(positions[int(hit_off[i]):int(hit_off[i + 1])], f_pos[int(f_off[i]):int(f_off[i + 1])]), \
            f"query {i} (m={int(qoff[i+1]-qoff[i])}, inside the envelope) differs from the faithful restatement of the reference"
    assert n_inside > (qoff.size - 1) // 2
    # ground truth for every query the reference does not reject
    for i in np.nonzero(ok_mask)[0]:
        a = positions[int(hit_off[i]):int(hit_off[i + 1])]
        b = n_pos[int(n_off[i]):int(n_off[i + 1])]
        assert np.array_equal(a, b), f"query {i} (m={int(qoff[i+1]-qoff[i])}) differs from the naive scan"
    res.close()
    idx.close()
    return kinds


@pytest.mark.parametrize("table", ["open", "dense"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_parity_vs_oracle_and_naive(engine, orc, case, table):
    name, sigma, n, ks, lengths = case
    text = synth.ranks(1000 + len(name), n, sigma)
    qranks, qoff = make_queries(text, sigma, lengths, 30, seed=77)
    tk = engine.TABLE_OPEN if table == "open" else engine.TABLE_DENSE
    kinds = _compare(engine, orc, text, sigma, ks, qranks, qoff, tk)
    assert (kinds == engine.KIND_EXACT).any()


def test_outside_the_envelope_the_engine_follows_ground_truth_not_the_reference(engine, orc):
    """The documented divergence, pinned (DESIGN 'Semantics'): for >= 3 full parts + a rest on one k
    (kmer_index.hpp:314) and for >= 3 multi-k summands (:526, :535) the reference's search() is wrong against ground truth.
    The engine returns the exact occurrence list there (== naive scan), which is NOT what the line-by-line restatement of the
    reference returns for planted queries; everywhere else in the same batches the two agree."""
    text = synth.ranks(31, 120_000, 4)
    for ks, outside_lengths, inside_lengths in (([5], [16, 17, 18, 19, 22], [4, 5, 7, 10, 13, 14, 15, 20]),
                                               ([9, 10], [27, 28, 29, 30], [9, 10, 18, 19, 20])):
        plan = orc.plan(ks)
        assert all(not inside_envelope(plan, ks, m) for m in outside_lengths) and all(inside_envelope(plan, ks, m) for m in inside_lengths)
        qs = [text[s0:s0 + m].copy() for m in outside_lengths + inside_lengths
              for s0 in ((t * 7919 + m * 104729) % (text.size - m) for t in range(8))]           # planted: every one has a true hit
        qranks, qoff = pack(qs)
        idx = engine.Index(text, 4, ks)
        hit_off, positions, status, kinds = idx.search(qranks, qoff).host()
        oidx = orc.Index(text, 4, ks)
        f_off, f_pos, f_st, _ = oidx.search_batch(qranks, qoff, mode=orc.MODE_FAITHFUL, n_threads=4)
        n_off, n_pos = orc.naive_batch(text, qranks, qoff)
        differs = 0
        for i, q in enumerate(qs):
            mine = positions[int(hit_off[i]):int(hit_off[i + 1])]
            truth = n_pos[int(n_off[i]):int(n_off[i + 1])]
            faithful = f_pos[int(f_off[i]):int(f_off[i + 1])]
            assert truth.size >= 1 and np.array_equal(mine, truth), (ks, len(q), i)
            if inside_envelope(plan, ks, len(q)):
                assert np.array_equal(mine, faithful), (ks, len(q), i)
            else:
                differs += int(not np.array_equal(mine, faithful))
        assert differs >= len(outside_lengths) * 4, (ks, differs)      # the reference's defects show on most planted queries
        idx.close()


def test_the_engine_does_not_reproduce_the_int_last_hash_defect(engine, orc):
    """Defect 4 (kmer_index.hpp:214-226, `int last_hash`): with sigma^k > 2^32 the reference reuses the previous part's bucket
    for a part whose hash equals the previous hash's low 32 bits.  The engine answers ground truth (== naive == INTENDED), which
    here is NOT the faithful restatement's answer; CPU twin: tests/test_oracle_cpu.py::test_last_hash_int_truncation_is_restated."""
    sigma, k = 4, 17
    part1 = np.array([2, 1] + [3, 0, 2, 1, 1, 3, 2, 0, 3, 1, 2, 2, 0, 1, 3], np.uint8)
    part2 = part1.copy()
    part2[0] = 0
    text = synth.ranks(4242, 6000, sigma)
    text[1000:1000 + k] = part1; text[1000 + k:1000 + 2 * k] = part2
    text[3000:3000 + k] = part1; text[3000 + k:3000 + 2 * k] = part1
    qs = [np.concatenate([part1, part2]), np.concatenate([part1, part1]), np.concatenate([part2, part1])]
    qranks, qoff = pack(qs)
    for keep in (False, True):
        idx = engine.Index(text, sigma, [k])
        hit_off, positions, status, kinds = idx.search(qranks, qoff, flags=engine.SEARCH_DEFAULT | (engine.SEARCH_KEEP_MASKS if keep else 0)).host()
        oidx = orc.Index(text, sigma, [k])
        lists = [positions[int(hit_off[i]):int(hit_off[i + 1])].tolist() for i in range(3)]
        assert lists == [orc.naive_scan(text, q).tolist() for q in qs] == [[1000], [3000], []]
        assert [oidx.search(q, mode=orc.MODE_INTENDED)[1].tolist() for q in qs] == lists
        assert oidx.search(qs[0], mode=orc.MODE_FAITHFUL)[1].tolist() == [3000]      # what the reference would return
        idx.close()


@pytest.mark.parametrize("sigma,ks", [(4, [8, 10, 12]), (20, [3, 5]), (4, [6, 9])])
def test_engine_plan_and_reference_plan_return_the_same_lists(engine, orc, sigma, ks):
    """A single-k query longer than its k is answered from the largest k that fits (kmx_plan_engine) unless the call asks for
    the reference's planner (KMX_SEARCH_REFERENCE_PLAN, implied by KEEP_MASKS): statuses, offsets and position lists are the
    same either way and equal the oracle's; kinds differ at most between NONE and STITCH for queries without hits."""
    text = synth.ranks(77 + sigma, 500_000, sigma)
    used = engine.plan_engine(ks, sigma, 200)
    multi, nk_sum = engine.plan(ks, 200)
    lengths = [m for m in range(max(ks) + 1, 70) if not (multi[m] and len(ks) > 1) and used[m] != nk_sum[m][0]][:14]
    assert len(lengths) >= 5, "the case must hold lengths on which the two planners disagree"
    lengths += [max(ks), 2 * max(ks), min(ks) - 1]
    qranks, qoff = make_queries(text, sigma, lengths, 60, seed=21)
    idx = engine.Index(text, sigma, ks)
    a = idx.search(qranks, qoff).host()
    b = idx.search(qranks, qoff, flags=engine.SEARCH_REFERENCE_PLAN).host()
    c = idx.search(qranks, qoff, flags=engine.SEARCH_KEEP_MASKS).host()
    o_off, o_pos, o_st, _ = orc.Index(text, sigma, ks).search_batch(qranks, qoff, n_threads=4)
    for r in (a, b, c):
        assert np.array_equal(r[2], o_st.astype(np.uint8)) and np.array_equal(r[0], o_off) and np.array_equal(r[1], o_pos)
    assert np.array_equal(b[3], c[3])
    hits = np.diff(a[0]) > 0
    assert np.array_equal(a[3][hits], b[3][hits])
    differ = a[3] != b[3]
    assert set(a[3][differ].tolist()) | set(b[3][differ].tolist()) <= {engine.KIND_NONE, engine.KIND_STITCH}
    idx.close()


def test_exact_k10_large_batch(engine, orc):
    """The headline shape, down-scaled: uniform random 10-mers, every one present many times."""
    text = synth.ranks(1002, 2_000_000, 4)
    qranks, qoff = synth.uniform_queries(2002, 200_000, 10, 4)
    idx = engine.Index(text, 4, [10], table=engine.TABLE_OPEN)
    res = idx.search(qranks, qoff)
    hit_off, positions, status, kinds = res.host()
    oidx = orc.Index(text, 4, [10])
    o_off, o_pos, o_status, _ = oidx.search_batch(qranks, qoff, n_threads=8)
    assert np.array_equal(hit_off, o_off)
    assert np.array_equal(positions, o_pos)
    assert (status == 0).all()
    c = res.counts()
    assert c["n_exact"] + (kinds == engine.KIND_NONE).sum() == 200_000
    # size-independent properties: every list ascending, every hit re-hashes to its query
    d = np.diff(positions.astype(np.int64))
    starts = hit_off[1:-1].astype(np.int64)
    inner = np.ones(positions.size - 1, bool)
    inner[starts[(starts > 0) & (starts < positions.size)] - 1] = False
    assert (d[inner] > 0).all()
    qi = np.repeat(np.arange(200_000), np.diff(hit_off).astype(np.int64))
    for j in range(10):
        assert np.array_equal(text[positions.astype(np.int64) + j], qranks.reshape(-1, 10)[qi, j])


def test_edge_cases(engine, orc):
    text = synth.ranks(5, 5000, 4)
    idx = engine.Index(text, 4, [6, 9])
    # empty batch
    r = idx.search(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    ho, pos, st, kd = r.host()
    assert ho.tolist() == [0] and pos.size == 0
    # empty query, too-long query, bad rank, and a normal one in the same batch
    qs = [np.zeros(0, np.uint8), np.zeros(10000, np.uint8), np.array([0, 1, 7, 2, 1, 0], np.uint8), text[10:16].copy(),
          text[4994:5000].copy(), text[0:9].copy(), np.zeros(9999, np.uint8)]
    qranks, qoff = pack(qs)
    r = idx.search(qranks, qoff)
    ho, pos, st, kd = r.host()
    assert st.tolist()[:5] == [engine.Q_EMPTY_QUERY, engine.Q_TOO_LONG, engine.Q_BAD_RANK, engine.Q_OK, engine.Q_OK]
    hits = engine.split_hits(ho, pos)
    assert 10 in hits[3].tolist() and 4994 in hits[4].tolist() and 0 in hits[5].tolist()
    for i in (3, 4, 5, 6):
        assert np.array_equal(hits[i], orc.naive_scan(text, qs[i]))


def test_subk_fanout_error_matches_reference(engine, orc):
    """sigma^(k-m) > 1e7 throws in the reference (kmer_index.hpp:119-122), also via the rest of a long query."""
    text = synth.ranks(9, 300_000, 4)
    idx = engine.Index(text, 4, [13])
    oidx = orc.Index(text, 4, [13])
    qs = [text[100:101].copy(), text[100:102].copy(), text[100:113 + 1].copy(), text[100:113 + 2].copy(),
          synth.ranks(1, 14, 4), text[5:5 + 12].copy()]
    qranks, qoff = pack(qs)
    r = idx.search(qranks, qoff)
    ho, pos, st, kd = r.host()
    o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff)
    assert np.array_equal(st, o_st.astype(np.uint8))
    assert engine.Q_SUBK_FANOUT in st.tolist()
    assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)


def test_masks_match_oracle_bitset(engine, orc):
    """KEEP_MASKS: candidate run + compressed_bitset words equal the reference-shaped result object."""
    text = synth.ranks(21, 200_000, 4)
    ks = [8, 10, 12]
    idx = engine.Index(text, 4, ks, keep_host_arena=True)
    oidx = orc.Index(text, 4, ks)
    qranks, qoff = make_queries(text, 4, [13, 16, 19, 20, 22, 24, 27], 12, seed=5)
    r = idx.search(qranks, qoff, flags=engine.SEARCH_KEEP_MASKS)
    ho, pos, st, kd = r.host()
    base, words_ptr, cand_cnt, cand_src = r.masks()
    arena = idx.arena_host()
    import ctypes as C
    n_checked = 0
    for i in range(qoff.size - 1):
        if kd[i] != engine.KIND_STITCH:
            continue
        q = qranks[int(qoff[i]):int(qoff[i + 1])]
        ost, opos, om = oidx.search(q, want_mask=True)
        assert not om["bypass"] and om["candidates"] == cand_cnt[i]
        nw = cand_cnt[i] // 64 + 1
        words = np.ctypeslib.as_array(C.cast(words_ptr, C.POINTER(C.c_uint64)), shape=(int(base[i]) + nw,))[int(base[i]):]
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:cand_cnt[i]].astype(bool)
        obits = np.unpackbits(om["words"].view(np.uint8), bitorder="little")[:om["bits"]].astype(bool)
        assert np.array_equal(bits, obits)
        cands = arena[int(cand_src[i]):int(cand_src[i]) + int(cand_cnt[i])]
        assert np.array_equal(cands[bits], pos[int(ho[i]):int(ho[i + 1])])
        n_checked += 1
    assert n_checked > 10


def test_result_reuse_and_growth(engine, orc):
    """One result handle reused across batches of growing and shrinking size (exercises the fused
    scan+tile-table path, the fallback partition kernel, and buffer growth)."""
    text = synth.ranks(77, 300_000, 4)
    idx = engine.Index(text, 4, [7])
    oidx = orc.Index(text, 4, [7])
    res = engine.Result()
    for nq in (10, 5000, 300, 40000, 40000, 1, 20000):
        qranks, qoff = synth.uniform_queries(500 + nq, nq, 7, 4)
        idx.search(qranks, qoff, result=res)
        ho, pos, st, kd = res.host()
        o_off, o_pos, _, _ = oidx.search_batch(qranks, qoff, n_threads=4)
        assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos), nq


@pytest.mark.parametrize("sigma,k", [(2, 63), (3, 40), (4, 31), (5, 27), (20, 14), (27, 13)])
def test_largest_valid_k_of_an_alphabet(engine, orc, sigma, k):
    """k at the limit of static_assert(k < 64 / log2(sigma)) (kmer_index.hpp:42-43): hashes use (nearly) all 64 bits.
    (2, 63) is the one pair whose key space the reference's fast_pow cannot express — fast_pow(2, 63) == 0
    (fast_pow.hpp:19) — found by the fuzz soak, seed 704."""
    rng = np.random.default_rng(sigma * 100 + k)
    text = (rng.integers(0, sigma, 9000) * (rng.integers(0, 3, 9000) == 0)).astype(np.uint8)    # low entropy: repeats exist
    idx = engine.Index(text, sigma, [k], table=engine.TABLE_OPEN)
    qs = [text[s0:s0 + m].copy() for m in (k, k, 2 * k, 2 * k + 3, k + 1, 3 * k) for s0 in (0, 17, 4000, 9000 - 3 * k)]
    qs += [rng.integers(0, sigma, k).astype(np.uint8), np.zeros(k, np.uint8), np.zeros(2 * k + 1, np.uint8)]
    qranks, qoff = pack(qs)
    ho, pos, st, _ = idx.search(qranks, qoff).host()
    o_off, o_pos, o_st, _ = orc.Index(text, sigma, [k]).search_batch(qranks, qoff, mode=orc.MODE_INTENDED, n_threads=2)
    assert np.array_equal(st, o_st.astype(np.uint8)) and np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
    assert int((st == 0).sum()) >= 15                         # (a rest of 1..3 letters is the fan-out error, :119-122)
    for i, q in enumerate(qs):
        if st[i] == 0:
            assert np.array_equal(pos[int(ho[i]):int(ho[i + 1])], orc.naive_scan(text, q))


@pytest.mark.parametrize("ks", [[20], [16, 24, 31], [9, 14]])
def test_sparse_buckets_long_queries_resolved_in_lookup(engine, orc, ks):
    """Large k: buckets of one to a few positions.  k_lookup follows the few start positions of such a query through
    its parts itself (single-k: candidates of the first part; multi-k: anchors from the last summand) and marks the
    query resolved when the survivors are one run of the first bucket; the others go to k_validate_tiny / k_validate.
    Segments of the text are repeated (buckets of 2-4, survivors in the middle of a bucket, several survivors) and
    queries end inside / outside the repeats.  Ground truth: the naive scan; with and without KEEP_MASKS."""
    rng = np.random.default_rng(sum(ks))
    text = rng.integers(0, 4, 300_000).astype(np.uint8)
    for rep, (src0, dst0, ln) in enumerate([(1000, 50_000, 400), (1000, 120_000, 250), (1100, 200_000, 180), (70_000, 260_000, 90)]):
        text[dst0:dst0 + ln] = text[src0:src0 + ln]
    text[150_000:150_060] = 0                                   # a low-complexity stretch: longer buckets in the middle
    idx = engine.Index(text, 4, ks)
    qs = []
    for m in (max(ks) + 1, 2 * min(ks), 48, 64, 100, 150, 200):
        for s0 in (1000, 1050, 1100, 1190, 1300, 50_000, 50_100, 120_100, 200_050, 70_010, 260_020, 149_990, 5, 299_000 - m):
            q = text[s0:s0 + m].copy()
            qs.append(q)
            if s0 % 100 == 0:
                q2 = q.copy()
                q2[m - 2] = (q2[m - 2] + 1) % 4                 # a mismatch near the end: every candidate fails late
                qs.append(q2)
    qranks, qoff = pack(qs)
    o_off, o_pos, o_st, _ = orc.Index(text, 4, ks).search_batch(qranks, qoff, mode=orc.MODE_INTENDED, n_threads=4)
    n_multi_hit = 0
    for flags in (engine.SEARCH_DEFAULT, engine.SEARCH_KEEP_MASKS, engine.SEARCH_COUNT_ONLY):
        r = idx.search(qranks, qoff, flags=flags)
        ho, pos, st, kd = r.host()
        assert np.array_equal(st, o_st.astype(np.uint8)) and np.array_equal(ho, o_off), flags
        if flags == engine.SEARCH_COUNT_ONLY:
            continue
        assert np.array_equal(pos, o_pos), flags
        for i, q in enumerate(qs):
            if st[i] == 0:
                want = orc.naive_scan(text, q)
                assert np.array_equal(pos[int(ho[i]):int(ho[i + 1])], want), (flags, i, len(q))
                n_multi_hit += want.size > 1
    assert n_multi_hit > 20                                      # the repeats really produce several occurrences


def test_async_searches_rotate_over_two_results(engine, orc):
    """KMX_SEARCH_ASYNC: the call returns with the first half of the search enqueued; the next touch of the result
    completes it.  Two results in rotation over batches of every kind mix (the second halves — validation, sorts,
    re-fills — then run behind the next batch's first half) must give what the synchronous calls give."""
    import torch
    text = synth.ranks(31, 400_000, 4)
    ks = [6, 9, 12]
    idx = engine.Index(text, 4, ks)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    batches = []
    for b in range(7):
        lens = [[9], [9, 12], [4, 5], [18, 21, 24, 30], [2, 6, 9, 13, 27, 40], [12], [9, 9, 9, 17]][b]
        qr, qo = synth.mixed_queries(400 + b, text, [3000, 1, 800, 5000, 2500, 20000, 64][b], lens, 4)
        batches.append((qr, qo, torch.from_numpy(qr).to(dev), torch.from_numpy(qo.view(np.int64)).to(dev)))
    want = [idx.search(qr, qo).host() for qr, qo, _, _ in batches]
    results = [engine.Result(), engine.Result()]
    inflight = [None, None]
    for rnd in range(3):
        for b, (qr, qo, d_q, d_o) in enumerate(batches):
            slot = (rnd * len(batches) + b) % 2
            if inflight[slot] is not None:                       # collect the batch this handle still holds, then reuse it
                got = results[slot].host()
                for x, y in zip(got, want[inflight[slot]]):
                    assert np.array_equal(x, y), (rnd, b, inflight[slot])
            idx.search_device(d_q.data_ptr(), d_o.data_ptr(), qo.size - 1, flags=engine.SEARCH_ASYNC, stream=stream.cuda_stream,
                              result=results[slot])
            inflight[slot] = b
    for slot in range(2):
        c = results[slot].counts()                                # counts alone complete a pending search
        assert c["n_hits"] == int(want[inflight[slot]][0][-1])
        got = results[slot].host()
        for x, y in zip(got, want[inflight[slot]]):
            assert np.array_equal(x, y)
    # a pending result may simply be released, or searched into again without having been read
    idx.search_device(batches[3][2].data_ptr(), batches[3][3].data_ptr(), batches[3][1].size - 1, flags=engine.SEARCH_ASYNC,
                      stream=stream.cuda_stream, result=results[0])
    idx.search_device(batches[5][2].data_ptr(), batches[5][3].data_ptr(), batches[5][1].size - 1, flags=engine.SEARCH_ASYNC,
                      stream=stream.cuda_stream, result=results[0])
    for x, y in zip(results[0].host(), want[5]):
        assert np.array_equal(x, y)
    idx.search_device(batches[4][2].data_ptr(), batches[4][3].data_ptr(), batches[4][1].size - 1, flags=engine.SEARCH_ASYNC,
                      stream=stream.cuda_stream, result=results[1])
    results[1].close()
    torch.cuda.synchronize()


def test_pooled_results_and_results_that_outlive_their_index(engine, orc):
    """kmx_result_free parks results in the index's pool; a later search without a result of its own takes one over
    (buffers of another batch, possibly of another size and kind mix).  A result may be released after its index."""
    text = synth.ranks(78, 200_000, 4)
    idx = engine.Index(text, 4, [6, 9])
    oidx = orc.Index(text, 4, [6, 9])
    for rnd, nq in enumerate((300, 1, 5000, 17, 5000, 2)):
        qranks, qoff = synth.mixed_queries(900 + rnd, text, nq, [4, 6, 9, 12, 15, 18], 4)
        fresh = [idx.search(qranks, qoff, flags=engine.SEARCH_KEEP_MASKS if i == 1 else engine.SEARCH_DEFAULT) for i in range(3)]
        o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, mode=orc.MODE_INTENDED, n_threads=4)
        for r in fresh:
            ho, pos, st, _ = r.host()
            assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos) and np.array_equal(st, o_st), (rnd, nq)
            r.close()                                   # back to the pool
    last = idx.search(qranks, qoff)
    ho, pos, _, _ = last.host()
    idx.close()                                         # the pool closes with the index ...
    assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
    last.close()                                        # ... and a late result is simply destroyed


def test_skewed_text_giant_buckets(engine, orc):
    """Low-entropy text: a few giant buckets (load imbalance, runs far longer than a tile, many mask words)."""
    n = 120_000
    text = np.zeros(n, np.uint8)
    text[::997] = 1
    text[5::4001] = 2
    idx = engine.Index(text, 4, [5, 9])
    oidx = orc.Index(text, 4, [5, 9])
    qs = [np.zeros(m, np.uint8) for m in (2, 5, 7, 9, 10, 14, 18, 23)]
    qs += [text[990:990 + m].copy() for m in (5, 9, 12, 18)] + [text[n - m:].copy() for m in (3, 5, 9, 14)]
    qranks, qoff = pack(qs)
    r = idx.search(qranks, qoff)
    ho, pos, st, kd = r.host()
    o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=4)
    assert np.array_equal(st, o_st.astype(np.uint8))
    assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
    for i, q in enumerate(qs):
        assert np.array_equal(pos[int(ho[i]):int(ho[i + 1])], orc.naive_scan(text, q))


def test_multi_part_survivors_are_dropped_from_hits_and_masks(engine, orc):
    """Queries of several parts in a repetitive text: many candidates pass the one part k_validate filters with and
    fail another one, so k_validate_more has to drop them — from the hit list, from the count and from the
    compressed_bitset words (KEEP_MASKS).  Checked against the naive scan and for mask/hit consistency."""
    rng = np.random.default_rng(5)
    motif = rng.integers(0, 4, 61).astype(np.uint8)
    text = np.tile(motif, 3000)
    mut = rng.integers(0, text.size, text.size // 23)
    text[mut] = rng.integers(0, 4, mut.size)              # mutations break most long matches but few k-mers
    text = np.ascontiguousarray(text)
    for ks in ([7], [6, 9, 11]):
        idx = engine.Index(text, 4, ks, keep_host_arena=True)
        qs = []
        for m in (21, 28, 30, 45, 66, 100):
            for s0 in (0, 13, 61 * 40 + 7, 61 * 1500 + 30):
                qs.append(text[s0:s0 + m].copy())
                qs.append(np.tile(motif, 3)[s0 % 61:s0 % 61 + m].copy())      # the unmutated repeat
        qranks, qoff = pack(qs)
        r = idx.search(qranks, qoff, flags=engine.SEARCH_KEEP_MASKS)
        ho, pos, st, kd = r.host()
        base, words_ptr, cand_cnt, cand_src = r.masks()
        arena = idx.arena_host()
        import ctypes as C
        dropped = 0
        for i, q in enumerate(qs):
            want = orc.naive_scan(text, q)
            got = pos[int(ho[i]):int(ho[i + 1])]
            assert np.array_equal(got, want), (ks, i, len(q))
            if kd[i] != engine.KIND_STITCH:
                continue
            nw = cand_cnt[i] // 64 + 1
            words = np.ctypeslib.as_array(C.cast(words_ptr, C.POINTER(C.c_uint64)), shape=(int(base[i]) + nw,))[int(base[i]):]
            bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:cand_cnt[i]].astype(bool)
            cands = arena[int(cand_src[i]):int(cand_src[i]) + int(cand_cnt[i])]
            assert np.array_equal(cands[bits], got), (ks, i, len(q))
            dropped += int(cand_cnt[i]) - got.size
        assert dropped > 1000                                # the text really makes candidates fail
        # the count-only form runs the same validation
        rc = idx.search(qranks, qoff, flags=engine.SEARCH_COUNT_ONLY)
        assert np.array_equal(rc.host()[0], ho)


def test_many_reads_with_few_to_ten_parts_take_the_thread_per_query_check(engine, orc):
    """k_validate_more_thread: in a batch with >= 2^18 multi-part queries, those with up to 10 further parts are checked one
    THREAD per query (up to 4 parts in any batch).  Reads of 26..66 letters against k = 6 (5..11 parts), half of them with one
    letter changed — the true start passes the filter part and fails another one, so survivors are dropped from the hit list,
    the count and the mask words.  Against the oracle in full, masks against hits on a sample."""
    import ctypes as C
    rng = np.random.default_rng(99)
    n, k, nq = 300_000, 6, 300_000
    text = synth.ranks(4242, n, 4)
    lens = rng.integers(26, 67, nq)
    starts = rng.integers(0, n - 70, nq)
    off = np.zeros(nq + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    rep = np.repeat(np.arange(nq), lens)
    within = np.arange(int(off[-1])) - np.repeat(off[:-1].astype(np.int64), lens)
    qr = text[starts[rep] + within].copy()
    changed = np.nonzero(rng.random(nq) < 0.5)[0]
    at = off[changed].astype(np.int64) + (rng.random(changed.size) * lens[changed]).astype(np.int64)
    qr[at] = (qr[at] + 1 + rng.integers(0, 3, changed.size)) % 4
    idx = engine.Index(text, 4, [k], keep_host_arena=True)
    r = idx.search(qr, off, flags=engine.SEARCH_KEEP_MASKS)
    ho, pos, st, kd = r.host()
    assert (kd == engine.KIND_STITCH).sum() > (1 << 18)
    oidx = orc.Index(text, 4, [k])
    o_off, o_pos, o_st, _ = oidx.search_batch(qr, off, n_threads=8)
    assert np.array_equal(st, o_st.astype(np.uint8))
    assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
    base, words_ptr, cand_cnt, cand_src = r.masks()
    arena = idx.arena_host()
    for i in rng.integers(0, nq, 3000):
        if kd[i] != engine.KIND_STITCH:
            continue
        nw = int(cand_cnt[i]) // 64 + 1
        words = np.ctypeslib.as_array(C.cast(words_ptr, C.POINTER(C.c_uint64)), shape=(int(base[i]) + nw,))[int(base[i]):]
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:cand_cnt[i]].astype(bool)
        cands = arena[int(cand_src[i]):int(cand_src[i]) + int(cand_cnt[i])]
        assert np.array_equal(cands[bits], pos[int(ho[i]):int(ho[i + 1])]), i
    # the same batch without masks (survivor lists only) and count-only
    r2 = idx.search(qr, off)
    assert np.array_equal(r2.host()[0], ho) and np.array_equal(r2.host()[1], pos)
    rc = idx.search(qr, off, flags=engine.SEARCH_COUNT_ONLY)
    assert np.array_equal(rc.host()[0], ho)
    # a small batch of the same reads (group-per-survivor path for the long ones) agrees
    sub = 5000
    r3 = idx.search(qr[:int(off[sub])], off[:sub + 1])
    assert np.array_equal(r3.host()[0], ho[:sub + 1]) and np.array_equal(r3.host()[1], pos[:int(ho[sub])])
    for x in (r, r2, rc, r3):
        x.close()
    idx.close()


@pytest.mark.parametrize("ks", [[6], [6, 9], [5]])
def test_filter_buckets_of_a_few_hundred_entries_take_the_wide_path(engine, orc, ks):
    """k_validate_wide: a filter bucket of 257..2048 entries does not fit the stage of a 16-lane group; the query gets a wave
    with a stage of its own.  k = 6 on 1.5e6 letters: buckets of about 366 positions; k = 5: about 1465 (more candidates than
    KMX_VBIG, but a filter bucket of the same order: not a case for anchoring on the smallest bucket); reads of 7..40 letters
    (rest parts, two parts, further parts), planted, planted with one letter changed, random.  Oracle in full, masks against hits."""
    import ctypes as C
    rng = np.random.default_rng(31)
    n, nq = 1_500_000, 6000
    text = synth.ranks(777, n, 4)
    lens = rng.integers(7, 41, nq)
    starts = rng.integers(0, n - 50, nq)
    qs = []
    for i in range(nq):
        q = text[starts[i]:starts[i] + lens[i]].copy()
        kind = i % 3
        if kind == 1:
            j = int(rng.integers(0, lens[i]))
            q[j] = (q[j] + 1 + int(rng.integers(0, 3))) % 4
        elif kind == 2:
            q = rng.integers(0, 4, lens[i]).astype(np.uint8)
        qs.append(q)
    qranks, qoff = pack(qs)
    idx = engine.Index(text, 4, ks, keep_host_arena=True)
    oidx = orc.Index(text, 4, ks)
    o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=8)
    for flags in (engine.SEARCH_DEFAULT, engine.SEARCH_KEEP_MASKS, engine.SEARCH_COUNT_ONLY):
        r = idx.search(qranks, qoff, flags=flags)
        ho, pos, st, kd = r.host()
        assert np.array_equal(st, o_st.astype(np.uint8)) and np.array_equal(ho, o_off)
        if flags != engine.SEARCH_COUNT_ONLY:
            assert np.array_equal(pos, o_pos)
        if flags == engine.SEARCH_KEEP_MASKS:
            base, words_ptr, cand_cnt, cand_src = r.masks()
            arena = idx.arena_host()
            wide = 0
            for i in range(0, nq, 5):
                if kd[i] != engine.KIND_STITCH:
                    continue
                nw = int(cand_cnt[i]) // 64 + 1
                words = np.ctypeslib.as_array(C.cast(words_ptr, C.POINTER(C.c_uint64)), shape=(int(base[i]) + nw,))[int(base[i]):]
                bits = np.unpackbits(words.view(np.uint8), bitorder="little")
                assert not bits[cand_cnt[i]:nw * 64].any(), i            # the padding bits of the last word stay 0
                bits = bits[:cand_cnt[i]].astype(bool)
                cands = arena[int(cand_src[i]):int(cand_src[i]) + int(cand_cnt[i])]
                assert np.array_equal(cands[bits], pos[int(ho[i]):int(ho[i + 1])]), i
                wide += int(cand_cnt[i] > 256)
            assert wide > 100
        r.close()
    idx.close()


def test_concurrent_host_threads_share_one_index(engine, orc):
    """kmx.h: one index may be searched from several host threads at once (search() is const in the reference,
    kmer_index.hpp:505).  Host-buffer calls serialise on the index's internal stream; device-buffer calls run on
    their own streams with their own result handles."""
    import threading
    import torch
    text = synth.ranks(31337, 400_000, 4)
    idx = engine.Index(text, 4, [8, 11])
    oidx = orc.Index(text, 4, [8, 11])
    batches = []
    for t in range(6):
        q, off = make_queries(text, 4, [5, 8, 11, 16, 19, 22], 40, seed=100 + t)
        o = oidx.search_batch(q, off, n_threads=2)
        batches.append((q, off, o[0], o[1]))
    errors = []

    def host_worker(t):
        try:
            q, off, want_off, want_pos = batches[t]
            for _ in range(5):
                ho, pos, st, kd = idx.search(q, off).host()
                assert np.array_equal(ho, want_off) and np.array_equal(pos, want_pos)
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    # torch objects are created on the main thread (torch's lazy CUDA init is not re-entrant across threads)
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    dev_inputs = {}
    for t in range(1, 6, 2):
        q, off, _, _ = batches[t]
        dev_inputs[t] = (torch.cuda.Stream(device=dev), torch.from_numpy(q).to(dev), torch.from_numpy(off.view(np.int64)).to(dev))
    torch.cuda.synchronize()

    def device_worker(t):
        try:
            q, off, want_off, want_pos = batches[t]
            stream, d_q, d_off = dev_inputs[t]
            res = engine.Result()
            for _ in range(5):
                idx.search_device(d_q.data_ptr(), d_off.data_ptr(), off.size - 1, stream=stream.cuda_stream, result=res)
                ho, pos, st, kd = res.host()
                assert np.array_equal(ho, want_off) and np.array_equal(pos, want_pos)
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=host_worker if t % 2 == 0 else device_worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("table", ["open", "dense"])
def test_save_load_roundtrip(engine, orc, tmp_path, table):
    """kmx_index_save / kmx_index_load: the loaded image answers exactly like the built index; corruption is detected."""
    text = synth.ranks(2718, 150_000, 5)
    ks = [6, 9]
    tk = engine.TABLE_OPEN if table == "open" else engine.TABLE_DENSE
    idx = engine.Index(text, 5, ks, table=tk)
    qranks, qoff = make_queries(text, 5, [3, 6, 9, 12, 15, 18, 21], 30, seed=8)
    a = idx.search(qranks, qoff).host()
    path = tmp_path / "index.kmx"
    idx.save(str(path))
    idx.close()
    idx2 = engine.Index.load(str(path))
    assert idx2.info()["ks"] == ks and idx2.info()["n"] == text.size and idx2.info()["tables"] == [tk, tk]
    b = idx2.search(qranks, qoff).host()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    o_off, o_pos, _, _ = orc.Index(text, 5, ks).search_batch(qranks, qoff, n_threads=4)
    assert np.array_equal(b[0], o_off) and np.array_equal(b[1], o_pos)
    # flip one byte in the middle of the payload: checksum mismatch
    raw = bytearray(path.read_bytes())
    raw[len(raw) // 2] ^= 0x40
    bad = tmp_path / "bad.kmx"
    bad.write_bytes(bytes(raw))
    with pytest.raises(engine.KmxError) as e:
        engine.Index.load(str(bad))
    assert "checksum" in str(e.value) or "corrupt" in str(e.value)


def test_large_k_sort_based_flatten_and_rec64(engine, orc):
    """Key spaces far beyond a histogram (4^20, 4^31 keys): the sort-based flatten + open table; and the 64-bit
    LDS-record variant of k_fill that serves arenas >= 4 GiB (forced here through its test hook)."""
    import os
    text = synth.ranks(161, 150_000, 4)
    for ks, lengths in (([20], [9, 15, 20, 25, 40, 45]), ([31], [20, 31, 40, 62, 70]), ([16, 20], [16, 20, 32, 36, 40])):
        qranks, qoff = make_queries(text, 4, lengths, 24, seed=ks[0])
        oidx = orc.Index(text, 4, ks)
        o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=4)
        for rec64 in ("0", "1"):
            os.environ["KMX_FORCE_REC64"] = rec64
            try:
                idx = engine.Index(text, 4, ks)
            finally:
                os.environ.pop("KMX_FORCE_REC64", None)
            assert idx.info()["tables"] == [engine.TABLE_OPEN] * len(ks)
            ho, pos, st, kd = idx.search(qranks, qoff).host()
            assert np.array_equal(st, o_st.astype(np.uint8)), (ks, rec64)
            assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos), (ks, rec64)
            idx.close()
    # the rec64 variant on a batch with many tiles
    os.environ["KMX_FORCE_REC64"] = "1"
    try:
        idx = engine.Index(text, 4, [6])
    finally:
        os.environ.pop("KMX_FORCE_REC64", None)
    q, off = synth.uniform_queries(3, 50_000, 6, 4)
    ho, pos, st, kd = idx.search(q, off).host()
    o_off, o_pos, _, _ = orc.Index(text, 4, [6]).search_batch(q, off, n_threads=8)
    assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)


def test_prefix_sort_all_size_classes(engine, orc):
    """m < k slices of every class: few runs / short (wave-level LDS rank pass), mid-size or many runs (block-level
    bitonic sort in LDS), and beyond 32 K positions (global merge passes) — all equal the oracle."""
    text = synth.ranks(606, 600_000, 4)
    idx = engine.Index(text, 4, [10], prefix_levels=-1)           # no pre-merged levels: every slice is merged per query
    oidx = orc.Index(text, 4, [10])
    qs = []
    for m, cnt in ((9, 40), (8, 40), (7, 30), (6, 20), (5, 10), (4, 6), (3, 4), (2, 2)):     # 4 .. 65536 runs; 2 .. 37 K hits
        for t in range(cnt):
            s0 = (t * 7919 + m * 104729) % (text.size - m)
            qs.append(text[s0:s0 + m].copy())
    qs += [text[text.size - m:].copy() for m in (9, 7, 4)]                                      # with last-kmer positions
    qranks, qoff = pack(qs)
    idx.stats_enable(True)
    r = idx.search(qranks, qoff)
    ho, pos, st, kd = r.host()
    o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=4)
    assert (kd == engine.KIND_PREFIX).all()
    assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
    k = idx.stats()
    assert k["k_prefix_sort_small"]["launches"] and k["k_prefix_sort_block"]["launches"] and k["k_prefix_merge_pass"]["launches"]
    assert k["k_prefix_split"]["launches"]                    # (m = 2: 37.5 K positions in 65536 runs — spread by value, two bands)
    # the same classes with LONG runs (146 positions per 6-mer): 4 runs / 586 positions (wave-level merge), 16 runs / 2.3 K and
    # 64 runs / 9.4 K (the two shapes of the block-level merge), 256 and 1024 runs (chunks + merge passes)
    idx6 = engine.Index(text, 4, [6], prefix_levels=-1)
    oidx6 = orc.Index(text, 4, [6])
    qs = [text[s0:s0 + m].copy() for m, cnt in ((5, 60), (4, 40), (3, 20), (2, 6), (1, 2)) for s0 in range(1000, 1000 + 37 * cnt, 37)]
    qs += [text[text.size - m:].copy() for m in (5, 3)]
    qranks, qoff = pack(qs)
    idx6.stats_enable(True)
    ho, pos, st, kd = idx6.search(qranks, qoff).host()
    o_off, o_pos, o_st, _ = oidx6.search_batch(qranks, qoff, n_threads=4)
    assert (kd == engine.KIND_PREFIX).all()
    assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
    k = idx6.stats()
    assert k["k_prefix_merge_small"]["launches"] and k["k_prefix_sort_block"]["launches"] and k["k_prefix_merge_pass"]["launches"]
    assert k["k_prefix_split"]["launches"]                    # (m = 1: 150 K positions in 1024 runs — seven bands; m = 2: 256 runs, two)


@pytest.mark.parametrize("sigma,ks,levels", [(4, [10], 0), (4, [10], 1), (4, [10], 3), (4, [6, 9, 12], 0), (5, [8], 0), (20, [4], 0),
                                             (4, [9, 10], 3)])
def test_prefix_levels_answer_like_the_merge(engine, orc, sigma, ks, levels):
    """Prefix levels (kmx_options.prefix_levels): the lists of every (k - L)-mer, merged once when the index is installed.
    A sub-k query then copies one list (length k - L) or merges sigma^L times fewer of them (shorter ones) — same positions,
    same order, same last-kmer positions as the oracle, for every sub-k length of every element, and the same as the index
    without levels; the memory they cost shows in device_bytes; queries a level answers outright run no sort kernel."""
    text = synth.ranks(4242 + sigma, 300_000, sigma)
    with_lv = engine.Index(text, sigma, ks, prefix_levels=levels)
    without = engine.Index(text, sigma, ks, prefix_levels=-1)
    oidx = orc.Index(text, sigma, ks)
    assert with_lv.info()["device_bytes"] >= without.info()["device_bytes"] + 4 * (text.size - max(ks))
    mw, mo = with_lv.memory(), without.memory()
    assert mo["prefix_levels"] == 0 and mw["prefix_levels"] >= 4 * (text.size - max(ks)) and mw["positions"] == mo["positions"] == 4 * sum(text.size - k + 1 for k in ks)
    assert sum(mw.values()) == with_lv.info()["device_bytes"] and sum(mo.values()) == without.info()["device_bytes"]
    qs = []
    rng = np.random.default_rng(7)
    for m in range(1, max(ks)):
        if sigma ** (min(k for k in ks if k >= m) - m) > 10_000:       # keep the oracle's fan-out cheap
            continue
        for t in range(6):
            s0 = int(rng.integers(0, text.size - m))
            qs.append(text[s0:s0 + m].copy())
        qs.append(text[text.size - m:].copy())                         # ends the text: last-kmer positions
        qs.append(text[text.size - m - 1:text.size - 1].copy())
        qs.append(rng.integers(0, sigma, m).astype(np.uint8))
    qranks, qoff = pack(qs)
    o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=4)
    for idx in (with_lv, without):
        ho, pos, st, kd = idx.search(qranks, qoff).host()
        assert np.array_equal(st, o_st.astype(np.uint8))
        assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
    # a batch of nothing but (k - 1)-letter queries on the smallest k: answered from level 1, nothing to sort
    k0 = min(ks)
    if k0 > 1:
        sub = [q for q in qs if len(q) == k0 - 1] * 1000           # (more than the one-launch latency path takes)
        sq, so = pack(sub)
        with_lv.stats_enable(True)
        with_lv.stats_reset()
        r = with_lv.search(sq, so)
        kst = with_lv.stats()
        assert kst["k_fill"]["launches"] and not kst["k_small"]["launches"]
        assert not any(kst[name]["launches"] for name in ("k_prefix_sort_small", "k_prefix_merge_small", "k_prefix_sort_block", "k_prefix_merge_pass"))
        a = r.host()
        b = without.search(sq, so).host()
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    with_lv.close()
    without.close()


def test_prefix_slices_of_one_long_run_and_of_lopsided_runs(engine, orc):
    """A periodic text: a sub-k query is answered by ONE run of tens of thousands of positions (in order as it lies — no
    sort kernel, no merge pass may touch it), or, where the period was broken, by one long run plus runs of a single
    position before or behind it; two interleaved phases give runs that alternate element by element."""
    unit = np.array([0, 1, 2], np.uint8)
    text = np.tile(unit, 40_000)
    text[77_001] = 3                                   # a broken period: one-position runs next to the long ones
    text[5] = 3
    for ks, levels in (([12], -1), ([9], -1), ([9], 0)):
        idx = engine.Index(text, 4, ks, table=engine.TABLE_OPEN if levels else engine.TABLE_DENSE, prefix_levels=levels)
        oidx = orc.Index(text, 4, ks)
        qs = [text[s0:s0 + m].copy() for m in (1, 2, 3, 5, 8) for s0 in (0, 1, 2, 30, 76_995, 77_000, text.size - 8)]
        qranks, qoff = pack(qs)
        ho, pos, st, kd = idx.search(qranks, qoff).host()
        o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=4)
        assert (kd == engine.KIND_PREFIX).all()
        assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
        idx.close()


def test_queries_of_very_many_parts_take_a_wave_each(engine, orc):
    """Queries of more than KMX_LONG_PARTS (256) parts: walked by one lane in the first batch on a handle, listed by k_lookup and
    taken by k_lookup_long (a wave per query, a lane per part) from the second batch on — the same statuses, kinds and lists:
    planted reads, reads with one changed letter (the walk stops at a missing part), a letter outside the alphabet before and
    behind the first missing part (kmer_index.hpp:216-227: only a part the walk reaches counts), a rest whose prefix range would
    throw (:119-122 via :234) with and without a missing part in front of it."""
    rng = np.random.default_rng(5)
    for sigma, k, n in ((4, 8, 300_000), (4, 13, 300_000), (20, 3, 100_000)):
        text = synth.ranks(900 + k, n, sigma)
        qs = []
        for m in (k * 257, k * 257 + 1, k * 300 + k // 2, k * 320, k * 321 + 1, k * 530 + 2, k * 100):
            for t in range(6):
                s0 = int(rng.integers(0, n - m))
                q = text[s0:s0 + m].copy()
                if t == 1:
                    q[int(rng.integers(0, m))] = (int(q[0]) + 1 + int(rng.integers(0, sigma - 1))) % sigma     # some part goes missing (or not)
                elif t == 2:
                    q[m // 2] = 250                                                   # a letter outside the alphabet, nothing missing in front of it
                elif t == 3:
                    q[5] = (int(q[5]) + 1) % sigma                                    # the first part is (very likely) missing ...
                    q[m - 3] = 250                                                    # ... so the bad letter near the end is never reached
                elif t == 4:
                    q[m - 1] = 250                                                    # in the rest / the last part
                qs.append(q)
        qs += [text[7:7 + k * 20].copy(), text[100:100 + k].copy()]                  # (ordinary queries beside them)
        assert max(len(q) for q in qs) < 10000
        qranks, qoff = pack(qs)
        idx = engine.Index(text, sigma, [k])
        oidx = orc.Index(text, sigma, [k])
        clean = qranks.copy()
        clean[clean >= sigma] = 0
        o_off, o_pos, o_st, _ = oidx.search_batch(clean, qoff, mode=orc.MODE_INTENDED, n_threads=4)     # (the oracle has no bad letters: statuses of those queries are the engine's own)
        res = engine.Result()
        outs = []
        for rep in range(3):
            idx.stats_enable(True)
            idx.search(np.tile(qranks, 1), qoff, result=res, flags=engine.SEARCH_KEEP_MASKS if rep == 2 else engine.SEARCH_DEFAULT)
            outs.append(res.host())
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b), (sigma, k)                                  # one lane per query == one wave per query
        assert np.array_equal(outs[2][0], outs[0][0]) and np.array_equal(outs[2][1], outs[0][1]) and np.array_equal(outs[2][2], outs[0][2])
        ho, pos, st, kd = outs[0]
        bad = np.array([bool((q >= sigma).any()) for q in qs])
        # KMX_Q_BAD_RANK; OK when an earlier part was missing; SUBK_FANOUT when the bad letter sits in a rest that is never looked at
        assert set(st[bad].tolist()) <= {0, 2, 4}
        assert (st[bad] == 4).any() and (k < 13 or (st[bad] == 0).any())         # (parts only go missing where the k-mers are sparse)
        good = ~bad
        assert np.array_equal(st[good], o_st[good].astype(np.uint8))
        for i in np.nonzero(good)[0]:
            assert np.array_equal(pos[int(ho[i]):int(ho[i + 1])], o_pos[int(o_off[i]):int(o_off[i + 1])]), (sigma, k, i)
            if st[i] == 0 and i % 5 == 0:
                assert np.array_equal(pos[int(ho[i]):int(ho[i + 1])], orc.naive_scan(text, qs[i]))
        assert (np.diff(ho)[bad] == 0).all()
        if k == 13:
            assert (st == 2).any()                                                   # KMX_Q_SUBK_FANOUT through a rest of one letter
        idx.close()


def test_prefix_slices_whose_positions_cluster(engine, orc):
    """Big sub-k slices whose positions crowd into a small part of the text — a long homopolymer (one run of consecutive
    positions), a periodic region (every 8th position, several runs) — through the block-level merge and the merge passes; with
    k = 8 the slices have hundreds of runs: the distribution sort of such chunks (distribute_sort_lds) meets value buckets of more
    than KMX_PBK_GIVE_UP positions there and hands the chunk to the bitonic network."""
    rng = np.random.default_rng(11)
    text = rng.integers(0, 4, 700_000).astype(np.uint8)
    text[100_000:220_000] = 0                                           # 120 000 x 'A': consecutive positions in ONE run
    text[400_000:460_000] = np.tile(np.array([0, 1, 0, 2, 0, 3, 0, 0], np.uint8), 7_500)    # period 8: every 8th position, several runs
    for ks in ([6], [8]):
        idx = engine.Index(text, 4, ks, prefix_levels=-1)
        oidx = orc.Index(text, 4, ks)
        qs = [np.array(q, np.uint8) for q in ([0], [0, 0], [0, 0, 0], [0, 1], [0, 1, 0], [1, 0, 2], [0, 0, 0, 0], [3], [2, 0], [0, 3, 0, 0])]
        qs += [text[s0:s0 + m].copy() for m in (1, 2, 3, 4) for s0 in (5, 150_000, 410_003, 650_000)]
        qranks, qoff = pack(qs)
        idx.stats_enable(True)
        ho, pos, st, kd = idx.search(qranks, qoff).host()
        o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=4)
        assert (kd == engine.KIND_PREFIX).all()
        assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos), ks
        assert idx.stats()["k_prefix_sort_block"]["launches"]
        idx.close()


def test_prefix_slices_cut_into_bands(engine, orc):
    """Slices beyond a chunk (32768 positions) of at most 64 runs are cut into value bands (k_prefix_bands / k_prefix_merge_band): on a
    text whose k-mers occur all over it every band fits a 256-thread block and no merge pass runs over the slice; where the occurrences
    of a prefix crowd into a part of the text (here: the first quarter is written in two letters, the rest in the other two) a band
    would overflow and the slice stays with the chunks + merge passes.  Both equal the oracle, and the two kinds share one batch."""
    rng = np.random.default_rng(23)
    n = 3_000_000
    even = rng.integers(0, 4, n).astype(np.uint8)
    skew = np.concatenate([rng.integers(0, 2, n // 4), rng.integers(2, 4, n - n // 4)]).astype(np.uint8)
    # the first fifth of the text without the letter A: the bands of 'AAA' there are empty, the others still fit
    gap = rng.integers(0, 4, 2_720_000).astype(np.uint8)
    gap[:544_000] = rng.integers(1, 4, 544_000)
    for text, sigma, ks in ((even, 4, [6]), (skew, 4, [6]), (even % 2, 2, [12]), (gap, 4, [6])):
        n = text.size
        idx = engine.Index(text, sigma, ks, prefix_levels=-1)
        oidx = orc.Index(text, sigma, ks)
        k = ks[0]
        # m = k - 3 on 4 letters: 64 runs, n / 64 = 46.9 K positions (8 bands); k - 2: 16 runs of 11.7 K (one chunk); on 2 letters
        # m = k - 6: 64 runs, 46.9 K positions; m = k - 5: 32 runs, 23.4 K; m = k - 4: 16 runs, 11.7 K
        lens = (k - 3, k - 2, k - 1) if sigma == 4 else (k - 6, k - 5, k - 4, k - 2)
        qs = [text[s0:s0 + m].copy() for m in lens for s0 in range(1000, 1000 + 53 * 24, 53)]
        qs += [rng.integers(0, sigma, m).astype(np.uint8) for m in lens for _ in range(8)]
        qs += [text[n - m:].copy() for m in lens]                                               # with last-kmer positions
        qs += [np.zeros(m, np.uint8) for m in lens]                                              # A...A
        qranks, qoff = pack(qs)
        idx.stats_enable(True)
        r = idx.search(qranks, qoff)
        ho, pos, st, kd = r.host()
        o_off, o_pos, o_st, _ = oidx.search_batch(qranks, qoff, n_threads=8)
        assert np.array_equal(st, o_st.astype(np.uint8))
        assert np.array_equal(ho, o_off) and np.array_equal(pos, o_pos)
        assert int(np.diff(ho).max()) > 40_000
        assert idx.stats()["k_prefix_bands"]["launches"]
        ho2, pos2, _, _ = idx.search(qranks, qoff, result=r).host()                              # the same handle again (buffers, counters)
        assert np.array_equal(ho2, o_off) and np.array_equal(pos2, o_pos)
        idx.close()


@pytest.mark.parametrize("sigma,ks", [(4, [16]), (4, [14, 20, 31]), (2, [40, 63]), (20, [7, 12]), (5, [12, 4])])
def test_sorted_pairs_device_build_equals_host_flatten(engine, orc, sigma, ks, tmp_path):
    """Elements whose key space is beyond the histogram path (sigma^k > 2^26) are built on the device from sorted
    (hash, position) pairs (kmx_build_sort.hip): same arena as the host flatten, same answers — exact, stitched
    and prefix queries, which read the slots, the candidates and the sorted key / offset arrays — and the image
    written from the device-built tables loads back."""
    rng = np.random.default_rng(sigma * 7 + ks[0])
    text = (rng.integers(0, sigma, 300_000) * (rng.integers(0, 4, 300_000) != 0)).astype(np.uint8)   # biased: repeats exist
    text[100_000:100_400] = text[:400]
    dev = engine.Index(text, sigma, ks, keep_host_arena=True)
    host = engine.Index(text, sigma, ks, keep_host_arena=True, host_flatten=True)
    assert np.array_equal(dev.arena_host(), host.arena_host())
    assert dev.info()["tables"] == host.info()["tables"]
    kmax = max(ks)
    lens = sorted({max(1, min(ks) - 2), min(ks), kmax - 1, kmax, kmax + 3, 2 * kmax, 3 * kmax + 1})
    qranks, qoff = make_queries(text, sigma, lens, 24, seed=3)
    a = dev.search(qranks, qoff).host()
    b = host.search(qranks, qoff).host()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    o_off, o_pos, o_st, _ = orc.Index(text, sigma, ks).search_batch(qranks, qoff, mode=orc.MODE_INTENDED, n_threads=4)
    assert np.array_equal(a[0], o_off) and np.array_equal(a[1], o_pos) and np.array_equal(a[2], o_st.astype(np.uint8))
    assert int(o_off[-1]) > 30
    path = str(tmp_path / "sparse.img")
    dev.save(path)
    again = engine.Index.load(path)
    c = again.search(qranks, qoff).host()
    for x, y in zip(a, c):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("table", ["open", "dense"])
def test_device_built_index_equals_host_flatten(engine, orc, table):
    """Index construction on the device (k_build_*, k_bucket_sort_*) yields the same arena as the host flatten
    (positions grouped by rank-hash, ascending inside a bucket) and the same answers."""
    text = synth.ranks(515, 500_000, 4)
    ks = [4, 9, 12]
    tk = engine.TABLE_OPEN if table == "open" else engine.TABLE_DENSE
    dev = engine.Index(text, 4, ks, table=tk, keep_host_arena=True)
    host = engine.Index(text, 4, ks, table=tk, keep_host_arena=True, host_flatten=True)
    assert np.array_equal(dev.arena_host(), host.arena_host())
    assert dev.info()["tables"] == host.info()["tables"] == [tk] * 3
    qranks, qoff = make_queries(text, 4, [2, 4, 6, 9, 12, 13, 18, 21, 24], 30, seed=12)
    a = dev.search(qranks, qoff).host()
    b = host.search(qranks, qoff).host()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    o_off, o_pos, _, _ = orc.Index(text, 4, ks).search_batch(qranks, qoff, n_threads=4)
    assert np.array_equal(a[0], o_off) and np.array_equal(a[1], o_pos)
    # the line-aligned second copy of long buckets is an option, not a semantic: same answers without it
    plain = engine.Index(text, 4, ks, table=tk, aligned_copy=False)
    assert plain.info()["device_bytes"] < dev.info()["device_bytes"]
    c = plain.search(qranks, qoff).host()
    for x, y in zip(a, c):
        assert np.array_equal(x, y)
    # a text with a bucket beyond the LDS sorts' capacity: that element's positions come from sorted (hash, position) pairs
    skew = np.zeros(100_000, np.uint8)
    skew[::3] = 1
    d2 = engine.Index(skew, 4, [6, 3], keep_host_arena=True)
    h2 = engine.Index(skew, 4, [6, 3], keep_host_arena=True, host_flatten=True)
    assert np.array_equal(d2.arena_host(), h2.arena_host())
    # ranks outside the alphabet are refused
    bad = text.copy()
    bad[1234] = 7
    with pytest.raises(engine.KmxError):
        engine.Index(bad, 4, [5])
