"""The latency path (k_small): a batch of up to 256 queries through the host-buffer entry point is ONE launch
that reads the queries from and writes the complete result to a page-locked block.  Its results — hit lists, statuses,
kinds, candidate runs and compressed_bitset words — must equal the general pipeline's bit for bit; a batch that is not made
for it falls back to the general pipeline on its own."""
import ctypes as C

import numpy as np
import pytest

from kmer_index_amd import synth
from tests.helpers import make_queries, pack

pytestmark = pytest.mark.gpu


def _lists(off, pos):
    return [pos[int(off[i]):int(off[i + 1])] for i in range(off.size - 1)]


def _words(ptr, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(max(int(n), 1),)).copy()


@pytest.mark.parametrize("sigma,ks,table", [(4, [8, 10, 12], "auto"), (4, [6], "open"), (5, [7, 9], "auto"), (20, [3, 4], "open")])
def test_small_batches_equal_the_general_pipeline(engine, sigma, ks, table):
    text = synth.ranks(1003 + sigma, 300_000, sigma)
    kmax = max(ks)
    lengths = sorted(set([1, 2, max(1, ks[0] - 2), ks[0], ks[0] + 1, kmax, kmax + 3, 2 * ks[0], 2 * kmax, 2 * kmax + 1, 3 * kmax + 2]))
    qranks, qoff = make_queries(text, sigma, lengths, 60, seed=7)          # random / planted / tail-planted per length
    nq = qoff.size - 1
    assert nq > 256
    idx = engine.Index(text, sigma, ks, table=engine.TABLE_OPEN if table == "open" else engine.TABLE_AUTO, keep_host_arena=True)
    idx.stats_enable(True)
    big = idx.search(qranks, qoff, flags=engine.SEARCH_KEEP_MASKS)          # > 256 queries: the general pipeline
    assert idx.stats()["k_small"]["launches"] == 0
    g_off, g_pos, g_st, g_kd = big.host()
    g_base, g_wptr, g_ccnt, g_csrc = big.masks()
    g_words = _words(g_wptr, int((g_base + g_ccnt // 64 + 1)[g_kd == engine.KIND_STITCH].max()) if (g_kd == engine.KIND_STITCH).any() else 1)
    want = _lists(g_off, g_pos)
    assert (g_kd == engine.KIND_STITCH).sum() > 20 and (g_kd == engine.KIND_PREFIX).sum() > 20 and (g_kd == engine.KIND_EXACT).sum() > 20
    rng = np.random.default_rng(5)
    order = rng.permutation(nq)
    res = engine.Result()
    n_small = 0
    at = 0
    for size in [1, 1, 1, 2, 3, 5, 8, 13, 64, 255, 256, 100, 1, 7]:
        sel = order[at:at + size] if at + size <= nq else order[:size]
        at += size
        q, off = pack([qranks[int(qoff[i]):int(qoff[i + 1])] for i in sel])
        before = idx.stats()["k_small"]["launches"]
        r = idx.search(q, off, flags=engine.SEARCH_KEEP_MASKS, result=res)
        n_small += idx.stats()["k_small"]["launches"] - before
        h_off, h_pos, h_st, h_kd = r.host()
        assert np.array_equal(h_st, g_st[sel]) and np.array_equal(h_kd, g_kd[sel])
        for j, i in enumerate(sel):
            assert np.array_equal(h_pos[int(h_off[j]):int(h_off[j + 1])], want[i]), (size, j, i)
        c = r.counts()
        assert c["nq"] == size and c["n_hits"] == h_pos.size and c["n_stitch"] == int((h_kd == engine.KIND_STITCH).sum())
        assert c["n_prefix"] == int((h_kd == engine.KIND_PREFIX).sum()) and c["n_error"] == int((h_st != 0).sum())
        base, wptr, ccnt, csrc = r.masks()
        st = np.nonzero(h_kd == engine.KIND_STITCH)[0]
        if st.size:
            words = _words(wptr, int((base + ccnt // 64 + 1)[st].max()))
            for j in st:
                i = sel[j]
                nw = int(ccnt[j]) // 64 + 1
                assert ccnt[j] == g_ccnt[i] and csrc[j] == g_csrc[i]
                assert np.array_equal(words[int(base[j]):int(base[j]) + nw], g_words[int(g_base[i]):int(g_base[i]) + nw])
    assert n_small >= 9, "the small batches did not take the latency path"
    # 256 queries of the index's own lengths (exact lookups) in one launch
    lens = np.diff(qoff).astype(np.int64)
    plain = np.nonzero(np.isin(lens, ks))[0]
    plain = np.resize(plain, 256)
    q, off = pack([qranks[int(qoff[i]):int(qoff[i + 1])] for i in plain])
    before = idx.stats()
    r = idx.search(q, off, result=res)
    h_off, h_pos, h_st, h_kd = r.host()
    total = sum(want[i].size for i in plain)
    if total <= 49152:
        after = idx.stats()
        assert after["k_small"]["launches"] == before["k_small"]["launches"] + 1 and after["k_lookup"]["launches"] == before["k_lookup"]["launches"]
    for j, i in enumerate(plain):
        assert np.array_equal(h_pos[int(h_off[j]):int(h_off[j + 1])], want[i]) and h_kd[j] == g_kd[i]
    # several workgroups in one launch (up to 32 x 256 queries): mostly plain lookups, every 17th query a slow one
    slow = np.nonzero(~np.isin(lens, ks) & (g_st == 0))[0]
    for total_q in (257, 700, 3000, 8192):
        pick = np.resize(plain, total_q).copy()
        pick[::17] = np.resize(slow, pick[::17].size)
        q, off = pack([qranks[int(qoff[i]):int(qoff[i + 1])] for i in pick])
        before = idx.stats()
        r = idx.search(q, off, flags=engine.SEARCH_KEEP_MASKS, result=res)
        after = idx.stats()
        h_off, h_pos, h_st, h_kd = r.host()
        blocks_ok = all(sum(want[i].size for i in pick[b:b + 256]) <= 49152 for b in range(0, total_q, 256))
        if blocks_ok:
            assert after["k_small"]["launches"] == before["k_small"]["launches"] + 1 and after["k_lookup"]["launches"] == before["k_lookup"]["launches"], total_q
        assert np.array_equal(h_st, g_st[pick]) and np.array_equal(h_kd, g_kd[pick])
        assert np.array_equal(np.diff(h_off), np.array([want[i].size for i in pick], np.uint64))
        assert np.array_equal(h_pos, np.concatenate([want[i] for i in pick]))
        base, wptr, ccnt, csrc = r.masks()
        stq = np.nonzero(h_kd == engine.KIND_STITCH)[0]
        words = _words(wptr, int((base + ccnt // 64 + 1)[stq].max())) if stq.size else None
        for j in stq[::5]:
            i = pick[j]
            nw = int(ccnt[j]) // 64 + 1
            assert ccnt[j] == g_ccnt[i] and csrc[j] == g_csrc[i]
            assert np.array_equal(words[int(base[j]):int(base[j]) + nw], g_words[int(g_base[i]):int(g_base[i]) + nw])
        c = r.counts()
        assert c["nq"] == total_q and c["n_hits"] == h_pos.size and c["n_stitch"] == stq.size
    # device views of a result that was produced on the latency path: materialised on demand, same contents
    import torch
    q, off = pack([qranks[int(qoff[i]):int(qoff[i + 1])] for i in order[:9]])
    r = idx.search(q, off, result=res)
    host = r.host()
    t_off, t_pos = r.device_tensors(torch.device("cuda", 0))
    torch.cuda.synchronize()
    assert np.array_equal(t_off.cpu().numpy().astype(np.uint64), host[0]) and np.array_equal(t_pos.cpu().numpy().view(np.uint32), host[1])
    res.close()
    big.close()
    idx.close()


def test_batches_the_small_kernel_declines_fall_back(engine, orc):
    """More hits than the mailbox holds, long candidate lists, many cross-referenced queries: same answers, general pipeline."""
    text = synth.ranks(3, 2_000_000, 4)
    idx = engine.Index(text, 4, [4, 6], prefix_levels=-1)          # (sub-k slices are merged per query; the reference's planner below)
    idx.stats_enable(True)
    oidx = orc.Index(text, 4, [4, 6])
    cases = {
        "too many hits": pack([text[100:104]] * 30),                                    # 30 x ~7800 hits
        "long candidate list": pack([text[500:512]]),                                   # 6 + 6: ~490 candidates (fine) ...
        "huge candidate list": pack([text[500:508]]),                                   # 4 + 4: ~7800 candidates > 4096
        "many stitch queries": pack([text[s:s + 13] for s in range(1000, 1045)]),       # 45 cross-referenced queries: the host does not even try
        "long prefix slice": pack([text[700:703]]),                                     # m = 3 < 4: 4 runs, ~31000 positions
        # three workgroups, the second of which finds more hits than it may write: it declines (and still publishes its
        # totals, so the third does not wait for ever), the whole batch goes to the general pipeline
        "a workgroup in the middle declines": pack([text[s:s + 6] for s in range(2000, 2256)] + [text[100:104]] * 30 +
                                                   [text[s:s + 6] for s in range(3000, 3400)]),
    }
    for name, (q, off) in cases.items():
        r = idx.search(q, off, flags=engine.SEARCH_REFERENCE_PLAN)
        o_off, o_pos, o_st, _ = oidx.search_batch(q, off, mode=orc.MODE_INTENDED)
        h = r.host()
        assert np.array_equal(h[0], o_off) and np.array_equal(h[1], o_pos) and np.array_equal(h[2], o_st.astype(np.uint8)), name
        r.close()
    st = idx.stats()
    assert st["k_small"]["launches"] == len(cases) - 1 and st["k_lookup"]["launches"] >= 4    # tried (but for the 45), declined, served by the general path
    idx.close()


def test_batch_of_nothing_but_empty_queries(engine):
    """Found by the fuzz soak of round 2: a batch whose queries are all empty has no letters — a NULL letter pointer is fine."""
    text = synth.ranks(9, 10_000, 4)
    idx = engine.Index(text, 4, [5])
    for nq in (1, 3, 300):
        r = idx.search(np.zeros(0, np.uint8), np.zeros(nq + 1, np.uint64))
        h = r.host()
        assert h[0].tolist() == [0] * (nq + 1) and (h[2] == engine.Q_EMPTY_QUERY).all() and r.counts()["n_error"] == nq
        r.close()
    idx.close()


def test_device_view_of_a_latency_path_result_after_its_index_is_gone(engine):
    """A result produced by k_small lives in host memory; its device views are materialised on demand from the index — which
    must still exist: asking after kmx_index_free is an error, not a use of freed memory; the host views stay valid."""
    text = synth.ranks(9, 50_000, 4)
    idx = engine.Index(text, 4, [6])
    q, off = synth.uniform_queries(3, 5, 6, 4)
    r = idx.search(q, off)
    want = r.host()
    idx.close()
    with pytest.raises(engine.KmxError):
        r.device_ptrs()
    assert all(np.array_equal(a, b) for a, b in zip(r.host(), want))
    r.close()


def test_latency_path_from_concurrent_host_threads(engine):
    """Several host threads share one index and search one query (or a few) at a time — the reference's call shape, concurrent:
    every call runs on its result's own stream and mailbox."""
    import threading
    text = synth.ranks(77, 300_000, 4)
    idx = engine.Index(text, 4, [8, 10])
    q, off = make_queries(text, 4, [6, 8, 10, 13, 18, 20, 25], 80, seed=5)
    nq = off.size - 1
    big = idx.search(q, off).host()                                       # general pipeline (560 queries)
    want = _lists(big[0], big[1])
    errors = []

    def worker(t):
        try:
            rng = np.random.default_rng(t)
            res = engine.Result()
            for it in range(120):
                size = 1 if it % 3 else int(rng.integers(2, 20))
                b = int(rng.integers(0, nq - size + 1))
                sq, so = pack([q[int(off[i]):int(off[i + 1])] for i in range(b, b + size)])
                h = idx.search(sq, so, result=res).host()
                for j in range(size):
                    assert np.array_equal(h[1][int(h[0][j]):int(h[0][j + 1])], want[b + j]) and h[2][j] == big[2][b + j]
            res.close()
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    idx.stats_enable(True)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert idx.stats()["k_small"]["launches"] >= 8 * 100
    idx.close()


@pytest.mark.parametrize("sigma,ks", [(4, [8, 10, 12]), (5, [9]), (20, [4])])
def test_latency_path_against_the_oracle_directly(engine, orc, sigma, ks):
    """k_small's results checked against the CPU oracle itself (not against the general pipeline): batches of 1, 7, 64 and
    256 queries of every kind — exact, sub-k (levels and merged slices), stitched, multi-k, absent, rejected — each one
    launch of the latency path; inside the envelope of SURVEY 4.3 also against the faithful restatement of the reference."""
    from tests.helpers import inside_envelope
    text = synth.ranks(2024 + sigma, 200_000, sigma)
    kmax = max(ks)
    lengths = sorted(set([1, 2, max(1, ks[0] - 2), ks[0] - 1, ks[0], ks[0] + 1, kmax, kmax + 3, 2 * ks[0], 2 * kmax + 1, 3 * kmax]))
    qranks, qoff = make_queries(text, sigma, lengths, 30, seed=11)
    nq = qoff.size - 1
    oidx = orc.Index(text, sigma, ks)
    plan = orc.plan(ks)
    rng = np.random.default_rng(3)
    for levels in (0, -1):
        idx = engine.Index(text, sigma, ks, prefix_levels=levels)
        idx.stats_enable(True)
        res = engine.Result()
        for size in (1, 7, 64, 256, 1, 1):
            sel = rng.choice(nq, size=size, replace=False)
            q, off = pack([qranks[int(qoff[i]):int(qoff[i + 1])] for i in sel])
            before = idx.stats()["k_small"]["launches"]
            h_off, h_pos, h_st, h_kd = idx.search(q, off, result=res).host()
            took_small = idx.stats()["k_small"]["launches"] - before
            o_off, o_pos, o_st, _ = oidx.search_batch(q, off, mode=orc.MODE_INTENDED, n_threads=4)
            assert np.array_equal(h_st, o_st.astype(np.uint8)) and np.array_equal(h_off, o_off) and np.array_equal(h_pos, o_pos), (levels, size)
            f_off, f_pos, f_st, _ = oidx.search_batch(q, off, mode=orc.MODE_FAITHFUL, n_threads=4)
            for j in range(size):
                if inside_envelope(plan, ks, int(off[j + 1] - off[j])):
                    assert h_st[j] == f_st[j] and np.array_equal(h_pos[int(h_off[j]):int(h_off[j + 1])], f_pos[int(f_off[j]):int(f_off[j + 1])])
            if size <= 7:
                assert took_small == 1, "a handful of queries must take the one-launch path"
        assert idx.stats()["k_small"]["launches"] >= 4
        idx.close()


def test_bucket_host_is_search_k_without_a_device_round_trip(engine, orc):
    """kmx_index_bucket_host (kmer_index_element::search_k, kmer_index.hpp:183-190): the bucket of one k-mer out of the host
    arena == the exact occurrence list, None on a miss; dense and open tables, device-built and host-flattened elements."""
    text = synth.ranks(77, 200_000, 4)
    for table, host_flatten, ks in ((engine.TABLE_DENSE, False, [6, 10]), (engine.TABLE_OPEN, False, [10, 14]), (engine.TABLE_OPEN, True, [9]),
                                    (engine.TABLE_AUTO, False, [20])):
        idx = engine.Index(text, 4, ks, table=table, keep_host_arena=True, host_flatten=host_flatten)
        launches_before = sum(v["launches"] for v in idx.stats().values())
        idx.stats_enable(True)
        for k in ks:
            n_miss = 0
            for t in range(60):
                if t % 2:
                    s0 = (t * 7919) % (text.size - k)
                    kmer = text[s0:s0 + k].copy()
                else:
                    kmer = synth.ranks(5000 + t, k, 4)
                got = idx.bucket_host(k, kmer)
                want = orc.naive_scan(text, kmer)
                if want.size == 0:
                    assert got is None
                    n_miss += 1
                else:
                    assert got is not None and np.array_equal(got, want), (table, k, t)
            assert k < 10 or n_miss > 0
        assert sum(v["launches"] for v in idx.stats().values()) == launches_before      # no kernel ran for any of it
        with pytest.raises(engine.KmxError):
            idx.bucket_host(7, text[:7])                                  # no element for this k
        with pytest.raises(engine.KmxError):
            idx.bucket_host(ks[0], np.full(ks[0], 9, np.uint8))          # a letter outside the alphabet
        idx.close()
    plain = engine.Index(text, 4, [8])
    with pytest.raises(engine.KmxError):
        plain.bucket_host(8, text[:8])                                    # built without keep_host_arena
    plain.close()


def test_levels_are_reported_and_old_abi_callers_get_none(engine):
    """kmx_index_levels says what kmx_options.prefix_levels got; a caller compiled against KMX_VERSION 1 / 2 (shorter
    struct_size, cannot name the field) pays for no level."""
    import ctypes as C
    text = synth.ranks(78, 300_000, 4)
    idx = engine.Index(text, 4, [8, 10])
    assert idx.levels() == [2, 1] or idx.levels()[0] >= 1, idx.levels()   # (how far a level goes also depends on the planner: k - L must still be this element's)
    assert idx.memory()["prefix_levels"] > 0
    idx.close()
    none = engine.Index(text, 4, [8, 10], prefix_levels=-1)
    assert none.levels() == [0, 0] and none.memory()["prefix_levels"] == 0
    none.close()
    L = engine.lib()
    ks = np.array([8, 10], np.uint32)
    for old_size in (engine.Options.n_devices.offset, engine.Options.prefix_levels.offset):
        o = engine.Options()
        o.struct_size = old_size
        o.device = -1
        h = C.c_void_p()
        assert L.kmx_index_build(text.ctypes.data, text.size, 4, ks.ctypes.data, 2, C.byref(o), C.byref(h)) == 0
        lv = np.zeros(32, np.uint32)
        assert L.kmx_index_levels(h, lv.ctypes.data) == 0 and not lv.any()
        L.kmx_index_free(h)
