"""The single-pass exact search (k_fused): lookup, offsets (decoupled look-back over tile descriptors) and the copy of the
hit lists in one launch.  Taken for a batch when the previous batch on the same result handle held nothing but plain exact
lookups; it checks its own assumptions and the host falls back to the general pipeline when one fails.  Same results bit for
bit either way."""
import os

import numpy as np
import pytest

from kmer_index_amd import synth
from tests.helpers import pack

pytestmark = pytest.mark.gpu


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.fixture
def items_env():
    old = os.environ.get("KMX_FUSED_ITEMS")
    yield
    if old is None:
        os.environ.pop("KMX_FUSED_ITEMS", None)
    else:
        os.environ["KMX_FUSED_ITEMS"] = old


@pytest.mark.parametrize("sigma,k,n,table,aligned", [(4, 10, 3_000_000, "auto", True), (4, 6, 400_000, "auto", True), (4, 6, 400_000, "open", True),
                                                     (5, 8, 2_000_000, "auto", True), (5, 8, 2_000_000, "open", False), (20, 4, 600_000, "auto", True),
                                                     (4, 12, 500_000, "auto", True)])
def test_fused_pass_equals_the_general_pipeline(engine, orc, items_env, sigma, k, n, table, aligned):
    text = synth.ranks(11 + sigma, n, sigma)
    idx = engine.Index(text, sigma, [k], table=engine.TABLE_OPEN if table == "open" else engine.TABLE_AUTO, aligned_copy=aligned)
    idx.stats_enable(True)
    nq = 70_001                                           # not a multiple of any tile size
    q, off = synth.mixed_queries(5, text, nq, [k], sigma, planted_frac=0.5)
    res = engine.Result()
    want = idx.search(q, off, result=res).host()          # first batch on the handle: the general pipeline
    assert idx.stats()["k_fused"]["launches"] == 0
    o_off, o_pos, o_st, _ = orc.Index(text, sigma, [k]).search_batch(q[:3000 * k], off[:3001], n_threads=4)
    assert np.array_equal(want[0][:3001], o_off) and np.array_equal(want[1][:int(o_off[-1])], o_pos)
    launches = 0
    for items in ("1", "2", "4", None):
        if items is None:
            os.environ.pop("KMX_FUSED_ITEMS", None)
        else:
            os.environ["KMX_FUSED_ITEMS"] = items
        got = idx.search(q, off, result=res)
        launches += 1
        assert idx.stats()["k_fused"]["launches"] == launches, "the steady-state batch did not take the single pass"
        assert got.counts()["n_hits"] == want[1].size
        assert _same(got.host(), want), (items,)
    # the same through the async form with two handles in rotation (what bench.py does)
    import torch
    d_q = torch.from_numpy(q).to("cuda:0")
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    rs = [engine.Result(), engine.Result()]
    for r in rs:
        idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, result=r)
    before = idx.stats()["k_fused"]["launches"]
    for i in range(6):
        idx.search_device(d_q.data_ptr(), d_off.data_ptr(), nq, flags=engine.SEARCH_ASYNC, result=rs[i % 2])
    for r in rs:
        assert _same(r.host(), want)
    assert idx.stats()["k_fused"]["launches"] == before + 6
    for r in rs + [res]:
        r.close()
    idx.close()


def test_fused_pass_aborts_and_the_general_pipeline_answers(engine, orc):
    sigma, k = 4, 8
    text = synth.ranks(21, 1_000_000, sigma)
    idx = engine.Index(text, sigma, [k])
    oidx = orc.Index(text, sigma, [k])
    idx.stats_enable(True)
    nq = 20_000
    exact_q, exact_off = synth.mixed_queries(6, text, nq, [k], sigma, planted_frac=0.3)
    qs = [exact_q[int(exact_off[i]):int(exact_off[i + 1])] for i in range(nq)]
    res = engine.Result()

    def check(queries, expect_fused_launch, expect_general, bad_rank=()):
        q, off = pack(queries)
        b = idx.stats()
        r = idx.search(q, off, result=res)
        a = idx.stats()
        assert a["k_fused"]["launches"] - b["k_fused"]["launches"] == int(expect_fused_launch)
        assert (a["k_lookup"]["launches"] - b["k_lookup"]["launches"] == 1) == expect_general
        o_off, o_pos, o_st, _ = oidx.search_batch(q, off, mode=orc.MODE_INTENDED, n_threads=8)
        h = r.host()
        want_st = o_st.astype(np.uint8)
        want_st[list(bad_rank)] = engine.Q_BAD_RANK             # (not representable in the reference's alphabet_t: the oracle has no such status)
        assert np.array_equal(h[0], o_off) and np.array_equal(h[1], o_pos) and np.array_equal(h[2], want_st)

    check(qs, False, True)                                # 1st batch: general; it was all exact
    check(qs, True, False)                                # 2nd: the single pass
    # errors the single pass reports itself: empty, too long, a letter outside the alphabet
    bad = list(qs)
    bad[7] = np.zeros(0, np.uint8)
    bad[4099] = np.zeros(10_001, np.uint8)
    bad[12_345] = np.full(k, 9, np.uint8)
    check(bad, True, False, bad_rank=(12_345,))
    # a cross-referenced query and a sub-k query void the pass: general pipeline, and the next batch does not try
    mixed = list(qs)
    mixed[100] = text[5000:5000 + 2 * k + 3].copy()
    mixed[9000] = text[777:777 + k - 2].copy()
    check(mixed, True, True)
    check(qs, False, True)                                # the previous batch was not all exact
    check(qs, True, False)
    # hit lists that outgrow the buffer kept from the previous batch: every query planted in a repeat
    rep = text.copy()
    rep[:400_000] = np.tile(text[1000:1000 + 40], 10_000)
    idx2 = engine.Index(rep, sigma, [k])
    idx2.stats_enable(True)
    few = pack(qs)
    many = pack([rep[40 + (i % 40):40 + (i % 40) + k].copy() for i in range(nq)])      # ~10 000 hits each
    r2 = engine.Result()
    idx2.search(*few, result=r2)
    got = idx2.search(*many, result=r2)
    assert idx2.stats()["k_fused"]["launches"] == 1 and idx2.stats()["k_lookup"]["launches"] == 2
    h = got.host()
    cnt = np.diff(h[0])
    assert (cnt >= 9_990).all() and h[1].size == int(cnt.sum())
    sel = [0, 1, 39, 40, nq - 1]
    for i in sel:
        assert np.array_equal(h[1][int(h[0][i]):int(h[0][i + 1])], orc.naive_scan(rep, many[0][i * k:(i + 1) * k]))
    r2.close(); res.close()
    idx.close(); idx2.close()
