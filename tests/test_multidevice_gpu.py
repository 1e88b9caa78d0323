"""The multi-device index of the C-ABI (kmx_options.devices): one build, replicated images, a host-buffer
batch search sharded contiguously over the replicas in-process, ONE result whose views concatenate the shards in replica
order — byte for byte what a single device returns.  Runs with two replicas on the box's one GPU (the replica list may
name a device twice) and, when the box has them, with one replica per visible device."""
import numpy as np
import pytest

from kmer_index_amd import synth
from tests.helpers import make_queries, pack

pytestmark = pytest.mark.gpu


def _device_sets():
    import torch
    n = torch.cuda.device_count()
    sets = [[0, 0], [0, 0, 0]]
    if n >= 2:
        sets.append(list(range(min(n, 8))))
    return sets


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("table", ["auto", "open"])
def test_n_replicas_return_the_single_device_result_byte_for_byte(engine, table):
    text = synth.ranks(1003, 400_000, 4)
    ks = [8, 10, 12]
    q, off = make_queries(text, 4, [3, 6, 8, 9, 10, 12, 13, 20, 22, 24, 31, 36], 400, seed=91)
    tk = engine.TABLE_OPEN if table == "open" else engine.TABLE_AUTO
    one = engine.Index(text, 4, ks, table=tk, device=0, keep_host_arena=True)
    assert one.devices() == [0]
    r1 = one.search(q, off, flags=engine.SEARCH_KEEP_MASKS)
    want = r1.host()
    base1, words1, cnt1, src1 = r1.masks()
    c1 = r1.counts()
    assert r1.n_parts() == 1
    for devs in _device_sets():
        many = engine.Index(text, 4, ks, table=tk, devices=devs, keep_host_arena=True)
        assert many.devices() == devs and many.info()["device_bytes"] == one.info()["device_bytes"]
        rn = many.search(q, off, flags=engine.SEARCH_KEEP_MASKS)
        assert rn.n_parts() == len(devs) and rn.counts() == c1
        got = rn.host()
        assert _same(got, want), devs
        # the zero-copy view: candidates (arena index is replica-independent) + mask words per STITCH query
        basen, wordsn, cntn, srcn = rn.masks()
        import ctypes as C
        st = np.nonzero(want[3] == engine.KIND_STITCH)[0]
        assert st.size > 100
        assert np.array_equal(cntn[st], cnt1[st]) and np.array_equal(srcn[st], src1[st])
        w1 = np.ctypeslib.as_array(C.cast(words1, C.POINTER(C.c_uint64)), shape=(int(base1[st].max()) + int(cnt1[st].max()) // 64 + 2,))
        wn = np.ctypeslib.as_array(C.cast(wordsn, C.POINTER(C.c_uint64)), shape=(int(basen[st].max()) + int(cntn[st].max()) // 64 + 2,))
        for i in st[:300]:
            nw = int(cnt1[i]) // 64 + 1
            assert np.array_equal(w1[int(base1[i]):int(base1[i]) + nw], wn[int(basen[i]):int(basen[i]) + nw])
        # per-part device views: part p holds queries [q_begin, q_end) with part-local offsets
        import torch
        covered = 0
        for p in range(rn.n_parts()):
            dev, qb, qe, d_off, d_pos, d_st = rn.part_device_ptrs(p)
            assert dev == devs[p] and qb == covered
            covered = qe
        assert covered == off.size - 1
        with pytest.raises(engine.KmxError):
            rn.device_ptrs()
        # the handle is reusable for the next batch, also a smaller one; count-only works over parts
        half = (off.size - 1) // 3
        rn2 = many.search(q[:int(off[half])], off[:half + 1], result=rn)
        h2, p2, s2, k2 = rn2.host()
        assert np.array_equal(h2, want[0][:half + 1]) and np.array_equal(p2, want[1][:int(want[0][half])])
        rc = many.search(q, off, flags=engine.SEARCH_COUNT_ONLY)
        assert np.array_equal(rc.host()[0], want[0])
        rc.close()
        rn.close()
        many.close()
    r1.close()
    one.close()


def test_replicas_on_distinct_physical_devices(engine):
    """The same contract with every replica on a GPU of its own (hipMemcpyPeer between devices, one host thread and one stream
    per device, per-part device views that live on different devices) — what the [0, 0] replica lists of the other tests
    cannot show.  SKIPPED, not faked, on a box with one GPU: until a multi-GPU box has run it, cross-device correctness of
    the in-process replicas is unpinned (README)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs two visible GPUs")
    devs = list(range(min(n, 4)))
    text = synth.ranks(1003, 400_000, 4)
    ks = [8, 10, 12]
    q, off = make_queries(text, 4, [3, 6, 8, 9, 10, 12, 13, 20, 22, 24, 31, 36], 300, seed=5)
    one = engine.Index(text, 4, ks, device=0)
    want = one.search(q, off).host()
    many = engine.Index(text, 4, ks, devices=devs)
    assert many.devices() == devs
    rn = many.search(q, off)
    assert rn.n_parts() == len(devs) and _same(rn.host(), want)
    for p in range(rn.n_parts()):
        dev, qb, qe, d_off, d_pos, d_st = rn.part_device_ptrs(p)
        assert dev == devs[p]
        with torch.cuda.device(dev):                       # the part's arrays really live on that device
            class _Arr:
                def __init__(self, ptr, cnt, typestr):
                    self.__cuda_array_interface__ = {"shape": (int(cnt),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
            t_off = torch.as_tensor(_Arr(d_off, qe - qb + 1, "<i8"), device=f"cuda:{dev}").cpu().numpy().astype(np.uint64)
        assert np.array_equal(t_off - t_off[0], want[0][qb:qe + 1] - want[0][qb])
    # the device form is served by the replica that lives where the queries are
    d = devs[-1]
    tq = torch.from_numpy(q).to(f"cuda:{d}")
    to = torch.from_numpy(off.view(np.int64)).to(f"cuda:{d}")
    with torch.cuda.device(d):
        r = many.search_device(tq.data_ptr(), to.data_ptr(), off.size - 1, stream=torch.cuda.current_stream().cuda_stream)
        assert _same(r.host(), want)
    # the planner range moves on every replica or on none
    many.extend_query_size_range(20_000)
    assert _same(many.search(q, off, result=rn).host(), want)
    rn.close(); r.close(); many.close(); one.close()


def test_fewer_queries_than_replicas_and_empty_batches(engine):
    text = synth.ranks(5, 50_000, 4)
    many = engine.Index(text, 4, [6], devices=[0, 0, 0])
    one = engine.Index(text, 4, [6], device=0)
    for nq in (0, 1, 2, 4):
        q, off = synth.uniform_queries(17, nq, 6, 4)
        a, b = many.search(q, off), one.search(q, off)
        assert _same(a.host(), b.host()) and a.counts() == b.counts()
        a.close(); b.close()
    many.close(); one.close()


def test_device_form_on_a_replicated_index_uses_the_replica_where_the_queries_live(engine):
    import torch
    text = synth.ranks(1002, 300_000, 4)
    idx = engine.Index(text, 4, [10], devices=[0, 0])
    q, off = synth.uniform_queries(2002, 30_000, 10, 4)
    d_q = torch.from_numpy(q).to("cuda:0")
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    r = idx.search_device(d_q.data_ptr(), d_off.data_ptr(), off.size - 1)
    ref = engine.Index(text, 4, [10], device=0).search(q, off)
    assert _same(r.host(), ref.host())
    r.close(); ref.close(); idx.close()


def test_replicated_image_loads_from_disk_and_env_var_widens_a_plain_build(engine, tmp_path, monkeypatch):
    text = synth.ranks(8, 120_000, 5)
    q, off = make_queries(text, 5, [4, 7, 9, 14, 18], 200, seed=3)
    one = engine.Index(text, 5, [7, 9], device=0)
    want = one.search(q, off).host()
    p = str(tmp_path / "img.kmx")
    one.save(p)
    loaded = engine.Index.load(p, devices=[0, 0])
    assert loaded.devices() == [0, 0] and _same(loaded.search(q, off).host(), want)
    loaded.close()
    monkeypatch.setenv("KMX_DEVICES", "0,0,0")           # what a make_kmer_index caller sets to use every GPU: "all"
    env = engine.Index(text, 5, [7, 9])
    assert env.devices() == [0, 0, 0] and _same(env.search(q, off).host(), want)
    env.close()
    monkeypatch.setenv("KMX_DEVICES", "all")
    env = engine.Index(text, 5, [7, 9])
    import torch
    assert env.devices() == list(range(torch.cuda.device_count())) and _same(env.search(q, off).host(), want)
    env.close()
    one.close()


def test_index_freed_before_a_pending_async_result(engine):
    """a KMX_SEARCH_ASYNC search left pending on a result whose index is closed first — kmx_index_free
    completes it, so touching the result afterwards neither launches kernels on freed memory nor loses the hits."""
    import torch
    text = synth.ranks(1002, 2_000_000, 4)
    q, off = synth.uniform_queries(2002, 400_000, 10, 4)
    d_q = torch.from_numpy(q).to("cuda:0")
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    idx = engine.Index(text, 4, [10], device=0)
    want = idx.search(q, off).host()
    res = engine.Result()
    idx.search_device(d_q.data_ptr(), d_off.data_ptr(), off.size - 1, flags=engine.SEARCH_ASYNC, result=res)   # first batch: the fill is NOT speculative
    idx.close()                                           # completes the pending half (validate / fill) before releasing the image
    assert res.counts()["n_hits"] == want[1].size
    assert _same(res.host(), want)
    res.close()


def test_batches_too_large_for_one_pass_are_streamed_in_chunks(engine, monkeypatch):
    """kmx_search_batch streams a batch that does not fit the device in one pass chunk by chunk (KMX_HOST_CHUNK forces small
    chunks here): counts, host views and masks of the whole batch equal the one-pass result; the handle is reusable."""
    text = synth.ranks(1003, 400_000, 4)
    ks = [8, 10, 12]
    q, off = make_queries(text, 4, [3, 6, 8, 9, 10, 12, 13, 20, 22, 24, 31, 36], 400, seed=17)
    idx = engine.Index(text, 4, ks, keep_host_arena=True)
    r1 = idx.search(q, off, flags=engine.SEARCH_KEEP_MASKS)
    want, c1 = r1.host(), r1.counts()
    base1, words1, cnt1, src1 = r1.masks()
    import ctypes as C
    st = np.nonzero(want[3] == engine.KIND_STITCH)[0]
    w1 = np.ctypeslib.as_array(C.cast(words1, C.POINTER(C.c_uint64)), shape=(int((base1[st] + cnt1[st] // 64 + 1).max()),)).copy()
    monkeypatch.setenv("KMX_HOST_CHUNK", "1000")
    rc = idx.search(q, off, flags=engine.SEARCH_KEEP_MASKS)
    assert rc.n_parts() == (off.size - 1 + 999) // 1000 and rc.counts() == c1
    assert _same(rc.host(), want)
    basen, wordsn, cntn, srcn = rc.masks()
    wn = np.ctypeslib.as_array(C.cast(wordsn, C.POINTER(C.c_uint64)), shape=(int((basen[st] + cntn[st] // 64 + 1).max()),))
    assert np.array_equal(cntn[st], cnt1[st]) and np.array_equal(srcn[st], src1[st])
    for i in st[::7]:
        nw = int(cnt1[i]) // 64 + 1
        assert np.array_equal(w1[int(base1[i]):int(base1[i]) + nw], wn[int(basen[i]):int(basen[i]) + nw])
    with pytest.raises(engine.KmxError):
        rc.device_ptrs()
    # the chunked handle serves the next batches too: another chunked one, a small one, an empty one
    half = (off.size - 1) // 2
    r2 = idx.search(q[:int(off[half])], off[:half + 1], result=rc)
    h2 = r2.host()
    assert np.array_equal(h2[0], want[0][:half + 1]) and np.array_equal(h2[1], want[1][:int(want[0][half])])
    monkeypatch.setenv("KMX_HOST_CHUNK", "100000000")
    r3 = idx.search(q[:int(off[50])], off[:51], result=rc)
    assert np.array_equal(r3.host()[1], want[1][:int(want[0][50])]) and r3.counts()["nq"] == 50
    r4 = idx.search(np.zeros(0, np.uint8), np.zeros(1, np.uint64), result=rc)
    assert r4.counts()["nq"] == 0 and r4.host()[0].tolist() == [0]
    # a plain handle turned into the worker of a chunked batch
    monkeypatch.setenv("KMX_HOST_CHUNK", "777")
    r5 = idx.search(q, off, flags=engine.SEARCH_KEEP_MASKS, result=r1)
    assert _same(r5.host(), want) and r5.counts() == c1
    rc.close(); r5.close(); idx.close()


def test_chunked_batches_whose_hits_arrive_late_and_count_only(engine, monkeypatch):
    """The chunk-streamed form sizes its host view of the positions from the first chunk's hits per query and grows it when a
    later chunk proves that short: a batch whose first chunks find next to nothing and whose last ones find tens of thousands
    of positions per query; the same batch count-only (no positions cross PCIe)."""
    text = synth.ranks(2025, 300_000, 4)
    idx = engine.Index(text, 4, [8, 11])
    rng = np.random.default_rng(8)
    qs = [rng.integers(0, 4, 30).astype(np.uint8) for _ in range(2500)]                       # random 30-mers: no hits
    qs += [text[s0:s0 + 8].copy() for s0 in range(100, 1100)]                                 # exact 8-mers: a few hits each
    qs += [text[s0:s0 + m].copy() for m in (3, 2, 4) for s0 in range(5000, 5300)]             # sub-k: thousands of hits each
    q, off = pack(qs)
    want = idx.search(q, off).host()
    monkeypatch.setenv("KMX_HOST_CHUNK", "500")
    rc = idx.search(q, off)
    assert rc.n_parts() == (len(qs) + 499) // 500 and _same(rc.host(), want)
    rc2 = idx.search(q, off, result=rc)                                                       # the grown views are reused
    assert _same(rc2.host(), want)
    rco = idx.search(q, off, flags=engine.SEARCH_COUNT_ONLY)
    h = rco.host()
    assert np.array_equal(h[0], want[0]) and h[1].size == 0 and np.array_equal(h[2], want[2])
    rc.close(); rco.close(); idx.close()


def test_no_device_memory_is_leaked(engine, monkeypatch):
    """Indexes (with replicas), results of every shape (general, latency path, multi-part, chunked) and the pools behind them give
    their device memory back: free memory after ten rounds of build / search / free equals what it was after the first."""
    import torch
    text = synth.ranks(55, 500_000, 4)
    q, off = make_queries(text, 4, [6, 8, 10, 13, 20, 25], 300, seed=2)

    def one_round():
        idx = engine.Index(text, 4, [8, 10], devices=[0, 0], keep_host_arena=True)
        r = idx.search(q, off, flags=engine.SEARCH_KEEP_MASKS)
        r.host(); r.masks()
        s = idx.search(q[:int(off[3])], off[:4])
        s.host()
        monkeypatch.setenv("KMX_HOST_CHUNK", "500")
        c = idx.search(q, off)
        c.host()
        monkeypatch.delenv("KMX_HOST_CHUNK")
        one = engine.Index(text, 4, [9], table=engine.TABLE_OPEN)
        d_q = torch.from_numpy(q).to("cuda:0")
        d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
        a = one.search_device(d_q.data_ptr(), d_off.data_ptr(), off.size - 1, flags=engine.SEARCH_ASYNC)
        one.close()                                        # completes the pending search
        a.host()
        for x in (r, s, c, a):
            x.close()
        idx.close()
        del d_q, d_off
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info(0)[0]

    first = one_round()
    for _ in range(9):
        last = one_round()
    assert abs(first - last) <= 8 << 20, (first, last)


def _dev_array(ptr, n, dtype, device):
    """n items at device pointer `ptr` as a numpy array (through a torch tensor aliasing the memory)."""
    import torch

    class _Arr:
        def __init__(self, p, count, typestr):
            self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(p), False), "version": 2}

    if n == 0:
        return np.zeros(0, dtype)
    typestr = {np.uint64: "<i8", np.uint32: "<i4", np.uint8: "|u1"}[dtype]
    t = torch.as_tensor(_Arr(ptr, n, typestr), device=f"cuda:{device}")
    return t.cpu().numpy().view(dtype).copy()


@pytest.mark.parametrize("devs", [[0, 0], [0, 0, 0, 0]])
def test_gather_device_returns_the_single_device_arrays_byte_for_byte(engine, devs):
    """kmx_result_gather_device (SURVEY 8e step 3 behind the C-ABI): the parts of a multi-replica result gathered into one set of
    device arrays == what one device returns for the whole batch.  One ordinal listed several times on a 1-GPU box (the peer
    copies are then device-local; the displacements, the rebase kernel and the stream ordering are the ones N GPUs use)."""
    text = synth.ranks(1003, 300_000, 4)
    ks = [8, 10, 12]
    q, off = make_queries(text, 4, [5, 8, 9, 10, 12, 13, 20, 22, 24, 31], 500, seed=17)
    nq = off.size - 1
    one = engine.Index(text, 4, ks, device=0)
    want = one.search(q, off).host()
    many = engine.Index(text, 4, ks, devices=devs)
    for flags in (engine.SEARCH_DEFAULT, engine.SEARCH_COUNT_ONLY):
        rn = many.search(q, off, flags=flags)
        assert rn.n_parts() == len(devs)
        for rep in range(2):                                           # (the second gather reuses the result's buffers)
            d_off, d_pos, d_st = rn.gather_device(0)
            assert np.array_equal(_dev_array(d_off, nq + 1, np.uint64, 0), want[0])
            assert np.array_equal(_dev_array(d_st, nq, np.uint8, 0), want[2])
            if flags == engine.SEARCH_COUNT_ONLY:
                assert not d_pos
            else:
                assert np.array_equal(_dev_array(d_pos, int(want[0][nq]), np.uint32, 0), want[1])
        rn.close()
    # a handful of queries (everything on the first replica), and a single-device result gathered where it lies
    few = many.search(q[:int(off[40])], off[:41])
    d_off, d_pos, d_st = few.gather_device(0)
    assert np.array_equal(_dev_array(d_off, 41, np.uint64, 0), want[0][:41])
    assert np.array_equal(_dev_array(d_pos, int(want[0][40]), np.uint32, 0), want[1][:int(want[0][40])])
    few.close()
    r1 = one.search(q, off)
    a, b, s = r1.gather_device(0)
    assert (a, b, s) == r1.device_ptrs()
    with pytest.raises(engine.KmxError):
        r1.gather_device(99)
    r1.close()
    many.close()
    one.close()
