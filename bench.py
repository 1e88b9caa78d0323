#!/usr/bin/env python3
"""bench.py — batch exact-match search throughput (BASELINE.json metric) on N GPUs of one node.

One "step" = one pass of the hot path over one batch: 1e7 uniform random DNA4 10-mers per GPU against a
k=10 index of a 1e8-bp synthetic text (BASELINE.json configs[1]); inputs resident in HBM before the timed
region; the index is replicated per GPU and every rank searches its own query shard (weak scaling, no
data-path collective in the `value` leg; per-rank totals are exchanged once after the timed region).

Launch forms (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI):
  python bench.py --gpus N ...                              this process only SPAWNS the N ranks (it never touches a
                                                            GPU), waits for them and exits with their status;
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
                                                            the launcher already made the ranks (RANK / LOCAL_RANK /
                                                            WORLD_SIZE / MASTER_* in the environment).
Fewer than N visible devices is an error (exit code 3), never a silent 1-GPU run.

At N > 1 the same JSON line also carries `gather_hits`: the same K steps with the RCCL gatherv of every hit list to
rank 0 inside each step (grouped point-to-point exchange, gather of step i under the search of step i+1).

Prints ONE JSON line on rank 0 (see the bench contract in the task / DESIGN.md §6).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
XGMI_LINK_GBPS = 153.0   # per direct link and direction

# (sigma, n, ks, queries per GPU, query lengths, planted share, text seed, query seed)
CFG = {2: (4, 100_000_000, [10], 10_000_000, [10], 0.0, 1002, 2002),
       3: (4, 100_000_000, [8, 10, 12], 10_000_000, [8, 10, 12, 20, 22, 24], 0.5, 1003, 2003),
       4: (5, 100_000_000, [10], 12_500_000, [10], 0.0, 1004, 2004),
       5: (20, 10_000_000, [5], 10_000_000, [5], 0.5, 1005, 2005)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[N-1]; 2 = the metric's workload (default), 3/4/5 are informational")
    ap.add_argument("--n", "--text-len", dest="n", type=int, default=0,
                    help="text length (0 = the config's); spell it --text-len under torch.distributed.run, whose own parser takes --n for an ambiguous prefix")
    ap.add_argument("--nq", type=int, default=0, help="queries per GPU per step (0 = the config's)")
    ap.add_argument("--table", choices=["open", "dense", "auto"], default="auto",
                    help="auto = the engine's default policy (direct addressing when sigma^k <= 4(n-k+1), else open addressing)")
    ap.add_argument("--no-open-compare", action="store_true",
                    help="skip the extra leg that times the same workload on the open-addressing table (N=1, config 2 only)")
    ap.add_argument("--gather", choices=["both", "totals", "hits"], default="both",
                    help="both (default): `value` from the sharded leg (hit lists stay where they were produced, per-shard totals "
                         "exchanged after the timed region) and, at N > 1, a second leg with the RCCL gatherv of every hit list to rank 0 "
                         "inside each step, reported as gather_hits; totals: first leg only; hits: `value` IS the gather leg")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="result handles the steps rotate over on ONE stream with KMX_SEARCH_ASYNC: the host enqueues step i+1 while "
                         "step i runs and reads step i's counters when its handle comes round again (1 = every step waits for its own)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the result handles of the sharded leg are spread over (handle i on stream i %% streams): with 2, the "
                         "lookup / scan of step i+1 may run under the tail of step i's fill")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the extra leg that times the same steps with the handles on two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true",
                    help="N=1, config 2 only: skip the PCIe-inclusive leg (kmx_search_batch + kmx_result_view from / to host memory)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N=1, config 2 only: skip the legs that run BASELINE configs[2..4] (--config 3/4/5) for the same K steps and "
                         "report them as other_configs in the same JSON line")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="queries in the CPU baseline sample")
    ap.add_argument("--cpu-passes", type=int, default=5, help="timed passes of every CPU baseline leg (>= 5; one warm-up pass in front)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline threads (0 = min(16, usable cores): the box's CPU share)")
    ap.add_argument("--verify", type=int, default=20000, help="queries checked against the oracle after timing")
    ap.add_argument("--sort-queries", action="store_true",
                    help="experiment: the shard's queries sorted by rank-hash on the host before upload — the best case of a hash-binned "
                         "lookup + fill for the READ side (table and arena swept in order); says nothing about the cost of binning or of "
                         "writing the hit lists back in query order")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal only: allow more ranks than visible GPUs (ranks share devices; the exchange runs over gloo because "
                         "RCCL refuses two ranks on one device).  The JSON line says so.")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher(args, argv):
    """`python bench.py --gpus N` without a distributed launcher: spawn the N ranks.  This process never initialises a
    GPU (torch.cuda.device_count() does not on this image), so the children start from a clean parent."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.oversubscribe:
        print(f"[bench] --gpus {args.gpus} but only {n_dev} GPU(s) visible: refusing to report a smaller run as N={args.gpus}",
              file=sys.stderr, flush=True)
        return 3
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(args.gpus)
    procs = []
    for r in range(args.gpus):
        e = dict(env)
        e["RANK"] = e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for o in pending:               # a failed rank leaves the others stuck in a collective
                        o.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launcher(args, argv))
    sys.exit(worker(args))


def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from kmer_index_amd import dist as kdist
    from kmer_index_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU", file=sys.stderr, flush=True)
        return 3
    n_dev = torch.cuda.device_count()
    if n_dev == 0 or not torch.cuda.is_available():
        print("[bench] bench.py needs a GPU: the engine has no CPU path", file=sys.stderr, flush=True)
        return 3
    if n_dev < local_world and not args.oversubscribe:
        print(f"[bench] {local_world} local ranks but only {n_dev} GPU(s) visible: refusing (use --oversubscribe to rehearse)",
              file=sys.stderr, flush=True)
        return 3
    dev_index = local_rank % n_dev                   # == local_rank unless oversubscribed
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = None
    if world > 1 or os.environ.get("KMX_BENCH_INIT_PG"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = "gloo" if (args.oversubscribe and n_dev < local_world) else os.environ.get("KMX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    comm_dev = dev if backend == "nccl" else None    # where collectives' tensors live

    sigma, n_cfg, ks, nq_cfg, qlens, planted, tseed, qseed = CFG[args.config]
    args.sigma = sigma
    args.n = args.n or n_cfg
    args.nq = args.nq or nq_cfg
    t0 = time.time()
    text = synth.ranks(tseed, args.n, args.sigma)                     # identical on every rank
    log(f"text n={args.n} sigma={args.sigma} generated in {time.time() - t0:.1f}s")
    t0 = time.time()
    table = {"open": engine.TABLE_OPEN, "dense": engine.TABLE_DENSE, "auto": engine.TABLE_AUTO}[args.table]
    idx = engine.Index(text, args.sigma, ks, table=table, device=dev_index)
    info = idx.info()
    log(f"index built+uploaded in {time.time() - t0:.1f}s: {info}")

    # this rank's query shard: letters [rank*nq*m, (rank+1)*nq*m) of the config's query stream
    def make_queries(text_, sigma_, nq_, qlens_, planted_, qseed_):
        if planted_ == 0.0:
            m_ = qlens_[0]
            qr = np.empty(nq_ * m_, np.uint8)
            chunk = 1 << 24
            for s in range(0, nq_ * m_, chunk):
                e = min(nq_ * m_, s + chunk)
                z = synth.u64_stream(qseed_, e - s, rank * nq_ * m_ + s)
                qr[s:e] = (((z >> np.uint64(32)) * np.uint64(sigma_)) >> np.uint64(32)).astype(np.uint8)
            qo = np.arange(nq_ + 1, dtype=np.uint64) * np.uint64(m_)
            if args.sort_queries:
                h = np.zeros(nq_, np.uint64)
                q2 = qr.reshape(nq_, m_)
                for j in range(m_):
                    h = h * np.uint64(sigma_) + q2[:, j]
                qr = np.ascontiguousarray(q2[np.argsort(h, kind="stable")]).reshape(-1)
            return m_, qr, qo
        qr, qo = synth.mixed_queries(qseed_ + 7919 * rank, text_, nq_, qlens_, sigma_, planted_frac=planted_)
        return 0, qr, qo

    nq = args.nq
    m, qr_host, qoff_host = make_queries(text, args.sigma, nq, qlens, planted, qseed)
    n_letters = int(qoff_host[-1])
    d_qr = torch.from_numpy(qr_host).to(dev)
    d_qoff = torch.from_numpy(qoff_host.view(np.int64)).to(dev)
    main_queries = (d_qr, d_qoff, nq)
    torch.cuda.synchronize()

    depth = max(1, args.pipeline)
    main_stream = torch.cuda.current_stream()

    def barrier():
        if backend is not None and world > 1:
            dist.barrier()

    # ------------------------------------------------------------------------------------------------------------
    # leg 1 — sharded: hit lists stay in the HBM of the GPU that produced them.  `depth` result handles rotate on one
    # stream with KMX_SEARCH_ASYNC (the host enqueues step i+1 while step i runs).
    # ------------------------------------------------------------------------------------------------------------
    sh_streams = [main_stream] + [torch.cuda.Stream(device=dev) for _ in range(max(depth, 2) - 1)]

    def run_sharded(index, steps, warmup, collect_stats, n_streams=1, queries=None, extra_flags=0):
        d_qr, d_qoff, nq = queries or main_queries
        n_streams = max(1, min(n_streams, depth))
        results = [engine.Result() for _ in range(depth)]
        flags = (engine.SEARCH_ASYNC if depth > 1 else engine.SEARCH_DEFAULT) | extra_flags
        no = [0]

        def step():
            h = no[0] % depth
            no[0] += 1
            index.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, flags=flags, stream=sh_streams[h % n_streams].cuda_stream, result=results[h])

        # setup, like the index build: every result handle allocates its device buffers (grow-only, sized by its first
        # batch) before the W warmup steps, so that neither warmup nor the timed region holds an allocation
        for h, r_ in enumerate(results):
            for _ in range(2):
                index.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, flags=extra_flags, stream=sh_streams[h % n_streams].cuda_stream, result=r_)
        torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        for r_ in results:
            r_.counts()                                   # no warmup step is left pending into the timed region
        torch.cuda.synchronize()
        if collect_stats:
            index.stats_enable(True)
            index.stats_reset()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        for r_ in results:
            r_.counts()                                   # completes a step that is still pending on its handle (inside the timed region)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        stats = None
        if collect_stats:
            stats = index.stats()
            index.stats_enable(False)
        return elapsed, results, stats

    # ------------------------------------------------------------------------------------------------------------
    # leg 2 — gather: every step ends with all hit lists on rank 0.  Two handles on two compute streams; the gather of
    # step i runs on a third stream (and on RCCL's own) while step i+1 searches; a handle is searched into again only
    # after its gather has drained it.
    # ------------------------------------------------------------------------------------------------------------
    def run_gather(index, steps, warmup):
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        comm = torch.cuda.Stream(device=dev)
        results = [engine.Result() for _ in range(2)]
        drained = [None, None]                            # event: the gather that read this handle is complete
        gat = kdist.HitGather(dst=0)
        g_events = []                                     # (start, end) pairs around every timed gather on `comm`
        timing = [False]
        for h in range(2):
            for _ in range(2):
                index.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, stream=streams[h].cuda_stream, result=results[h])
        torch.cuda.synchronize()

        def enqueue(i):
            h = i % 2
            if drained[h] is not None:
                streams[h].wait_event(drained[h])
            index.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, flags=engine.SEARCH_ASYNC, stream=streams[h].cuda_stream, result=results[h])

        def complete(i):
            h = i % 2
            c = results[h].counts()                       # host waits for step i's counters; launches whatever the first half left
            t_off, t_pos = results[h].device_tensors(dev)
            ready = torch.cuda.Event()
            ready.record(streams[h])
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timing[0] else None
                if ev:
                    ev[0].record(comm)
                if backend == "nccl":
                    gat.gather(t_off, t_pos)
                else:                                     # gloo rehearsal: the exchange goes through host memory
                    comm.synchronize()
                    gat.gather(t_off.cpu(), t_pos.cpu())
                if ev:
                    ev[1].record(comm)
                    g_events.append(ev)
                drained[h] = torch.cuda.Event()
                drained[h].record(comm)
            return c

        def run(k):
            enqueue(0)
            for i in range(1, k):
                enqueue(i)
                complete(i - 1)
            c = complete(k - 1)
            comm.synchronize()
            return c

        run(max(warmup, 2))
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        timing[0] = True
        t0 = time.perf_counter()
        c = run(steps)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        g_ms = [a.elapsed_time(b) for a, b in g_events]
        out = {"elapsed": elapsed, "gather_ms": sum(g_ms) / max(len(g_ms), 1), "bytes_per_peer": list(gat.last_bytes_per_peer),
               "n_hits": c["n_hits"]}
        if rank == 0 and gat.g_off is not None:
            nq_total = nq * world
            out["gathered_queries"] = nq_total
            out["gathered_hits"] = int(gat.g_off[nq_total].item())
        for r_ in results:
            r_.close()
        return out

    def reread_ok(text_, qr_, qo_, hit_off_, positions_, nv_):
        # every reported position of the first nv_ queries re-reads to its query; lists ascending without repeats
        cntv = np.diff(hit_off_[:nv_ + 1]).astype(np.int64)
        qi = np.repeat(np.arange(nv_), cntv)
        pos = positions_[:int(hit_off_[nv_])].astype(np.int64)
        lens = np.diff(qo_[:nv_ + 1]).astype(np.int64)
        ok = True
        for j in range(int(lens.max()) if nv_ else 0):
            sel = lens[qi] > j
            ok &= bool(np.array_equal(text_[pos[sel] + j], qr_[qo_[qi[sel]].astype(np.int64) + j]))
        if pos.size > 1:
            same_q = qi[1:] == qi[:-1]
            ok &= bool((pos[1:][same_q] > pos[:-1][same_q]).all())
        return bool(ok)

    def windows_complete_ok(text_, sigma_, qr_, qo_, hit_off_, positions_, nv_, W=4_000_000):
        # COMPLETENESS of the first nv_ queries' lists inside two windows of the text (its first and its last W letters: the tail
        # is where the last-kmer fix-up of sub-k queries lives): every occurrence a plain scan of the window finds must be
        # reported, and nothing else inside it.  reread_ok proves the reported positions right; this proves none is missing
        # there (a dropped run or position of the merge / level / largest-k paths would show).  numpy only: a rolling
        # polynomial hash per query length, candidates confirmed letter by letter.
        n_ = text_.size
        lens = np.diff(qo_[:nv_ + 1]).astype(np.int64)
        ok = True
        for lo in sorted({0, max(0, n_ - W)}):
            win = text_[lo:lo + W].astype(np.uint64)
            for m_ in np.unique(lens):
                m_ = int(m_)
                if m_ == 0 or m_ > win.size:
                    continue
                qsel = np.nonzero(lens == m_)[0]
                with np.errstate(over="ignore"):
                    h = np.zeros(win.size - m_ + 1, np.uint64)
                    for j in range(m_):
                        h = h * np.uint64(1099511628211) + win[j:j + h.size]
                    qh = np.zeros(qsel.size, np.uint64)
                    for j in range(m_):
                        qh = qh * np.uint64(1099511628211) + qr_[qo_[qsel].astype(np.int64) + j].astype(np.uint64)
                order = np.argsort(qh, kind="stable")
                qh_s = qh[order]
                cand = np.nonzero(np.isin(h, qh_s))[0]                 # window offsets whose hash equals some query's
                first = np.searchsorted(qh_s, h[cand], side="left")
                last = np.searchsorted(qh_s, h[cand], side="right")
                want = {int(q): [] for q in qsel}
                for c, a, b in zip(cand.tolist(), first.tolist(), last.tolist()):
                    for t in range(a, b):                               # (several of the first queries may be the same m-mer)
                        q = int(qsel[order[t]])
                        if np.array_equal(text_[lo + c:lo + c + m_], qr_[int(qo_[q]):int(qo_[q]) + m_]):
                            want[q].append(lo + c)
                for q in qsel.tolist():
                    got = positions_[int(hit_off_[q]):int(hit_off_[q + 1])].astype(np.int64)
                    got = got[(got >= lo) & (got + m_ <= lo + win.size)]
                    ok &= bool(np.array_equal(got, np.asarray(want[q], np.int64)))
        return bool(ok)

    # BASELINE configs[2..4] for the same K steps each, reported beside the headline (N = 1 only): own text, own index, the
    # same pipeline as `value`, checked by re-reading the first 2000 queries' positions and by the hit total of a second pass
    # A query length the configs do not hold, on an index that is already built: nq_p queries of m letters (half planted),
    # the same pipeline, checked by re-reading the first 2000 queries' positions.  Sub-k lengths and long reads: informational.
    def length_probe(index, text_, sigma_, m_, nq_p=500_000, flags_env=None, nv_p=2000):
        qr_p, qo_p = synth.mixed_queries(100 + m_, text_, nq_p, [m_], sigma_)
        q_p = (torch.from_numpy(qr_p).to(dev), torch.from_numpy(qo_p.view(np.int64)).to(dev), nq_p)
        el_p, res_p, st_p = run_sharded(index, max(3, args.steps // 4), 1, True, 1, q_p)
        cn = res_p[0].counts()
        ho, po, _, _ = res_p[0].host(copy=False)
        ok = reread_ok(text_, qr_p, qo_p, ho, po, min(nv_p, nq_p)) and windows_complete_ok(text_, sigma_, qr_p, qo_p, ho, po, min(nv_p // 4, nq_p))
        steps_p = max(3, args.steps // 4)
        o = {"M_queries_per_s": round(nq_p * steps_p / el_p / 1e6, 1), "G_hits_per_s": round(cn["n_hits"] * steps_p / el_p / 1e9, 1),
             "verified": bool(ok), "queries": nq_p,
             "kernels_avg_ms": {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in st_p.items() if v["launches"]}}
        for r_ in res_p:
            r_.close()
        return o

    def other_config(cfg):
        sg, n_c, ks_c, nq_c, ql_c, pl_c, ts_c, qs_c = CFG[cfg]
        t_c = time.time()
        text_c = synth.ranks(ts_c, n_c, sg)
        idx_c = engine.Index(text_c, sg, ks_c, device=dev_index)
        info_c = idx_c.info()
        _, qr_c, qo_c = make_queries(text_c, sg, nq_c, ql_c, pl_c, qs_c)
        q_c = (torch.from_numpy(qr_c).to(dev), torch.from_numpy(qo_c.view(np.int64)).to(dev), nq_c)
        el_c, res_c, st_c = run_sharded(idx_c, args.steps, args.warmup, True, 1, q_c)
        cn = res_c[0].counts()
        ho, po, _, _ = res_c[0].host(copy=False)
        ok = reread_ok(text_c, qr_c, qo_c, ho, po, min(2000, nq_c)) and all(r_.counts()["n_hits"] == cn["n_hits"] for r_ in res_c)
        ok = ok and windows_complete_ok(text_c, sg, qr_c, qo_c, ho, po, min(1000, nq_c))
        ms = el_c / args.steps * 1e3
        fill_c = st_c.get("k_fill", {"launches": 0, "total_ms": 0.0})
        f_ms = fill_c["total_ms"] / max(fill_c["launches"], 1)
        job_b = float(int(qo_c[-1])) + 24.0 * nq_c + 8.0 * cn["n_hits"]                     # the same R + W definition as the headline's
        o = {"value": round(nq_c * args.steps / el_c / 1e6, 3), "unit": "M queries/s", "ms_per_step": round(ms, 4),
             "job_frac": round(job_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
             "k_fill_frac": round(8.0 * cn["n_hits"] / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if f_ms > 0 else None,
             "verified": bool(ok), "queries": nq_c, "hits_per_step": cn["n_hits"], "index_device_bytes": info_c["device_bytes"],
             "kernels_avg_ms": {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in st_c.items() if v["launches"]},
             "workload": f"BASELINE configs[{cfg - 1}]: sigma={sg} text {n_c}, ks={ks_c}, lengths {ql_c}, planted {pl_c}", "wall_s": None}
        for r_ in res_c:
            r_.close()
        if not args.no_two_streams:
            el2, res2c, _ = run_sharded(idx_c, args.steps, args.warmup, False, 2, q_c)
            o["two_streams"] = {"value": round(nq_c * args.steps / el2 / 1e6, 3), "ms_per_step": round(el2 / args.steps * 1e3, 4),
                                "same_hit_total": bool(res2c[0].counts()["n_hits"] == cn["n_hits"])}
            for r_ in res2c:
                r_.close()
        if cfg == 3:
            # the same steps on the REFERENCE's planner table (KMX_SEARCH_REFERENCE_PLAN): sums of two different ks stitched as the
            # reference stitches them (kmer_index.hpp:515-557) — the default answers those lengths from the largest k
            el_r, res_r, st_r = run_sharded(idx_c, args.steps, args.warmup, True, 1, q_c, engine.SEARCH_REFERENCE_PLAN)
            o["reference_plan"] = {"value": round(nq_c * args.steps / el_r / 1e6, 3), "ms_per_step": round(el_r / args.steps * 1e3, 4),
                                   "job_frac": round(job_b / (el_r / args.steps) / 1e9 / HBM_PEAK_GBPS, 4),
                                   "same_hit_total": bool(res_r[0].counts()["n_hits"] == cn["n_hits"]),
                                   "kernels_avg_ms": {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in st_r.items() if v["launches"]},
                                   "what": "KMX_SEARCH_REFERENCE_PLAN: every query on the element(s) the reference's planner names (multi-k sums as sums)"}
            for r_ in res_r:
                r_.close()
        if cfg == 3:                                                   # sub-k queries and short reads on the multi-k index
            o["length_probes"] = {f"m={m_}": length_probe(idx_c, text_c, sg, m_, nq_p) for m_, nq_p in ((6, 50_000), (7, 200_000), (13, 500_000), (16, 500_000))}
        idx_c.close()
        del q_c
        o["wall_s"] = round(time.time() - t_c, 1)
        return o

    # (KMX_BENCH_INIT_PG: a 1-rank process group, so that a 1-GPU box still drives the RCCL code path end to end)
    want_gather = backend is not None and args.gather in ("both", "hits")
    elapsed, results, stats = run_sharded(idx, args.steps, args.warmup, True, args.streams)
    res = results[0]
    counts = res.counts()
    # the same K steps with the two result handles on TWO streams (step i+1's lookup, scan and fill may overlap step i's):
    # more queries per second, but concurrent kernels have no meaningful duration of their own, so the roofline line stays
    # with the one-stream leg above and this one is reported beside it
    two_streams = None
    if args.streams == 1 and depth >= 2 and not args.no_two_streams:
        dt2, res2, _ = run_sharded(idx, args.steps, args.warmup, False, 2)
        same2 = res2[0].counts()["n_hits"] == counts["n_hits"]
        t2 = torch.tensor([dt2], dtype=torch.float64, device=comm_dev)
        if backend is not None:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        dt2 = float(t2.item())
        two_streams = {"value": round(nq * world * args.steps / dt2 / 1e6, 3), "unit": "M queries/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                       "same_hit_total": bool(same2), "how": "two result handles on two HIP streams, KMX_SEARCH_ASYNC"}
        for r_ in res2:
            r_.close()
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
    total_hits = counts["n_hits"]
    fill = stats.get("k_fill", {"launches": 0, "total_ms": 0.0})
    fill_ms = fill["total_ms"] / max(fill["launches"], 1)
    per_rank = None
    rccl = None
    if backend is not None:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)                    # max over ranks (the contract's clock)
        totals = kdist.all_gather_totals(nq, counts["n_hits"], device=comm_dev)    # the one exchange: per-shard totals
        total_hits = int(totals[:, 1].sum())
        mine = torch.tensor([fill_ms, float(counts["n_hits"]), float(elapsed)], dtype=torch.float64, device=comm_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu().numpy()
        chk = torch.tensor([rank + 1], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(chk)
        rccl = {"backend": backend + (" (RCCL)" if backend == "nccl" else " (rehearsal: ranks share a device)"),
                "ranks_seen": int(dist.get_world_size()), "all_reduce_ok": int(chk.item()) == world * (world + 1) // 2,
                "devices_visible": n_dev}
    elapsed = float(t_el[0].item())

    # ---- config 3 alone (`--config 3`): the same steps on the reference's planner table (multi-k sums stitched as sums) ----
    ref_plan_leg = None
    if world == 1 and args.config == 3 and not args.no_open_compare:
        el_r, res_r, st_r = run_sharded(idx, args.steps, args.warmup, True, 1, None, engine.SEARCH_REFERENCE_PLAN)
        ref_plan_leg = {"value": round(nq * args.steps / el_r / 1e6, 3), "ms_per_step": round(el_r / args.steps * 1e3, 4),
                        "same_hit_total": bool(res_r[0].counts()["n_hits"] == counts["n_hits"]),
                        "kernels_avg_ms": {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in st_r.items() if v["launches"]},
                        "what": "KMX_SEARCH_REFERENCE_PLAN: every query on the element(s) the reference's planner names (multi-k sums as sums, kmer_index.hpp:515-557)"}
        for r_ in res_r:
            r_.close()

    # ---- the literal north_star variant (open-addressing probe) on the same workload and the same pipeline, outside the timed region ----
    open_leg = None
    if world == 1 and args.config == 2 and not args.no_open_compare and info["tables"] != [engine.TABLE_OPEN] * len(ks):
        idx_o = engine.Index(text, args.sigma, ks, table=engine.TABLE_OPEN, device=dev_index)
        dt_o, res_o, _ = run_sharded(idx_o, args.steps, args.warmup, False)
        same = res_o[0].counts()["n_hits"] == counts["n_hits"]
        open_leg = {"value": round(nq * args.steps / dt_o / 1e6, 3), "unit": "M queries/s", "ms_per_step": round(dt_o / args.steps * 1e3, 4),
                    "table": "open addressing, 16-B slots, load <= 0.5", "pipeline_depth": depth, "same_hit_total": bool(same)}
        for r_ in res_o:
            r_.close()
        idx_o.close()

    # ---- verification of a sample against the CPU oracle (after the timed region) ----
    verified = None
    cpu_baseline = None
    if rank == 0:
        from oracle import orc
        nv = min(args.verify, nq)
        hit_off, positions, status, kinds = res.host()
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            t1 = time.time()
            usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            # a CPU-time quota of the container (cgroup) caps what any number of threads can get: the all-cores leg then says
            # what the quota is worth, not what the host's cores are
            quota_cores = None
            try:
                txt = open("/sys/fs/cgroup/cpu.max").read().split()
                if txt and txt[0] != "max":
                    quota_cores = float(txt[0]) / float(txt[1])
            except Exception:
                try:
                    q_us = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                    p_us = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    if q_us > 0:
                        quota_cores = q_us / p_us
                except Exception:
                    pass
            T = args.cpu_threads or max(1, min(16, usable))
            oidx = orc.Index(text, args.sigma, ks, n_threads=T)
            log(f"oracle (CPU restatement) index built in {time.time() - t1:.1f}s")
            # SURVEY 8d / BASELINE.md 3: one warm-up pass, then >= 5 timed passes; the MEDIAN is the value, min / max beside it.
            # The sample keeps a leg under ~10 s of wall: ns queries on T threads, n1 on one.
            ns = min(args.cpu_sample, nq)
            ref_pool = orc.ref_lib() is not None and hasattr(orc.ref_lib(), "ref_pool_run")   # the REFERENCE's thread_pool.cpp, built in oracle/_ref
            n_pass = max(5, args.cpu_passes)

            def cpu_leg(n_q, threads, pool):
                def one():
                    t0 = time.perf_counter()
                    oidx.search_batch(qr_host[:n_q * m], qoff_host[:n_q + 1], n_threads=threads, keep_hits=False, reference_pool=pool)
                    return time.perf_counter() - t0
                warm = one()
                # (a first pass far slower than the budget: fewer queries per pass, still >= 5 passes)
                while warm * (n_pass + 1) > 12.0 and n_q > 200_000:
                    n_q //= 2
                    warm = one()
                ts = sorted(one() for _ in range(n_pass))
                rates = [n_q / t / 1e6 for t in ts]
                return {"value": round(float(np.median(rates)), 4), "min": round(min(rates), 4), "max": round(max(rates), 4), "passes": n_pass,
                        "queries_per_pass": n_q, "wall_s": round(sum(ts) + warm, 1)}

            leg_T = cpu_leg(ns, T, ref_pool)
            # "T = hardware_concurrency, also T = 1" — the same sample on every usable core (chunked >= 4 T by the batch)
            leg_all = None
            if usable > T and not (quota_cores and quota_cores <= T + 0.5):      # (a quota of T cores: more threads only share them)
                leg_all = cpu_leg(ns, usable, ref_pool)
            leg_1 = cpu_leg(min(ns, 1_000_000), 1, False)         # the same restatement on one thread ("also T=1")
            cpu_model = "unknown CPU"
            try:
                with open("/proc/cpuinfo") as f:
                    cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
            except Exception:
                pass
            cpu_baseline = {"value": leg_T["value"], "unit": "M queries/s", "cores": T, "kind": "port",
                            "min": leg_T["min"], "max": leg_T["max"], "passes": leg_T["passes"],
                            "sample": f"first {leg_T['queries_per_pass']} of the {nq} queries, same 1e8-bp text, the restated search(q).to_vector() per query, tasks carried by "
                                      + ("the reference's own thread_pool (thread_pool.{hpp,cpp} compiled from its sources into oracle/_ref)" if ref_pool
                                         else "the restated thread pool (oracle/_ref not built)")
                                      + f" ({T} threads of {usable} usable, {cpu_model}; std::unordered_map buckets in place of robin_hood); one warm-up pass, "
                                        f"median of {leg_T['passes']} timed passes, {leg_T['wall_s']} s wall",
                            "single_thread_value": leg_1["value"], "single_thread_min": leg_1["min"], "single_thread_max": leg_1["max"],
                            "single_thread_sample": f"first {leg_1['queries_per_pass']} queries, one warm-up pass, median of {leg_1['passes']} timed passes, {leg_1['wall_s']} s wall",
                            "all_cores_value": leg_all["value"] if leg_all else None, "all_cores": usable if leg_all else None,
                            "all_cores_min": leg_all["min"] if leg_all else None, "all_cores_max": leg_all["max"] if leg_all else None,
                            "all_cores_sample": (f"the same {leg_all['queries_per_pass']} queries on all {usable} usable hardware threads, median of {leg_all['passes']} passes, {leg_all['wall_s']} s wall"
                                                 + (f"; the container's CPU quota is {quota_cores:.1f} cores, which is what this leg measures" if quota_cores else "; no CPU quota found (cgroup)")) if leg_all else None,
                            "cpu_quota_cores": quota_cores,
                            "all_cores_skipped": (f"the container's CPU quota is {quota_cores:.1f} cores: {usable} threads would only share them")
                                                 if (leg_all is None and quota_cores and usable > T) else None}
            o_off, o_pos, o_st, _ = oidx.search_batch(qr_host[:nv * m], qoff_host[:nv + 1], n_threads=T)
            verified = bool(np.array_equal(o_off, hit_off[:nv + 1]) and np.array_equal(o_pos, positions[:int(hit_off[nv])]))
            oidx.close()
        else:
            # no oracle index in this leg: every reported position of the first queries must re-read to its query
            verified = reread_ok(text, qr_host, qoff_host, hit_off, positions, min(nv, 2000))
        if not verified:
            log("VERIFICATION FAILED")

    # ---- the PCIe-inclusive rate of the host-buffer form (never `value`): queries in pageable host memory in, hit lists in
    # the result's host memory out — one pass, and streamed in four chunks (chunk i's device-to-host copy under chunk i+1's search)
    host_api = None
    if world == 1 and args.config == 2 and not args.no_host_api and rank == 0:
        def host_leg(chunk):
            if chunk:
                os.environ["KMX_HOST_CHUNK"] = str(chunk)
            try:
                r_h = engine.Result()
                best = None
                for _ in range(2):                                    # the first pass sizes (and page-locks) the host views
                    t1 = time.perf_counter()
                    idx.search(qr_host, qoff_host, result=r_h)
                    ho_h, po_h, _, _ = r_h.host(copy=False)
                    dt_h = time.perf_counter() - t1
                    best = dt_h if best is None else min(best, dt_h)
                same_h = int(ho_h[-1]) == counts["n_hits"] and po_h.size == counts["n_hits"]
                r_h.close()
                return best, same_h
            finally:
                os.environ.pop("KMX_HOST_CHUNK", None)
        dt_one, ok_one = host_leg(0)
        dt_ch, ok_ch = host_leg(max(nq // 4, 1))
        host_api = {"value": round(nq / dt_one / 1e6, 3), "unit": "M queries/s", "ms": round(dt_one * 1e3, 2),
                    "hits_GBps": round(4.0 * counts["n_hits"] / dt_one / 1e9, 1), "same_hit_total": bool(ok_one),
                    "chunked_value": round(nq / dt_ch / 1e6, 3), "chunked_ms": round(dt_ch * 1e3, 2), "chunked_same_hit_total": bool(ok_ch),
                    "what": "kmx_search_batch (queries from pageable host memory) + kmx_result_view (hit lists in host memory), end to end; "
                            "chunked: the same batch streamed through the device in 4 chunks, chunk i's copies under chunk i+1's search"}

    # ---- sub-k and longer-than-k queries on the metric's own index (informational, N = 1 only) ----
    length_probes = None
    if world == 1 and args.config == 2 and not args.no_other_configs and rank == 0 and args.n == n_cfg:
        length_probes = {f"m={m_}": length_probe(idx, text, args.sigma, m_) for m_ in (8, 9, 13, 25)}
        # deeper than the prefix levels: lists merged per query (m = k - 4: 16 runs / 24 K positions, one chunk of k_prefix_merge_block;
        # m = k - 5: 64 runs / 98 K positions, value bands) — few queries, each a long list; fewer of them re-read
        if ks == [10] and args.n >= 50_000_000:
            length_probes.update({f"m={m_}": length_probe(idx, text, args.sigma, m_, nq_p, nv_p=nv_) for m_, nq_p, nv_ in ((6, 20_000, 200), (5, 4_000, 40))})
        if any(not v["verified"] for v in length_probes.values()):
            verified = False
            log("VERIFICATION FAILED in length_probes")

    other = None
    if world == 1 and args.config == 2 and not args.no_other_configs and not args.sort_queries and args.nq == nq_cfg and args.n == n_cfg:
        other = {}
        for cfg in (3, 4, 5):
            try:
                other[str(cfg)] = other_config(cfg)
                log(f"config {cfg}: {other[str(cfg)]['value']} M queries/s, verified={other[str(cfg)]['verified']}, {other[str(cfg)]['wall_s']} s")
            except Exception as e:                                    # reported, not fatal: the headline stands on its own
                other[str(cfg)] = {"error": f"{type(e).__name__}: {e}"}
        if any(isinstance(v, dict) and (v.get("verified") is False or any(not p_["verified"] for p_ in v.get("length_probes", {}).values()))
               for v in other.values()):
            verified = False
            log("VERIFICATION FAILED in other_configs")

    out = None
    if rank == 0:
        n_total_q = nq * world
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total_q * args.steps / elapsed / 1e6
        n_hits_rank = counts["n_hits"]
        # algorithmic bytes: per query R = m + 16 + 4c, W = 8 + 4c.  k_fill moves the 4c + 4c part.
        fill_bytes = 8.0 * n_hits_rank
        read_bytes = float(n_letters) + 16.0 * nq + 4.0 * n_hits_rank
        job_bytes = read_bytes + 8.0 * nq + 4.0 * n_hits_rank
        achieved = fill_bytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0
        kernels_ms = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in stats.items() if v["launches"]}
        # HBM traffic of the dominant kernel from the committed PMC profile of this same command (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 read correction) — only when the workload matches it.
        traffic, traffic_src = None, None
        for name in ("pmc_summary_current.json", f"pmc_summary_current_cfg{args.config}.json"):
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", name)))
                if prof["k_fill"]["algorithmic_bytes_per_launch"] == int(fill_bytes):
                    traffic, traffic_src = prof["k_fill"]["hbm_bytes_per_launch"], "profiles/" + name
                    break
            except Exception:
                pass
        roofline = {"bound": "hbm", "kernel": "k_fill", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": fill_bytes, "avg_launch_ms": round(fill_ms, 4),
                    "job_algorithmic_GBps": round(job_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                    "job_frac": round(job_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    # north_star's wording is the READ roofline: sum of R over the step time (per GPU), next to R+W above
                    "read_only_GBps": round(read_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                    "read_only_frac": round(read_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        if per_rank is not None:
            roofline["per_rank_frac"] = [round(8.0 * h / (f * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if f > 0 else 0.0 for f, h, _ in per_rank]
            roofline["per_rank_step_ms"] = [round(e / args.steps * 1e3, 4) for _, _, e in per_rank]
        out = {
            "metric": "M queries/sec, DNA4 k=10 exact-match batch search, 1e8-bp text" if args.config == 2 else f"M queries/sec, BASELINE configs[{args.config - 1}] (informational)",
            "value": round(value, 3),
            "unit": "M queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32/u64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: sigma={args.sigma} text {args.n} letters, ks={ks}, {nq} queries per GPU per step "
                                   f"(lengths {qlens}, planted share {planted}), materialised sorted position lists (to_vector), table={args.table}"
                                   f"{'(dense)' if info['tables'][0] == engine.TABLE_DENSE else '(open)'}",
                       "queries_per_gpu": nq, "hits_per_step_per_gpu": n_hits_rank, "total_hits_all_gpus": total_hits,
                       "index_device_bytes": info["device_bytes"], "index_memory": idx.memory(), "parallelism": f"query-shard x{world}, index replicated",
                       "gather": "totals", "pipeline_depth": depth, "streams": max(1, min(args.streams, depth))},
            "roofline": roofline,
            "rccl": rccl,
            "gather_hits": None,
            "two_streams": two_streams,
            "open_addressing_table": open_leg,
            "reference_plan": ref_plan_leg,
            "cpu_baseline": cpu_baseline,
            "host_api": host_api,
            "length_probes": length_probes,
            "other_configs": other,
            "kernels_avg_ms": kernels_ms,
            "verified_vs_oracle": verified,
        }
    # ---- leg 2 last, under a deadline: everything `value` needs is already in `out`, so an exchange that fails or
    # never returns costs the line its gather_hits object (an "error" entry instead) and nothing else ----
    if want_gather:
        import threading
        deadline = float(os.environ.get("KMX_BENCH_GATHER_DEADLINE", "300"))
        once = threading.Lock()

        def give_up():
            if not once.acquire(blocking=False):
                return
            if rank == 0:
                out["gather_hits"] = {"error": f"the gather leg did not finish within {deadline:.0f} s; abandoned"}
                if args.gather == "hits":                             # the gather leg WAS the metric asked for: no number, not the other leg's
                    out["value"], out["gather_failed"] = None, True
                print(json.dumps(out), flush=True)
            log(f"gather leg abandoned after {deadline:.0f} s")
            os._exit(gather_exit_code())

        def gather_exit_code():
            # --gather hits asked for the gather metric: a leg that failed or was abandoned is a failed run (5), not rc 0 with
            # the sharded leg's numbers; with --gather both the sharded leg is the metric and the line stands
            if not (verified is None or verified):
                return 4
            return 5 if (args.gather == "hits" and rank == 0) else 0      # (rank 0 only, after it has printed: a launcher ends the
                                                                             # other ranks when one fails)

        guard = threading.Timer(deadline, give_up)
        guard.daemon = True
        guard.start()
        gather_leg, gather_err = None, None
        try:
            gather_leg = run_gather(idx, args.steps, args.warmup)
            t_g = torch.tensor([gather_leg["elapsed"]], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t_g, op=dist.ReduceOp.MAX)
            gather_leg["elapsed"] = float(t_g.item())
        except Exception as e:                                        # reported, not fatal: leg 1 stands on its own
            gather_leg, gather_err = None, f"{type(e).__name__}: {e}"
            log("gather leg failed: " + gather_err)
        guard.cancel()
        if not once.acquire(blocking=False):                          # the guard is already printing the line
            time.sleep(3600)
        if rank == 0:
            if gather_leg:
                n_total_q = nq * world
                g_ms_step = gather_leg["elapsed"] / args.steps * 1e3
                peers = [b for b in gather_leg["bytes_per_peer"] if b]
                per_link = (max(peers) / (gather_leg["gather_ms"] * 1e-3) / 1e9) if peers and gather_leg["gather_ms"] > 0 else 0.0
                out["gather_hits"] = {
                    "value": round(n_total_q * args.steps / gather_leg["elapsed"] / 1e6, 3), "unit": "M queries/s",
                    "ms_per_step": round(g_ms_step, 4), "gather_ms": round(gather_leg["gather_ms"], 4),
                    "bytes_per_peer_per_step": max(peers) if peers else 0,
                    "per_link_GBps": round(per_link, 1), "per_link_frac_of_xgmi": round(per_link / XGMI_LINK_GBPS, 4),
                    "root_ingress_GBps": round(sum(peers) / (gather_leg["gather_ms"] * 1e-3) / 1e9, 1) if peers and gather_leg["gather_ms"] > 0 else 0.0,
                    "gathered_hits": gather_leg.get("gathered_hits"),
                    "how": "dist.batch_isend_irecv: one grouped exchange, all peers' links at once, into buffers kept between "
                           "steps; the gather of step i overlaps the search of step i+1 (two handles, two compute streams)"}
                if gather_leg.get("gathered_hits") != total_hits:     # the gathered arrays must account for every shard
                    verified = False
                    out["verified_vs_oracle"] = False
                    log("VERIFICATION FAILED: gathered hits != sum of the shards")
                if args.gather == "hits":
                    out["value"], out["ms_per_step"] = out["gather_hits"]["value"], out["gather_hits"]["ms_per_step"]
                    out["config"]["gather"] = "hits"
            else:
                out["gather_hits"] = {"error": gather_err}
                if args.gather == "hits":
                    out["value"], out["gather_failed"] = None, True
    if rank == 0:
        print(json.dumps(out), flush=True)
    if want_gather and gather_err is not None:                        # the process group may be unusable: no teardown through it
        sys.stderr.flush()
        os._exit(gather_exit_code())
    for r_ in results:
        r_.close()
    idx.close()
    if backend is not None:
        dist.destroy_process_group()
    return 0 if (verified is None or verified) else 4


if __name__ == "__main__":
    main()
