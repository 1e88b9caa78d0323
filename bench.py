#!/usr/bin/env python3
"""bench.py — batch exact-match search throughput (BASELINE.json metric) on N GPUs of one node.

One "step" = one pass of the hot path over one batch: 1e7 uniform random DNA4 10-mers per GPU against a
k=10 index of a 1e8-bp synthetic text (BASELINE.json configs[1]); inputs resident in HBM before the timed
region; the index is replicated per GPU and every rank searches its own query shard (weak scaling, no
data-path collective; per-rank totals are exchanged once after the timed region).

Prints ONE JSON line on rank 0 (see the bench contract in the task / DESIGN.md §6).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[N-1]; 2 = the metric's workload (default), 3/4/5 are informational")
    ap.add_argument("--n", type=int, default=0, help="text length (0 = the config's)")
    ap.add_argument("--nq", type=int, default=0, help="queries per GPU per step (0 = the config's)")
    ap.add_argument("--table", choices=["open", "dense", "auto"], default="auto",
                    help="auto = the engine's default policy (direct addressing when sigma^k <= 4(n-k+1), else open addressing)")
    ap.add_argument("--no-open-compare", action="store_true",
                    help="skip the extra leg that times the same workload on the open-addressing table (N=1, config 2 only)")
    ap.add_argument("--gather", choices=["totals", "hits"], default="totals",
                    help="totals: hit lists stay sharded where they were produced, per-shard totals exchanged after the timed region "
                         "(default, zero data-path collective); hits: RCCL gatherv of every hit list to rank 0 inside each step")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (and result handles) the steps alternate over; >1 lets step i+1's lookup/scan overlap step i's fill")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="result handles the steps rotate over on ONE stream with KMX_SEARCH_ASYNC: the host enqueues step i+1 while "
                         "step i runs and reads step i's counters when its handle comes round again (1 = every step waits for its own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="queries in the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline threads (0 = min(16, usable cores): the box's CPU share)")
    ap.add_argument("--verify", type=int, default=20000, help="queries checked against the oracle after timing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from kmer_index_amd import dist as kdist
    from kmer_index_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)          # (== local_rank on a real N-GPU node)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("KMX_DIST_BACKEND", "nccl")   # "gloo" only to rehearse N ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    # (sigma, n, ks, queries per GPU, query lengths, planted share, text seed, query seed)
    CFG = {2: (4, 100_000_000, [10], 10_000_000, [10], 0.0, 1002, 2002),
           3: (4, 100_000_000, [8, 10, 12], 10_000_000, [8, 10, 12, 20, 22, 24], 0.5, 1003, 2003),
           4: (5, 100_000_000, [10], 12_500_000, [10], 0.0, 1004, 2004),
           5: (20, 10_000_000, [5], 10_000_000, [5], 0.5, 1005, 2005)}
    sigma, n_cfg, ks, nq_cfg, qlens, planted, tseed, qseed = CFG[args.config]
    args.sigma, args.k = sigma, ks[0] if len(ks) == 1 else 0
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

    # this rank's query shard: letters [rank*nq*m, (rank+1)*nq*m) of query stream 2002
    nq = args.nq
    if planted == 0.0:
        m = qlens[0]
        qr_host = np.empty(nq * m, np.uint8)
        chunk = 1 << 24
        for s in range(0, nq * m, chunk):
            e = min(nq * m, s + chunk)
            z = synth.u64_stream(qseed, e - s, rank * nq * m + s)
            qr_host[s:e] = (((z >> np.uint64(32)) * np.uint64(args.sigma)) >> np.uint64(32)).astype(np.uint8)
        qoff_host = np.arange(nq + 1, dtype=np.uint64) * np.uint64(m)
    else:
        m = 0
        qr_host, qoff_host = synth.mixed_queries(qseed + 7919 * rank, text, nq, qlens, args.sigma, planted_frac=planted)
    n_letters = int(qoff_host[-1])
    d_qr = torch.from_numpy(qr_host).to(dev)
    d_qoff = torch.from_numpy(qoff_host.view(np.int64)).to(dev)
    torch.cuda.synchronize()

    n_streams = max(1, args.streams)
    depth = max(1, args.pipeline) if n_streams == 1 else 1
    t_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    results = [engine.Result() for _ in range(max(n_streams, depth))]
    res = results[0]
    step_no = [0]
    step_flags = engine.SEARCH_ASYNC if depth > 1 else engine.SEARCH_DEFAULT

    def step():
        i = step_no[0] % len(results)
        step_no[0] += 1
        idx.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, flags=step_flags, stream=t_streams[i % n_streams].cuda_stream, result=results[i])
        if world > 1 and args.gather == "hits":
            with torch.cuda.stream(t_streams[i]):
                t_off, t_pos = results[i].device_tensors(dev)
                kdist.gather_hit_lists(t_off, t_pos, dst=0)

    # setup, like the index build: every result handle allocates its device buffers (grow-only, sized by its first
    # batch) before the W warmup steps, so that neither warmup nor the timed region holds an allocation
    for r_ in results:
        for _ in range(2):
            idx.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, stream=t_streams[0].cuda_stream, result=r_)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    for r_ in results:
        r_.counts()                                   # no warmup step is left pending into the timed region
    torch.cuda.synchronize()
    idx.stats_enable(True)
    idx.stats_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for r_ in results:
        r_.counts()                                   # completes a step that is still pending on its handle (inside the timed region)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stats = idx.stats()
    idx.stats_enable(False)
    counts = res.counts()

    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    total_hits = counts["n_hits"]
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        totals = kdist.all_gather_totals(nq, counts["n_hits"], device=dev)    # the one exchange: per-shard totals
        total_hits = int(totals[:, 1].sum())
    elapsed = float(t_el.item())

    # ---- the literal north_star variant (open-addressing probe) on the same workload, outside the timed region ----
    open_leg = None
    if world == 1 and args.config == 2 and not args.no_open_compare and info["tables"] != [engine.TABLE_OPEN] * len(ks):
        idx_o = engine.Index(text, args.sigma, ks, table=engine.TABLE_OPEN, device=dev_index)
        res_o = engine.Result()
        for _ in range(args.warmup):
            idx_o.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, stream=t_streams[0].cuda_stream, result=res_o)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            idx_o.search_device(d_qr.data_ptr(), d_qoff.data_ptr(), nq, stream=t_streams[0].cuda_stream, result=res_o)
        torch.cuda.synchronize()
        dt_o = time.perf_counter() - t1
        same = res_o.counts()["n_hits"] == counts["n_hits"]
        open_leg = {"value": round(nq * args.steps / dt_o / 1e6, 3), "unit": "M queries/s", "ms_per_step": round(dt_o / args.steps * 1e3, 4),
                    "table": "open addressing, 16-B slots, load <= 0.5", "same_hit_total": bool(same)}
        res_o.close()
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
            T = args.cpu_threads or max(1, min(16, usable))
            oidx = orc.Index(text, args.sigma, ks, n_threads=T)
            log(f"oracle (CPU restatement) index built in {time.time() - t1:.1f}s")
            ns = min(args.cpu_sample, nq)
            t1 = time.perf_counter()
            oidx.search_batch(qr_host[:ns * m], qoff_host[:ns + 1], n_threads=T, keep_hits=False)
            dt = time.perf_counter() - t1
            n1 = min(ns, 1_000_000)                               # the same restatement on one thread ("also T=1")
            t1 = time.perf_counter()
            oidx.search_batch(qr_host[:n1 * m], qoff_host[:n1 + 1], n_threads=1, keep_hits=False)
            dt1 = time.perf_counter() - t1
            cpu_model = "unknown CPU"
            try:
                with open("/proc/cpuinfo") as f:
                    cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
            except Exception:
                pass
            cpu_baseline = {"value": round(ns / dt / 1e6, 4), "unit": "M queries/s", "cores": T, "kind": "port",
                            "sample": f"first {ns} of the {nq} queries, same 1e8-bp text, search(q).to_vector() per query "
                                      f"on the oracle's thread pool ({T} threads of {usable} usable, {cpu_model}; "
                                      f"std::unordered_map buckets), {dt:.1f}s wall",
                            "single_thread_value": round(n1 / dt1 / 1e6, 4), "single_thread_sample": f"first {n1} queries, {dt1:.1f}s"}
            o_off, o_pos, o_st, _ = oidx.search_batch(qr_host[:nv * m], qoff_host[:nv + 1], n_threads=T)
            verified = bool(np.array_equal(o_off, hit_off[:nv + 1]) and np.array_equal(o_pos, positions[:int(hit_off[nv])]))
            oidx.close()
        else:
            # no oracle index in this leg: every reported position of the first queries must re-read to its query
            nv = min(nv, 2000)
            cntv = np.diff(hit_off[:nv + 1]).astype(np.int64)
            qi = np.repeat(np.arange(nv), cntv)
            pos = positions[:int(hit_off[nv])].astype(np.int64)
            lens = np.diff(qoff_host[:nv + 1]).astype(np.int64)
            ok = True
            for j in range(int(lens.max()) if nv else 0):
                sel = lens[qi] > j
                ok &= bool(np.array_equal(text[pos[sel] + j], qr_host[qoff_host[qi[sel]].astype(np.int64) + j]))
            verified = bool(ok)
        if not verified:
            log("VERIFICATION FAILED")

    if rank == 0:
        n_total_q = nq * world
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total_q * args.steps / elapsed / 1e6
        fill = stats.get("k_fill", {"launches": 0, "total_ms": 0.0})
        fill_ms = fill["total_ms"] / max(fill["launches"], 1)
        n_hits_rank = counts["n_hits"]
        # algorithmic bytes: per query R = m + 16 + 4c, W = 8 + 4c.  k_fill moves the 4c + 4c part.
        fill_bytes = 8.0 * n_hits_rank
        job_bytes = float(n_letters) + float(nq) * (16 + 8) + 8.0 * n_hits_rank
        achieved = fill_bytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0
        kernels_ms = {k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in stats.items() if v["launches"]}
        # HBM traffic of the dominant kernel from the committed PMC profile of this same command (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 read correction) — only when the workload matches it.
        traffic, traffic_src = None, None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary_current.json")))
            if prof["k_fill"]["algorithmic_bytes_per_launch"] == int(fill_bytes):
                traffic, traffic_src = prof["k_fill"]["hbm_bytes_per_launch"], "profiles/pmc_summary_current.json"
        except Exception:
            pass
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
                       "index_device_bytes": info["device_bytes"], "parallelism": f"query-shard x{world}, index replicated", "gather": args.gather, "streams": n_streams, "pipeline_depth": depth},
            "roofline": {"bound": "hbm", "kernel": "k_fill", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": fill_bytes, "avg_launch_ms": round(fill_ms, 4),
                         "job_algorithmic_GBps": round(job_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                         "job_frac": round(job_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "open_addressing_table": open_leg,
            "cpu_baseline": cpu_baseline,
            "kernels_avg_ms": kernels_ms,
            "verified_vs_oracle": verified,
        }
        print(json.dumps(out), flush=True)
    for r_ in results:
        r_.close()
    idx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
