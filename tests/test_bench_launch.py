"""bench.py's multi-rank launch forms: `python bench.py --gpus N` spawns the ranks itself and never
reports a smaller run as N; the RCCL ("nccl") path runs whenever the box has the devices for it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--text-len", "300000", "--nq", "40000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-open-compare"]


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    line = next((l for l in p.stdout.splitlines()[::-1] if l.startswith("{")), None)
    return p.returncode, (json.loads(line) if line else None), p.stderr


def _n_devices():
    import torch
    return torch.cuda.device_count()


def test_more_gpus_than_visible_is_refused_not_downgraded():
    """Launched plainly with --gpus N on a box with fewer devices: non-zero exit and no JSON line (round 1 printed n_gpus 1)."""
    n = _n_devices()
    rc, out, err = _run(["--gpus", str(n + 2)] + SMALL)
    assert rc != 0 and out is None
    assert "refusing" in err


def test_world_size_must_match_gpus():
    rc, out, err = _run(["--gpus", "1"] + SMALL, env_extra={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert rc != 0 and out is None and "WORLD_SIZE" in err


@pytest.mark.gpu
def test_single_rank_process_group_over_rccl():
    """One rank, backend nccl: init, all_reduce, all_gather and the gather leg's stream/event choreography run through RCCL
    on the one GPU of the test box."""
    rc, out, err = _run(["--gpus", "1", "--gather", "both"] + SMALL, env_extra={"KMX_BENCH_INIT_PG": "1", "MASTER_PORT": "29533"})
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 1 and out["verified_vs_oracle"] is True
    assert out["rccl"]["backend"].startswith("nccl") and out["rccl"]["all_reduce_ok"] and out["rccl"]["ranks_seen"] == 1
    assert out["gather_hits"]["gathered_hits"] == out["config"]["total_hits_all_gpus"]


@pytest.mark.gpu
def test_two_ranks_spawned_by_bench_itself_rehearsal():
    """`python bench.py --gpus 2 --oversubscribe`: the launcher spawns two ranks that share the test box's GPU (gloo carries
    the exchange); both legs run and the gathered arrays account for every shard."""
    rc, out, err = _run(["--gpus", "2", "--oversubscribe"] + SMALL)
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["verified_vs_oracle"] is True
    assert out["rccl"]["ranks_seen"] == 2 and out["rccl"]["all_reduce_ok"]
    g = out["gather_hits"]
    assert g["gathered_hits"] == out["config"]["total_hits_all_gpus"] and g["ms_per_step"] > 0 and g["bytes_per_peer_per_step"] > 0
    assert len(out["roofline"]["per_rank_frac"]) == 2


@pytest.mark.gpu
def test_two_ranks_under_the_distributed_launcher_rehearsal():
    """The driver's launch form: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the launcher); two ranks share the test box's GPU, so the exchange runs over gloo."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29587",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe"] + SMALL
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = next((l for l in p.stdout.splitlines()[::-1] if l.startswith("{")), None)
    assert p.returncode == 0 and line, p.stderr[-2000:]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["verified_vs_oracle"] is True and out["rccl"]["ranks_seen"] == 2
    assert out["gather_hits"]["gathered_hits"] == out["config"]["total_hits_all_gpus"]


@pytest.mark.gpu
def test_a_gather_leg_that_never_returns_costs_only_its_own_object():
    """The gather leg runs last under a deadline (KMX_BENCH_GATHER_DEADLINE): when it passes, rank 0 still prints the line —
    `value`, roofline and verification from the sharded leg, an error entry in place of gather_hits — and every rank exits 0."""
    rc, out, err = _run(["--gpus", "2", "--oversubscribe"] + SMALL, env_extra={"KMX_BENCH_GATHER_DEADLINE": "0.001"})
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["verified_vs_oracle"] is True and out["roofline"]["frac"] > 0
    assert "did not finish" in out["gather_hits"]["error"]


@pytest.mark.gpu
def test_a_failed_gather_leg_is_a_failed_run_when_the_gather_was_the_metric():
    """--gather hits asks for the gather leg's number as `value`: when that leg is abandoned the line says so (value null,
    gather_failed) and the ranks exit 5 — never rc 0 with the sharded leg's numbers under the gather's name."""
    rc, out, err = _run(["--gpus", "2", "--oversubscribe", "--gather", "hits"] + SMALL, env_extra={"KMX_BENCH_GATHER_DEADLINE": "0.001"})
    assert rc == 5, err[-2000:]
    assert out["value"] is None and out["gather_failed"] is True and "did not finish" in out["gather_hits"]["error"]


@pytest.mark.gpu
def test_n_ranks_over_rccl_when_the_box_has_the_devices():
    """The real thing: one rank per GPU over RCCL/xGMI.  Skipped (not faked with gloo) on a 1-GPU box."""
    n = _n_devices()
    if n < 2:
        pytest.skip("needs >= 2 GPUs for an RCCL exchange between ranks")
    n = min(n, 4)
    rc, out, err = _run(["--gpus", str(n)] + SMALL)
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == n and out["rccl"]["backend"].startswith("nccl") and out["rccl"]["ranks_seen"] == n
    assert out["gather_hits"]["gathered_hits"] == out["config"]["total_hits_all_gpus"]
    assert out["gather_hits"]["per_link_GBps"] > 0
