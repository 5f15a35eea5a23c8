"""bench.py --gpus N: the script launches its own ranks when no launcher did, refuses to measure fewer GPUs than it was
asked for, and at N > 1 headlines the sharded (hash-partitioned MAG240M-shaped) workload with the replica run as a sub-record."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_more_ranks_than_devices_is_an_error_not_a_one_gpu_run():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--small"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 2 and "needs" in p.stderr and "visible HIP devices" in p.stderr
    assert not p.stdout.strip()  # no JSON line: nothing was measured


def test_world_size_must_match_gpus():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--small"],
                       env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr and not p.stdout.strip()


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_gpu():
    """GIGL_BENCH_SHARE_GPU=1: both ranks on device 0, gloo collectives, the library's host-callback transport — the
    launcher, the per-rank root sharding, the max-over-ranks timing and the sharded sub-record run end to end"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--small", "--no-cpu-baseline", "--min-seconds", "0.3",
                        "--min-reps", "2", "--min-rounds", "2", "--group", "8", "--shard-group", "4", "--shard-scale",
                        "0.0005", "--steps", "16", "--warmup", "8"],
                       env=_env(GIGL_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    # the N > 1 headline IS the hash-partitioned MAG240M-shaped workload, and says between how many ranks its exchanges ran
    assert "MAG240M-shaped" in line["config"]["workload"] and "hash-partitioned over 2 rank(s)" in line["config"]["workload"]
    assert line["comm"]["ranks"] == 2 and line["comm"]["transport"] == "host-callback"  # (gloo here; "rccl" on real GPUs)
    assert line["rccl_ranks"] == 0  # ... so no RCCL rank is claimed for this functional run
    assert line["comm"]["xgmi_bytes_per_step_per_gpu_mean"] > 0
    assert line["roofline_xgmi"]["ranks"] == 2
    assert line["config"]["pulled_feature_rows_per_step"] > 0  # rows really travelled between the ranks
    # same node, one process per rank: the peer-mapped route (the ranks read each other's tables through hipIpc mappings)
    # — no row ever sits in a bucket; the bucketed route reports how full its row buckets were
    if line["config"]["feature_route"] == "peer":
        assert line["config"]["row_bucket_fill"] is None and line["config"]["feature_route_note"] is None
        assert line["config"]["hop_route"] == "peer-sampled"  # (the ranks' graph shards are mapped as well: no exchange at all)
    else:
        assert 0 < line["config"]["row_bucket_fill"] <= 1.0
    assert "replicated as hot rows" in line["config"]["workload"]  # hub replication is on by default at world > 1
    rep = line["replicas"]  # the replica-per-GPU run of the same launch: a sub-record now
    assert rep["n_gpus"] == 2 and rep["value"] > 0 and "replica per GPU" in rep["config"]["graph"]


@pytest.mark.gpu
def test_a_failing_sharded_sub_record_never_costs_the_headline():
    """the sub-record's collectives have not run on a real multi-GPU node yet: it runs in child processes, so if it
    crashes, fails or hangs, rank 0 still prints the (complete) headline line, with the failure noted"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--small", "--no-cpu-baseline", "--min-seconds", "0.3",
                        "--min-reps", "2", "--min-rounds", "2", "--group", "8", "--shard-group", "4", "--shard-scale",
                        "0.0005", "--steps", "16", "--warmup", "8"],
                       env=_env(GIGL_BENCH_SHARE_GPU="1", GIGL_BENCH_SUB_TIMEOUT="0.001"), capture_output=True, text=True,
                       timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["roofline"] is not None
    assert "did not finish" in line["sharded"]["error"]
    assert line["headline_is"].startswith("FALLBACK") and "replica per GPU" in line["config"]["graph"]


@pytest.mark.gpu
def test_under_torchrun_the_sharded_headline_rendezvous_works():
    """the driver launches N > 1 as `python -m torch.distributed.run ... bench.py --gpus N`: the sharded workload's child
    processes must host their own rendezvous store (torchrun's TORCHELASTIC_USE_AGENT_STORE would make them wait for an
    agent store on their port until the time limit, and the line would fall back to the replica run)"""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29671", BENCH, "--gpus", "2", "--small", "--no-cpu-baseline",
                        "--min-seconds", "0.3", "--min-reps", "2", "--min-rounds", "2", "--group", "8", "--shard-group", "4",
                        "--shard-scale", "0.0005", "--steps", "16", "--warmup", "8"],
                       env=_env(GIGL_BENCH_SHARE_GPU="1", GIGL_BENCH_SUB_TIMEOUT="300"), capture_output=True, text=True,
                       timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "sharded" not in line, line.get("sharded")
    assert "MAG240M-shaped" in line["config"]["workload"] and line["comm"]["ranks"] == 2
    assert line["headline_is"].startswith("mag240m-sharded") and line["replicas"]["value"] > 0
