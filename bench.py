#!/usr/bin/env python3
"""bench.py — sampled+aggregated edges/s of the HIP hot path (BASELINE.json metric).

A "step" = one batch of B roots through the whole path, inputs already resident in HBM:
    k-hop sample (parity mode, Spark-hash permutation)  ->  batch union graph (dedup + CSR)
    ->  GraphSAGE forward (gather-mean + fp32 MFMA projection per layer, trimmed schedule)
    ->  one embedding row per root
enqueued by ONE library call (gigl_sage_plan_run).  Workload at N=1 = BASELINE.json configs[1]:
ogbn-products-SHAPED synthetic graph (N=2,449,029, RMAT(.57,.19,.19) power-law, ~61.9M undirected
pairs bidirectionalised, D=100 fp32), fanout [25,10], B=1024, GraphSAGE 100->256->47 (SURVEY.md
§8(d) C2).  MAG240M (the config the metric is quoted on) does not fit one GPU (375 GB of features),
so per the contract the N=1 line is the largest single-GPU configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--group G] [--workload W]

Steps are independent batches.  One library call takes G consecutive batches through ONE set of ~20 launches
(gigl_sage_plan_set_groups: every batch keeps its own union graph, rows are bit-identical to G single-batch
calls — tests/test_gpu_groups.py); calls are pipelined over S HIP streams (one library ctx + one host thread per
stream, all sharing the HBM-resident graph and hash table; defaults S = 3, G = 64 — a sweep of S in 2..6, G in
16..128 stays within 7 % of the best).  The regime does not depend on --steps: --steps is a MINIMUM, the timed range is
a whole number of rounds (S*G steps) repeated until the timed region lasts >= --min-seconds; median / p10 / p90 over
the repetitions are reported next to the aggregate.
N>1: one process per GPU (torch.distributed, RCCL); every rank holds a replica of the graph and takes
its own root batches — the path shards by roots with no data-path collective ("weak" scaling);
time = max over ranks, value = total edges of all ranks / that time.

Counting: sampled edge = one (src->dst) pair emitted by a hop expansion before batch
dedup; aggregated edge = one edge actually consumed by one layer's segmented reduce (sum_l |E_l|,
trimmed schedule — never the inflated L*|E_union|, which is reported in config for context).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def emit(line: dict) -> None:
    """the run's ONE JSON line, as the last line of stdout: what native libraries left in C stdio's buffer (RCCL's
    version banner, printed at communicator creation and otherwise flushed at exit, after this line) goes out first"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # same guide: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
# The projection runs split-precision: every fp32 product is SIX bf16 MFMA products (agg.hip: linear_split_kernel), so
# its bound in fp32-equivalent FLOP/s is the dense bf16 matrix peak (~2.5 PFLOP/s, same guide) / 6
MFMA_SPLIT_PEAK_TF = 2500.0 / 6.0
# ... and THREE fp16 products where the library could bound the operands inside the fp16 range (linear_split_kernel<.., HS>,
# gigl_sage_plan_half_split): the projection is priced in 16-bit MFMA products actually issued against the dense peak
MFMA_16BIT_PEAK_TF = 2500.0

# library timer id -> name prefixes of the device functions it brackets (as rocprofv3 prints them, scripts/pmc_summary.py)
PMC_KERNELS = {
    "expand": ["plan_rows_kernel", "expand_rows_kernel"],
    # (round 5, fused layers: the last layer is sage_fused_out_kernel, both projections linear_fused2_kernel)
    "gather_mean": ["gather_mean_kernel", "sage_fused_out_kernel"],
    "linear": ["linear_split_kernel", "linear_lds_kernel", "linear_mfma_kernel", "linear_fused2_kernel"],
    # (the one-call plan's two-hop union build, union.hip "LG2"; the generic build's kernels have other names)
    # (round 4: the LDS-staged build "LG3" — lg3_* — replaced lg2_insert / extras / count / assign / fill)
    "union_insert": ["lg2_init_kernel", "lg2_insert_kernel", "lg2_extras_kernel", "lg3_init_kernel", "lg3_dedup_kernel"],
    "union_nodes": ["lg2_count_kernel", "lg2_assign_kernel", "lg3_assign_kernel"],
    "union_edge_sort": ["lg2_fill_kernel", "lg3_fill_kernel"],
    "union_csr": ["lg2_row_sort_tiny_kernel", "lg2_row_sort_kernel", "lg2_row_sort_big_kernel"],
}
_LIVE_PMC = {}  # workload-shape key -> summary collected by THIS run (collect_live_pmc)


def collect_live_pmc(extra_args, timeout_s: float = 420.0, env_extra=None):
    """HBM traffic per kernel measured by THIS run: the four rocprofv3 passes of scripts/gpu_pmc.sh (FETCH_SIZE and
    WRITE_SIZE in separate passes, --kernel-trace only, each over the known-byte calibration launches and over a short
    single-stream eager --timed-only run of this same workload) as child processes once the timed region is over, folded
    by scripts/pmc_summary.py with the guide's calibration.  -> (summary dict, None) or (None, reason)."""
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tag = f"live{os.getpid()}"
    out_root = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_root, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", GIGL_BENCH_CHILD="1", **(env_extra or {}))
    t0 = time.time()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            base = os.path.join(out_root, f"pmc_{tag}_{ctr}")
            for what, cmd in (("calib", [sys.executable, os.path.join(ROOT, "scripts", "pmc_calib.py")]),
                              ("bench", [sys.executable, os.path.abspath(__file__), "--streams", "1", "--no-graph", "--steps",
                                         "64", "--min-rounds", "2", "--warmup", "32", "--timed-only", "--no-cpu-baseline",
                                         "--no-live-pmc"] + list(extra_args))):
                left = timeout_s - (time.time() - t0)
                if left < 20:
                    return None, f"live PMC passes did not fit {timeout_s:.0f} s"
                with open(os.path.join(out_root, f"pmc_{tag}_{ctr}_{what}.log"), "w") as log:
                    cp = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "-f", "csv", "-d", os.path.join(base, what),
                                         "-o", what, "--"] + cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT,
                                        timeout=left)
                if cp.returncode != 0:
                    return None, f"rocprofv3 --pmc {ctr} ({what}) exited with {cp.returncode}"
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), tag], cwd=ROOT, env=env,
                            capture_output=True, text=True, timeout=120)
        if cp.returncode != 0:
            return None, f"pmc_summary failed: {cp.stderr.strip()[-200:]}"
        doc = json.load(open(os.path.join(out_root, f"pmc_{tag}.json")))
        doc["collected_s"] = round(time.time() - t0, 1)
        return doc, None
    except subprocess.TimeoutExpired:
        return None, f"live PMC passes did not finish within {timeout_s:.0f} s"
    except Exception as ex:  # noqa: BLE001
        return None, f"{type(ex).__name__}: {str(ex)[:200]}"
    finally:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            shutil.rmtree(os.path.join(out_root, f"pmc_{tag}_{ctr}"), ignore_errors=True)


def step_traffic_of(doc, steps_executed: int, prefixes=None, min_calls=None):
    """HBM bytes per step from a live PMC summary of a child run that executed `steps_executed` steps of the path (warm-up
    included): sum over the LIBRARY's kernels launched at least `min_calls` times (default: once per step) — setup
    kernels (graph build, threshold table) run a handful of times and drop out, torch / rocPRIM / runtime kernels
    (synthetic tables, copies of a few bytes) are left out by name — of bytes per launch x launches, / steps.
    `prefixes`: only kernels whose name starts with one of them.  -> (bytes per step, {kernel: bytes per step})"""
    per = {}
    need = steps_executed if min_calls is None else min_calls
    for name, e in doc.get("kernels", {}).items():
        calls = e.get("FETCH_SIZE_calls", 0)
        if calls < need or (prefixes is not None and not any(name.startswith(p) for p in prefixes)):
            continue
        if prefixes is None and any(t in name for t in ("at::", "rocprim", "hiprand", "__amd_rocclr", "elementwise")):
            continue
        per[name] = e["hbm_bytes_per_launch"] * calls / steps_executed
    return sum(per.values()), per


def pmc_traffic(kernel_id: str, batches_per_call: int, workload: str = "products", projected: bool = False):
    """HBM-side bytes per launch of `kernel_id` from the newest committed rocprofv3 PMC summary OF THIS WORKLOAD
    (profiles/*_pmc*.json: FETCH_SIZE and WRITE_SIZE collected in separate passes by scripts/gpu_pmc.sh on the same
    workload and launch shape — `workload`, `batches_per_call` and the projected-input mode must match — corrected with
    the factors calibrated there).  bench.py cannot collect PMC counters on itself, so this is a measured constant of
    the committed build, refreshed whenever the profile is; None when no matching summary is committed."""
    import glob
    if kernel_id not in PMC_KERNELS:
        return None, None
    doc = src = None
    live = _LIVE_PMC.get((workload, batches_per_call, bool(projected)))
    if live is not None:  # counters collected by this very run take precedence over any committed summary
        doc, src = live, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (collect_live_pmc)"
    for f in ([] if doc is not None else sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc*.json")), reverse=True)):
        try:
            cand = json.load(open(f))
        except (OSError, ValueError):
            continue
        if not isinstance(cand, dict) or "kernels" not in cand:
            continue
        if cand.get("workload", "products") == workload and cand.get("batches_per_call") == batches_per_call and \
                bool(cand.get("projected_input")) == bool(projected):
            doc, src = cand, os.path.basename(f)
            break
    if doc is None:
        return None, None
    tot_bytes = tot_calls = 0.0
    for name, e in doc["kernels"].items():
        # (rocprofv3 leaves a name with a _Float16 parameter mangled: "_ZN12_GLOBAL__N_120linear_fused2_kernelEPKf...")
        if not any(name.startswith(pfx) or (name.startswith("_Z") and pfx in name) for pfx in PMC_KERNELS[kernel_id]):
            continue
        calls = e.get("FETCH_SIZE_calls", 0)
        tot_bytes += e["hbm_bytes_per_launch"] * calls
        # a union group is several kernels launched once per call each; the others are one kernel launched repeatedly
        tot_calls = max(tot_calls, calls) if kernel_id.startswith("union") else tot_calls + calls
    return (tot_bytes / tot_calls if tot_calls else None), src


def rmat_edges_gpu(scale: int, n_edges: int, seed: int, device, a=0.57, b=0.19, c=0.19):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    src = torch.zeros(n_edges, dtype=torch.int64, device=device)
    dst = torch.zeros(n_edges, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(n_edges, generator=g, device=device)
        src = src * 2 + (r >= a + b).to(torch.int64)
        dst = dst * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)
    return src, dst


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: spawn the N ranks here (one process per GPU, the environment
    torchrun would set: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and wait for them; rank 0's JSON
    line goes to this process's stdout.  Fewer devices than ranks is an error (exit code 2) — never a silent 1-GPU
    run — unless GIGL_BENCH_SHARE_GPU=1 (tests: every rank on device 0, gloo collectives, the library's host-callback
    transport instead of RCCL, which refuses two ranks on one device)."""
    import socket
    import subprocess
    n = int(args.gpus)
    have = torch.cuda.device_count()
    share = os.environ.get("GIGL_BENCH_SHARE_GPU") == "1"
    if have < n and not share:
        print(f"bench.py: --gpus {n} needs {n} visible HIP devices, this host has {have}; run on a node with {n} GPUs "
              "(GIGL_BENCH_SHARE_GPU=1 puts every rank on device 0 over gloo — a functional check, not a measurement)",
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share else r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for r, p in enumerate(procs):
        p.wait()
        if p.returncode != 0:
            print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr)
            rc = rc or p.returncode or 1
    return rc


def dist_backend() -> str:
    return "gloo" if os.environ.get("GIGL_BENCH_SHARE_GPU") == "1" else "nccl"


def all_reduce(t: torch.Tensor, op) -> None:
    """dist.all_reduce on a device tensor under either backend (gloo reduces a host copy)"""
    import torch.distributed as dist
    if dist.get_backend() == "gloo" and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


# workload -> (nodes, rmat scale, edges drawn, feature dim, feature dtype, directed, hidden, out, rmat seed, label)
WORKLOADS = {
    # BASELINE.json configs[1] / SURVEY.md §8(d) C2
    "products": (2_449_029, 22, 61_859_140, 100, torch.float32, False, 256, 47, 2, "ogbn-products-shaped RMAT"),
    # the per-GPU share of BASELINE.json configs[2] (MAG240M, SURVEY.md §8(d) C3: N=244,160,499, E=1,728,364,232
    # directed, D=768 fp16, SAGE 768->256->256) held as ONE self-contained graph: 1/8 of the nodes, edges and
    # feature bytes (47 GB) — what each of the 8 shards stores; the full graph needs 375 GB of features
    "mag-shard": (30_520_062, 25, 216_045_529, 768, torch.float16, True, 256, 256, 3,
                  "MAG240M/8-shaped RMAT (one GPU's share of the 8-way sharded graph)"),
    "small": (200_000, 18, 3_000_000, 100, torch.float32, False, 256, 47, 2, "products-shaped-small"),
    # BASELINE.json configs[0] / SURVEY.md §8(d) C1: Cora-shaped (2,708 nodes, 5,278 undirected edges, D=1,433,
    # 7 classes), GraphSAGE 1433->16->7, fanout [10,5] (--fanouts 10,5 --batch 512)
    "cora": (2_708, 12, 5_278, 1_433, torch.float32, False, 16, 7, 1, "Cora-shaped random graph"),
    # the per-GPU share of BASELINE.json configs[3] / C4: RMAT scale-30 (N=2^30, E=1.6e10, D=128 fp16) over 8 GPUs
    # held as one self-contained graph: 2^27 nodes, 2e9 directed edges, 34 GB of features; fanout [15,10], B=4096
    # (--fanouts 15,10 --batch 4096), SAGE 128->256->256
    "rmat-shard": (1 << 27, 27, 2_000_000_000, 128, torch.float16, True, 256, 256, 4,
                   "RMAT scale-30 / 8 (one GPU's share of the 8-way sharded graph)"),
}
WORKLOAD_DEFAULTS = {"cora": ("10,5", 512), "rmat-shard": ("15,10", 4096), "typed-dblp": ("10,5", 4096)}


def cora_c1(seed: int = 1):
    """SURVEY.md 8(d) C1 (BASELINE configs[0]) as host arrays: 2,708 nodes, 5,278 distinct undirected random edges
    (no self loops; 10,556 directed after bidirectionalisation), D = 1,433 fp32 bag-of-words rows ~ Bernoulli(0.0127),
    L1-normalised (a row without a word stays zero), labels uniform over 7 classes -> (n, src, dst, x, labels)"""
    import numpy as np
    n, pairs, d = 2_708, 5_278, 1_433
    rng = np.random.default_rng(seed)
    seen, src, dst = set(), [], []
    while len(src) < pairs:
        a, b = (int(v) for v in rng.integers(0, n, 2))
        key = (min(a, b), max(a, b))
        if a == b or key in seen:
            continue
        seen.add(key)
        src.append(a)
        dst.append(b)
    x = (rng.random((n, d)) < 0.0127).astype(np.float32)
    x /= np.maximum(x.sum(axis=1, keepdims=True), 1.0)
    labels = rng.integers(0, 7, n).astype(np.int64)
    return n, np.array(src, np.int32), np.array(dst, np.int32), x, labels


def build_workload(eng, args):
    dev = eng.device
    name = "small" if getattr(args, "small", False) else getattr(args, "workload", "products")
    n, scale, pairs, d, dtype, directed, hid, out_dim, seed, label = WORKLOADS[name]
    perm_mul = 0x9E3779B1
    if name == "cora":  # (uniform random pairs: Cora is not power-law; bag-of-words rows)
        _, src_h, dst_h, x_h, _ = cora_c1(seed)
        eng.build_from_coo(n, torch.from_numpy(src_h).to(dev), torch.from_numpy(dst_h).to(dev), is_directed=directed)
        eng.load_features(torch.from_numpy(x_h).to(dev))
        args._workload = (name, label, hid, out_dim, directed, dtype)
        return n, d
    # fold the 2^scale id space onto [0, n) and scatter ids so hubs are not the low ids; drawn in chunks (the
    # int64 temporaries of 2e9 edges would not leave room for the sort)
    parts, chunk = [], 1 << 28
    for ci, c0 in enumerate(range(0, pairs, chunk)):
        a_, b_ = rmat_edges_gpu(scale, min(chunk, pairs - c0), seed=seed + 7919 * ci, device=dev)
        parts.append((((a_ * perm_mul) % n).to(torch.int32), ((b_ * perm_mul) % n).to(torch.int32)))
        del a_, b_
    src = torch.cat([q[0] for q in parts]) if len(parts) > 1 else parts[0][0]
    dst = torch.cat([q[1] for q in parts]) if len(parts) > 1 else parts[0][1]
    del parts
    eng.build_from_coo(n, src, dst, is_directed=directed)
    del src, dst
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    x = torch.empty((n, d), device=dev, dtype=dtype)
    step = max(1, (1 << 28) // d)  # generate in <= 1 GiB fp32 pieces (the fp16 table alone is 47 GB for mag-shard)
    for i in range(0, n, step):
        x[i:i + step] = torch.randn((min(step, n - i), d), generator=g, device=dev, dtype=torch.float32).to(dtype)
    eng.load_features(x)
    del x
    torch.cuda.empty_cache()
    args._workload = (name, label, hid, out_dim, directed, dtype)
    return n, d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=960,
                    help="MINIMUM number of timed steps; the timed range is rounded up to whole rounds and repeated "
                         "until --min-seconds (the regime — streams x batches per call — does not depend on it)")
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--min-seconds", type=float, default=2.5, help="lower bound of the timed region")
    ap.add_argument("--min-rounds", type=int, default=10,
                    help="rounds (streams x batches-per-call steps) per timed repetition, at least")
    ap.add_argument("--min-reps", type=int, default=7, help="timed repetitions, at least (median / p10 / p90)")
    ap.add_argument("--batch", type=int, default=0, help="roots per batch (0: the workload's: 1024)")
    ap.add_argument("--fanouts", type=str, default="", help="per-hop fanouts (empty: the workload's: 25,10)")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--group", type=int, default=64,
                    help="batches per library call: G independent batches of B roots share one set of launches "
                         "(each keeps its own union graph; results are bit-identical to G single-batch calls)")
    ap.add_argument("--workload", type=str, default="products",
                    choices=["products", "mag-shard", "mag240m-sharded", "cora", "rmat-shard", "gat-lp", "typed-dblp"],
                    help="products = BASELINE configs[1] (default, the N=1 workload; N>1: a replica per GPU); mag-shard = "
                         "one GPU's 1/8 share of the MAG240M-shaped graph as a self-contained graph (D=768 fp16, SAGE "
                         "768->256->256); mag240m-sharded = BASELINE configs[2]: the MAG240M-shaped graph hash-"
                         "partitioned over the ranks (owner = id %% world), per-hop all_to_all frontier exchange and "
                         "feature pull over RCCL — needs >= 2 GPUs at full size (--shard-scale shrinks it)")
    ap.add_argument("--shard-group", type=int, default=None,
                    help="mag240m-sharded: batches of B roots exchanged per set of collectives (dedup stays per batch); "
                         "default 32, 16 under --emulate-world (eight ranks' workspaces share one GPU's HBM)")
    ap.add_argument("--shard-hot-frac", type=float, default=-1.0,
                    help="mag240m-sharded: fraction of the nodes (the most-referenced ones) whose feature rows are "
                         "replicated on every rank and never pulled (hub-row replication); -1 (default) = auto: on "
                         "whenever world > 1, sized to 4 %% of the free HBM, at most 5 %% of the nodes")
    ap.add_argument("--shard-encoder", type=str, default="sage", choices=["sage", "gat"],
                    help="mag240m-sharded: sage = GraphSAGE 768->256->256 through gigl_dist_plan (dense pull bookkeeping, "
                         "hot rows); gat = BASELINE configs[4]'s encoder, 2-layer GAT heads 2 hid 128 out 128, through "
                         "gigl_dist_gat_plan (raw rows, generic union)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="mag240m-sharded on ONE GPU: all W ranks of the hash-partitioned job as ctxs of this process "
                         "(in-process transport): per-rank pulled rows / bytes / bucket fill / hub-row hit rate / compute "
                         "time are measured, the W-GPU step is projected from them (labelled so)")
    ap.add_argument("--no-emulated-sub", action="store_true",
                    help="N=1 headline: skip the `sharded_emulated` sub-record (an 8-rank emulated world at a reduced scale, "
                         "run in a child process)")
    ap.add_argument("--no-sharded-sub", action="store_true",
                    help="N > 1 headline: skip the `sharded` sub-record (the mag240m-sharded workload at this N)")
    ap.add_argument("--shard-plans", type=int, default=3, help="mag240m-sharded: sharded plans in flight per rank")
    ap.add_argument("--shard-scale", type=float, default=0.0,
                    help="mag240m-sharded: fraction of MAG240M's nodes and edges to generate (0 = world/8, capped at 1: "
                         "every GPU holds the share it has in the 8-GPU job; 1.0 needs 8 GPUs' HBM)")
    ap.add_argument("--project-on-owner", action="store_true",
                    help="mag240m-sharded: owners apply the first layer's weights before sending (256 fp32 per row "
                         "instead of 768 fp16)")
    ap.add_argument("--small", action="store_true", help="200k-node graph (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="counter-collection runs (scripts/gpu_pmc.sh): only warm-up + the timed region, so every "
                         "library launch in the trace is a grouped launch; prints timing without edge counts")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N=1 headline: skip the rocprofv3 counter passes that measure this run's HBM traffic per kernel "
                         "(roofline.traffic then comes from the newest committed profiles/*_pmc*.json of the workload, or is "
                         "null); the passes run as child processes after the timed region, ~1.5 min")
    ap.add_argument("--mode", type=str, default="parity", choices=["parity", "fast"])
    ap.add_argument("--project-input", type=str, default="auto", choices=["auto", "on", "off"],
                    help="first layer over PROJECTED rows (X W_l^T / X W_r^T computed once over the resident table, "
                         "gigl_sage_plan_set_projected_input): auto = when projected rows are narrower than stored rows "
                         "(mag-shard: 768 fp16 -> 256 fp32; not products).  The precompute is timed and charged to "
                         "every step as 1 / (steps of a full inference pass = N / B) of its duration")
    ap.add_argument("--train", action="store_true",
                    help="training step instead of the inference step: a batch sampled in HBM (sample + union graph), "
                         "GraphSAGE forward with autograd over the union graph, cross-entropy on the roots, backward "
                         "(gigl_gather_reduce_backward + the projections' backward GEMMs) and the Adam update — the loop of "
                         "NodeClassificationModelingTaskSpec._train; a secondary line with its own roofline / cpu_baseline")
    ap.add_argument("--gat-train-autograd", action="store_true",
                    help="--workload gat-lp --train: the autograd-driven step (round 4's line) instead of the library plan "
                         "(gigl_gat_nablp_train_plan_*)")
    ap.add_argument("--no-train-prefetch", action="store_true",
                    help="--train --train-task lp: every step samples its own batch (A/B of the next batch's graph part beside "
                         "this step's layers)")
    ap.add_argument("--train-task", type=str, default="snc", choices=["snc", "lp"],
                    help="--train: snc = node classification (gigl_sage_train_plan_*); lp = the link-prediction step of the "
                         "reference's default trainer (GraphSAGE encoder, Retrieval task) as ONE library call "
                         "(gigl_nablp_train_plan_*): --batch anchors (default 2048) with one positive each + 512 random "
                         "negatives per step")
    ap.add_argument("--entry", type=str, default="plan", choices=["plan", "inferencer", "sampler"],
                    help="plan = the library's one-call plan driven by this script (the headline); inferencer = the same "
                         "workload through the drop-in entry point's own loop (gigl_amd.inferencer.Inferencer."
                         "infer_resident -> plugin.infer_batch -> in-HBM route -> Avro shards): one step = one batch of "
                         "the full inference pass over every node; sampler = the Subgraph Sampler job's step (S3-S9): "
                         "k-hop sample of a batch of roots + its RootedNodeNeighborhood TFRecords encoded on the device "
                         "(gigl_sample_khop + gigl_records_encode), records left in HBM; --batch defaults to the job's "
                         "4096 roots")
    ap.add_argument("--entry-sink", type=str, default="avro-device", choices=["avro-device", "avro-files", "none"],
                    help="--entry inferencer: avro-device (the line's value) = rows encoded as Avro data blocks on the "
                         "device, the blocks stay in HBM (outputs resident, like the inputs); avro-files = additionally "
                         "copied out and appended to shard files in a tmpfs scratch directory by the exporter's writer "
                         "thread (PCIe + file inclusive; always measured and reported next to the value); none = bare rows")
    args = ap.parse_args()
    if args.shard_group is None:
        args.shard_group = 16 if getattr(args, "emulate_world", 0) and args.emulate_world > 1 else 32
    wl_fan, wl_b = WORKLOAD_DEFAULTS.get(args.workload, ("25,10", 1024))
    args.fanouts = args.fanouts or wl_fan
    args.batch = args.batch or (4096 if args.entry == "sampler" else wl_b)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # no launcher: this process becomes one
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GIGL_BENCH_SHARE_GPU") == "1":
        local_rank = 0  # (functional check on a one-GPU box, also under a launcher that numbers the ranks' devices)
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node equal to --gpus "
              "(or without a launcher: bench.py spawns the ranks itself)", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if dist_backend() == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    if args.train and args.workload == "gat-lp":
        if args.gat_train_autograd:
            return run_gat_lp_train(args, rank, world, local_rank)
        return run_gat_lp_train_plan(args, rank, world, local_rank)
    if args.train and args.train_task == "lp":
        return run_lp_train(args, rank, world, local_rank)
    if args.train:
        return run_train(args, rank, world, local_rank)
    if args.entry == "inferencer":
        return run_entry_inferencer(args, rank, world, local_rank)
    if args.entry == "sampler":
        return run_entry_sampler(args, rank, world, local_rank)
    if args.workload == "mag240m-sharded" and args.emulate_world > 1:
        if world != 1:
            print("bench.py: --emulate-world runs in one process on one GPU", file=sys.stderr)
            sys.exit(2)
        return run_emulated_world(args, local_rank)
    if args.workload == "mag240m-sharded":
        return run_sharded(args, rank, world, local_rank)
    if args.workload == "gat-lp":
        return run_gat_lp(args, rank, world, local_rank)
    if args.workload == "typed-dblp":
        return run_typed(args, rank, world, local_rank)

    from gigl_amd._lib import KERNEL_IDS, MODE_FAST, MODE_SPARK_HASH, STATS, STATS_LEN
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE

    torch.cuda.set_device(local_rank)
    eng0 = HipEngine(local_rank)
    dev = eng0.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    B, K, W = args.batch, max(1, args.steps), max(0, args.warmup)
    # the execution regime is fixed — S streams x G batches per library call — whatever --steps asks for: --steps is
    # the MINIMUM number of timed steps; the timed range is a whole number of rounds (S*G steps) and is repeated until
    # the timed region lasts >= --min-seconds (>= 200 batches or >= 5 s, median and p10/p90)
    S, G = max(1, args.streams), max(1, args.group)
    rnd = S * G
    K_rep = max(-(-K // rnd), args.min_rounds) * rnd
    L = len(fanouts)
    mode = MODE_SPARK_HASH if args.mode == "parity" else MODE_FAST

    t0 = time.time()
    n, d = build_workload(eng0, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    esz = 4 if wl_dtype == torch.float32 else 2  # bytes per feature element in the resident table
    torch.manual_seed(0)
    model = GraphSAGE(d, hid, out_dim, num_layers=L).to(dev)
    # roots: seeded permutation of node ids (seed 42); rank r takes batches r, r+world, ...
    # pool = warm-up batches + N_SEG segments of K_rep batches; repetition r of the timed range takes segment r % N_SEG
    N_SEG = 2
    Wp = -(-max(W, 1) // rnd) * rnd  # warm-up steps actually run: whole rounds >= --warmup
    pool = Wp + N_SEG * K_rep
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    total_batches = pool * world
    perm = torch.randperm(n, generator=gp)
    if perm.numel() < total_batches * B:
        perm = perm.repeat((total_batches * B + perm.numel() - 1) // perm.numel())
    my = perm[: total_batches * B].view(total_batches, B)[rank::world].to(torch.int32).to(dev).contiguous()
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    # S pipelines: ctx + stream + plan + host thread each, all sampling the same resident graph
    engines, streams, plans, outs = [eng0], [], [], []
    for s in range(1, S):
        e = HipEngine(local_rank)
        e.share_resident(eng0)
        engines.append(e)
    for s in range(S):
        st = torch.cuda.Stream(device=dev)
        engines[s].bind_stream(st)
        streams.append(st)
        plans.append(model.make_plan(engines[s], B, fanouts, groups=G))
        if not args.no_graph:
            plans[s].use_graph(True)  # the call's launches replayed as one hipGraph launch
        outs.append(torch.empty((G * B, out_dim), dtype=torch.float32, device=dev))
    # projected input: the first layer's projection of the WHOLE table, once (timed: charged to the steps below)
    projected = args.project_input == "on" or (args.project_input == "auto" and model.projected_input_pays(eng0))
    pre_s, proj_tables = 0.0, None
    if projected:
        # (the table is allocated once per job — setup, like the feature table itself; what recurs per model state and
        # is charged to the steps is the projection that fills it)
        proj_tables = torch.empty((n, 2 * hid), dtype=torch.float32, device=dev)
        with torch.cuda.stream(streams[0]):
            proj_tables.zero_()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        eng0.project_features(model.conv_layers[0].fused_weight(), out=proj_tables)
        torch.cuda.synchronize()
        pre_s = time.perf_counter() - tp
        for p_ in plans:
            p_.set_projected_input(proj_tables)
    steps_per_pass = -(-n // B)  # steps of one inference pass over every node: what the precompute is amortised over
    pre_per_step_s = pre_s / steps_per_pass

    def run_range(lo, hi, S=S):
        """steps (= batches of B roots) lo..hi-1, a whole number of calls: call c takes the G consecutive batches
        lo+c*G.. on pipeline c % S (one host thread per pipeline)"""
        n_calls = (hi - lo) // G
        assert n_calls * G == hi - lo

        def worker(s):
            for c in range(s, n_calls, S):
                i = lo + c * G
                plans[s].run(my[i:i + G].view(-1), out=outs[s], mode=mode)
        if S == 1:
            return worker(0)
        ths = [threading.Thread(target=worker, args=(s,)) for s in range(S)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    def seg_range(r):
        lo = Wp + (r % N_SEG) * K_rep
        return lo, lo + K_rep

    names = list(KERNEL_IDS)
    if args.timed_only:  # counter-collection runs: warm-up + one timed repetition, grouped launches only
        run_range(0, Wp)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_range(*seg_range(0))
        torch.cuda.synchronize()
        print(json.dumps({"timed_only": True, "steps": K_rep, "batches_per_call": G, "streams": S,
                          "workload": wl_name, "projected_input": bool(projected),
                          "ms_per_step": (time.perf_counter() - t1) / K_rep * 1e3}))
        for e in reversed(engines):
            e.close()
        return
    # ---- untimed: warm-up, then every kernel group's own duration with all event timers on, on ONE stream (with S
    # streams an event interval includes time shared with the other streams' kernels)
    run_range(0, Wp)
    torch.cuda.synchronize()
    P = 2 * rnd
    for e in engines:
        e.profile_enable(names, capacity=(P // G + 4) * 24)
    plo = Wp
    run_range(plo, plo + G, S=1)  # (graph mode: the first call after a mask change is the eager pass of the re-capture)
    for e in engines:
        e.profile_reset()
    run_range(plo, plo + P, S=1)
    for p in plans:
        p.flush_profile()
    prof = {k: [sum(x) for x in zip(*[e.profile_read(k) for e in engines])] for k in names}
    probe_acc = torch.zeros(STATS_LEN, dtype=torch.int64, device=dev)
    with torch.cuda.stream(streams[0]):
        for c in range(P // G):
            r_ = my[plo + c * G: plo + (c + 1) * G].view(-1)
            plans[0].run(r_, out=outs[0], mode=mode)
            plans[0].stats(r_, probe_acc)
    streams[0].synchronize()
    dominant = max(prof, key=lambda k: prof[k][0])
    for e in engines:
        e.profile_enable([], 0)

    # ---- untimed: exact edge counts and algorithmic bytes of every batch of the pool segments, counted on the device
    # (gigl_sage_plan_stats; sampling is deterministic, so these are the timed batches' counts)
    seg_acc = torch.zeros((N_SEG, STATS_LEN), dtype=torch.int64, device=dev)
    with torch.cuda.stream(streams[0]):
        for sg in range(N_SEG):
            lo, hi = seg_range(sg)
            for i in range(lo, hi, G):
                r_ = my[i:i + G].view(-1)
                plans[0].run(r_, out=outs[0], mode=mode)
                plans[0].stats(r_, seg_acc[sg])
    streams[0].synchronize()
    seg_stats = seg_acc.cpu().numpy().astype(np.float64)
    if seg_stats[:, STATS["overflow"]].any() or int(probe_acc[STATS["overflow"]].item()):
        raise RuntimeError("union dedup / workspace overflow in a benchmark batch (meta[GIGL_META_OVERFLOW])")

    # ---- untimed: the same timers under the TIMED regime (S streams, G batches per call, launches eager so the events
    # bracket them) — with the other streams' kernels resident a launch lasts longer than alone, and not by the same
    # factor for every kernel: both figures are reported for every group (roofline.groups)
    prof_alone = prof
    if S > 1:
        for e in engines:
            e.profile_enable(names, capacity=(P // G + 4) * 24)
        run_range(plo, plo + rnd)
        for e in engines:
            e.profile_reset()
        run_range(plo, plo + P)
        for p in plans:
            p.flush_profile()
        prof_ovl = {k: [sum(x) for x in zip(*[e.profile_read(k) for e in engines])] for k in names}
        for e in engines:
            e.profile_enable([], 0)
    else:
        prof_ovl = prof
    # the dominant group = the one with the largest duration of its OWN (single-stream probe): a stable ranking — under
    # overlap two near-equal groups trade places from run to run; every group's overlapped figure is in roofline.groups
    dominant = max(prof, key=lambda k: prof[k][0])

    # ---- calibration repetition (untimed; also re-captures every plan's hipGraph under the final timer mask)
    for e in engines:
        e.profile_enable([dominant], capacity=64)
    run_range(0, rnd)
    torch.cuda.synchronize()
    tc = time.perf_counter()
    run_range(*seg_range(0))
    torch.cuda.synchronize()
    t_cal = time.perf_counter() - tc
    reps = int(min(max(np.ceil(args.min_seconds / max(t_cal, 1e-6)), args.min_reps), 2000))
    if world > 1:
        rr = torch.tensor([reps], dtype=torch.int64, device=dev)
        all_reduce(rr, dist.ReduceOp.MAX)
        reps = int(rr.item())
    calls_per_rep = K_rep // G
    for e in engines:
        e.profile_enable([dominant], capacity=(reps * (calls_per_rep // S + 2) + 8) * 4)
    run_range(0, rnd)  # re-capture after the capacity change
    torch.cuda.synchronize()
    for e in engines:
        e.profile_reset()

    # ---- timed region: `reps` repetitions of K_rep steps, each bracketed by barrier + synchronize on both sides
    rep_s = []
    for r in range(reps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_range(*seg_range(r))
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t1)
    if world > 1:
        dist.barrier()
    for p in plans:
        p.flush_profile()
    dom_ms, dom_launches = [sum(x) for x in zip(*[e.profile_read(dominant) for e in engines])]
    for e in engines:
        e.profile_enable([], 0)

    # ---- reduce over ranks: a repetition lasts as long as its slowest rank
    rep_t = torch.tensor(rep_s, dtype=torch.float64, device=dev)
    seg_use = np.array([sum(1 for r in range(reps) if r % N_SEG == sg) for sg in range(N_SEG)], dtype=np.float64)
    tot = (seg_stats * seg_use[:, None]).sum(0)  # this rank's counts over the whole timed region
    if world > 1:
        all_reduce(rep_t, dist.ReduceOp.MAX)
        cc = torch.tensor(tot, dtype=torch.float64, device=dev)
        all_reduce(cc, dist.ReduceOp.SUM)
        tot_all = cc.cpu().numpy()
    else:
        tot_all = tot
    rep_np = rep_t.cpu().numpy() + K_rep * pre_per_step_s  # (+ every step's share of the table projection, if any)
    elapsed = float(rep_np.sum())
    steps_total = reps * K_rep
    sampled_all, aggregated_all = float(tot_all[STATS["sampled"]]), float(tot_all[STATS["aggregated"]])
    ref_equiv_all = float(L * tot_all[STATS["union_edges"]])
    value = (sampled_all + aggregated_all) / elapsed
    ms_rep = rep_np / K_rep * 1e3  # ms per step of every repetition
    # edges of repetition r (all ranks ~ world x this rank's) -> per-repetition throughput spread
    per_rep_edges = np.array([seg_stats[r % N_SEG, STATS["sampled"]] + seg_stats[r % N_SEG, STATS["aggregated"]]
                              for r in range(reps)]) * (tot_all[STATS["sampled"]] + tot_all[STATS["aggregated"]]) / \
        max(tot[STATS["sampled"]] + tot[STATS["aggregated"]], 1.0)
    rate_rep = per_rep_edges / rep_np

    # ---- algorithmic bytes / flops from the exact counts
    dims = [d] + [hid] * (L - 1)
    half_split = (not projected) and hasattr(plans[0], "half_split") and plans[0].half_split()
    # both projections in one kernel (gigl_sage_plan_fused_layers): 2 x 96 floats of [W_l h | W_r h] (two K-split planes)
    # leave the first projection per row instead of the hidden row, the last layer is one reduction over them
    fused_layers = (not projected) and hasattr(plans[0], "fused_layers") and plans[0].fused_layers()

    def alg_of(st):
        """st: a STATS vector -> (bytes per kernel group, projection flops)"""
        ab = {k: 0.0 for k in names}
        fl = 0.0  # (fp32-equivalent flops; alg_of.issued = the 16-bit MFMA flops they take)
        alg_of.issued = 0.0
        for l in range(L):
            agg_l, rows_l = st[STATS["agg_layer0"] + l], st[STATS["rows_layer0"] + l]
            s_in = esz if l == 0 else 4  # layer 0 gathers rows of the resident table, later layers fp32 activations
            dout = hid if l < L - 1 else out_dim
            if l == 0 and projected:  # fp32 rows of W_l x per edge, the W_r x row of the destination, the output row
                ab["gather_mean"] += agg_l * (4 + dout * 4) + rows_l * (8 + 2 * dout * 4)
                continue
            if fused_layers and l == 1:
                # the last layer over p rows: per edge the W_l half (48 floats) of both planes, per root the W_r half of
                # both planes + the output row; no projection
                ab["gather_mean"] += agg_l * (4 + 2 * 48 * 4) + rows_l * (8 + 2 * 48 * 4 + out_dim * 4)
                continue
            if fused_layers and l == 0:
                two_src = True
                ab["gather_mean"] += agg_l * (4 + dims[l] * s_in) + rows_l * (8 + dims[l] * 4)
                # operand rows in, two planes of 96 floats out; + the second product's flops (256 -> 96, three products)
                ab["linear"] += rows_l * (2 * dims[l] + 2 * 96) * 4 + dout * 2 * dims[l] * 4 + 96 * dout * 4
                fl += 2.0 * rows_l * (2 * dims[l] * dout + dout * 96)
                alg_of.issued += 2.0 * rows_l * (2 * dims[l] * dout + dout * 96) * 3
                continue
            #  gather layer l: E_l*(4 + D_l*s) + N_dst*(8 + D_l*s_out); the self half of the projection's operand is
            #  read by the projection itself from the fp32 source rows (two-source operand) — or, for an fp16 table's
            #  first layer, copied alongside by the gather (D_l*s read + D_l*4 written)
            two_src = (l > 0 or esz == 4 or half_split) and all(v % 4 == 0 for v in dims) and not os.environ.get("GIGL_PLAN_SELF_COPY")
            ab["gather_mean"] += agg_l * (4 + dims[l] * s_in) + rows_l * (8 + dims[l] * 4) + \
                (0 if two_src else rows_l * (dims[l] * s_in + dims[l] * 4))
            ab["linear"] += rows_l * (2 * dims[l] + dout) * 4 + dout * 2 * dims[l] * 4
            fl += 2.0 * rows_l * 2 * dims[l] * dout
            alg_of.issued += 2.0 * rows_l * 2 * dims[l] * dout * (3 if (l == 0 and half_split) else 6)
        #  union: 16 B per sampled edge + 4 B per unique node, attributed evenly to its phases
        for k in ("union_insert", "union_relax", "union_nodes", "union_edge_sort", "union_csr"):
            ab[k] = (16 * st[STATS["sampled"]] + 4 * st[STATS["union_nodes"]]) / 4.0
        # parity mode, what a position-keyed sampler must move per frontier node (gigl_sage_plan_stats): 16 + (a row of
        # <= f neighbours: 4 deg, else the <= lambda threshold-list pairs 8 lambda + the f chosen ids 4 f) + 8 min(deg, f)
        ab["expand"] = st[STATS["expand_bytes"]]
        return ab, fl

    alg_timed, _ = alg_of(tot)  # this rank's timed region (the event timers are this rank's too)
    alg_probe, flops_probe = alg_of(probe_acc.cpu().numpy().astype(np.float64))
    issued_probe = alg_of.issued
    avg_launch_ms = dom_ms / max(dom_launches, 1)
    bytes_per_launch = alg_timed[dominant] / max(dom_launches, 1)
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    live_pmc_note = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)  # (no nested rocprofv3 runs)
    if rank == 0 and world == 1 and not args.no_live_pmc and not args.timed_only and not under_profiler and \
            not os.environ.get("GIGL_BENCH_CHILD"):
        passthrough = ["--workload", args.workload, "--batch", str(B), "--fanouts", ",".join(str(f) for f in fanouts),
                       "--group", str(G), "--mode", args.mode, "--project-input", args.project_input] + \
            (["--small"] if args.small else [])
        torch.cuda.synchronize()
        doc_, live_pmc_note = collect_live_pmc(passthrough)
        if doc_ is not None:
            _LIVE_PMC[(wl_name, G, bool(projected))] = doc_
    traffic, traffic_src = pmc_traffic(dominant, G, wl_name, projected)
    # every kernel group against its own bound, from the single-stream probe (P steps, all timers on)
    by_kernel = {}
    for k, v in prof.items():
        ms_step = v[0] / P
        if ms_step <= 0:
            continue
        if k == "linear":
            tf = flops_probe / P / (ms_step * 1e-3) / 1e12
            tf16 = issued_probe / P / (ms_step * 1e-3) / 1e12
            by_kernel[k] = {"bound": "mfma", "achieved": round(tf16, 2), "peak": MFMA_16BIT_PEAK_TF,
                            "unit": "TFLOP/s of 16-bit MFMA products issued (6 bf16 products per fp32 product; 3 fp16 "
                                    "products in a half-split first layer)",
                            "frac": round(tf16 / MFMA_16BIT_PEAK_TF, 4), "ms_per_step": round(ms_step, 5),
                            "fp32_equivalent_tflops": round(tf, 2), "half_split_first_layer": bool(half_split),
                            "vs_native_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 4)}
            # operand rows in, output rows out (rows * (2 d + d_out) * 4 + the weights): with three products per fp32
            # product the K = 2 d projection of narrow rows moves its bytes faster than it fills the matrix pipe —
            # the binding roofline is whichever fraction is larger
            gbs = alg_probe[k] / P / (ms_step * 1e-3) / 1e9
            if gbs / HBM_PEAK_GBS > by_kernel[k]["frac"]:
                by_kernel[k] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_step": round(ms_step, 5),
                                "mfma": {kk: vv for kk, vv in by_kernel[k].items() if kk != "ms_per_step"}}
        else:
            gbs = alg_probe[k] / P / (ms_step * 1e-3) / 1e9
            by_kernel[k] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_step": round(ms_step, 5)}
            tk, _src = pmc_traffic(k, G, wl_name, projected)
            if tk is not None and v[1] > 0:  # counter traffic per launch / the kernel's own (single-stream) duration
                by_kernel[k]["traffic_frac"] = round(tk / (v[0] / v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if by_kernel[k]["frac"] > 1.0:  # the byte model counts bytes the kernel does not move (the sampler reads a
                # few % of 4*deg per row): never a fraction above 1 — the measured traffic, or none
                by_kernel[k]["algorithmic_frac"] = by_kernel[k]["frac"]
                by_kernel[k]["frac"] = by_kernel[k].get("traffic_frac")
    if dominant == "linear":  # the dense projection is the one MFMA-bound kernel
        _, fl_t = alg_of(tot)
        tf = fl_t / max(dom_launches, 1) / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        tf16 = alg_of.issued / max(dom_launches, 1) / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        head = {"bound": "mfma", "kernel": dominant, "achieved": round(tf16, 2), "peak": MFMA_16BIT_PEAK_TF,
                "unit": "TFLOP/s", "frac": round(tf16 / MFMA_16BIT_PEAK_TF, 5),
                "fp32_equivalent_tflops": round(tf, 2), "half_split_first_layer": bool(half_split),
                "vs_native_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 5)}
        if achieved / HBM_PEAK_GBS > head["frac"]:  # (see by_kernel: the projection's bytes bind before its MFMAs)
            head = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "mfma": head}
        note = None
    else:
        alg_frac = achieved / HBM_PEAK_GBS
        head = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(alg_frac, 5)}
        note = None
        if alg_frac > 1.0 or "algorithmic_frac" in by_kernel.get(dominant, {}):  # (overlapped or on its own stream)
            # The kernel does not move the contract's algorithmic bytes (parity sampling never reads the adjacency row:
            # it selects positions from the precomputed table of the hash sequence and fetches only the f chosen
            # ids), so bytes/duration is not a bandwidth.  The headline is then the MEASURED fabric traffic per launch
            # (rocprofv3 PMC, profiles/) over the live launch duration; the contract figure stays in `algorithmic`.
            head["algorithmic"] = {"achieved": round(achieved, 2), "frac": round(alg_frac, 5)}
            if traffic is not None and avg_launch_ms > 0:
                t_gbs = traffic / (avg_launch_ms * 1e-3) / 1e9
                head.update({"achieved": round(t_gbs, 2), "frac": round(t_gbs / HBM_PEAK_GBS, 5)})
                note = ("dominant kernel moves fewer bytes than SURVEY.md 8(d) counts for it (algorithmic frac > 1): "
                        "achieved/frac = PMC fabric traffic per launch / live launch duration; the kernel is "
                        "instruction-bound, not HBM-bound; `algorithmic` holds the contract figure")
            else:  # no counter summary for this launch shape: headline the slowest group whose byte model holds
                cand = {k: v for k, v in by_kernel.items()
                        if v["bound"] == "hbm" and "algorithmic_frac" not in v and v["frac"] is not None}
                k2 = max(cand, key=lambda k: cand[k]["ms_per_step"])
                head.update({"kernel": k2, "achieved": cand[k2]["achieved"], "frac": cand[k2]["frac"]})
                note = (f"dominant kernel `{dominant}` has algorithmic frac > 1 and no PMC summary for this launch "
                        f"shape is committed: headline = `{k2}`, the slowest HBM-bound group (single-stream probe)")
    head["frac_overlapped"] = head["frac"]  # the kernel while the other streams' kernels share the GPU (timed region)
    head["frac_alone"] = by_kernel.get(head["kernel"], {}).get("frac")  # ... and on its own (single-stream probe)
    # every group, alone and overlapped, against the bytes it must move and the bytes it did move (counters)
    groups, step_alg, step_traffic, traffic_complete = {}, 0.0, 0.0, True
    for k in names:
        if prof[k][0] <= 0:
            continue
        alone_ms, ovl_ms = prof[k][0] / P, prof_ovl[k][0] / P
        if k == dominant and dom_ms > 0:
            # ONE overlapped figure per group: the dominant group's comes from the timed region itself (its HIP-event
            # timer stays on there), the others' from the untimed probe of the same regime
            ovl_ms = dom_ms / steps_total
        ab = alg_probe[k] / P
        tk, _ = pmc_traffic(k, G, wl_name, projected)
        tb = None  # counter bytes per step
        if tk is not None and k.startswith("union"):
            tb = tk / G  # (a union group's kernels run once per call each: pmc_traffic returns the group's bytes per call)
        elif tk is not None:
            tb = tk * (prof[k][1] / max(P // G, 1)) / G  # bytes per launch x launches per call / batches per call
        step_alg += ab
        if tb is None:
            traffic_complete = False
        else:
            step_traffic += tb
        fr = lambda byts, ms: None if byts is None or ms <= 0 else round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        groups[k] = {"ms_per_step_alone": round(alone_ms, 5), "ms_per_step_overlapped": round(ovl_ms, 5),
                     "alg_bytes_per_step": round(ab), "frac_alone": fr(ab, alone_ms), "frac_overlapped": fr(ab, ovl_ms),
                     "traffic_bytes_per_step": None if tb is None else round(tb),
                     "traffic_frac_alone": fr(tb, alone_ms), "traffic_frac_overlapped": fr(tb, ovl_ms)}
        if k == "linear":
            groups[k]["mfma_16bit_frac_alone"] = round(issued_probe / P / (alone_ms * 1e-3) / 1e12 / MFMA_16BIT_PEAK_TF, 4)
            groups[k]["mfma_16bit_frac_overlapped"] = round(issued_probe / P / (max(ovl_ms, 1e-9) * 1e-3) / 1e12 / MFMA_16BIT_PEAK_TF, 4)
    step_ms = elapsed / steps_total * 1e3
    step_level = {"ms_per_step": round(step_ms, 5), "alg_bytes_per_step": round(step_alg),
                  "alg_frac": round(step_alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "traffic_bytes_per_step": round(step_traffic) if traffic_complete and step_traffic > 0 else None,
                  "traffic_frac": (round(step_traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                   if traffic_complete and step_traffic > 0 else None),
                  "note": "all kernel groups of a step together: bytes per step / the timed region's ms_per_step / the HBM "
                          "peak — independent of which group is called dominant"}
    roofline = {**head, "groups": groups, "step": step_level, "live_pmc": live_pmc_note or ("collected" if _LIVE_PMC else None),
                "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_src,
                "dominant": dominant,
                "dominant_from": "largest HIP-event time per kernel group on its own (single-stream untimed probe, all "
                                 "timers on): stable from run to run; `groups` lists every group alone and overlapped",
                "avg_launch_us": round(avg_launch_ms * 1e3, 2),
                "alg_bytes_per_launch": round(bytes_per_launch), "launches": int(dom_launches), "note": note,
                "timing": f"HIP events on the kernel's stream over the timed region ({S} streams: intervals include "
                          "overlap with the other streams' kernels); by_kernel: single-stream untimed probe",
                "by_kernel": by_kernel}

    cpu_baseline = cpu_baseline_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (a reported baseline: N=1 only)
        cpu_baseline, cpu_baseline_all = run_cpu_baseline(eng0, model, my, fanouts, Wp, n, d)

    if rank == 0:
        q = lambda a, p: float(np.percentile(a, p))
        line = {
            "metric": "sampled+aggregated edges/s", "value": value, "unit": "edges/s", "n_gpus": world,
            "steps": steps_total, "warmup": Wp, "ms_per_step": elapsed / steps_total * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "steps_requested": K, "warmup_requested": W,
            "timing": {"repetitions": reps, "steps_per_repetition": K_rep, "timed_region_s": round(elapsed, 3),
                       "ms_per_step_median": q(ms_rep, 50), "ms_per_step_p10": q(ms_rep, 10),
                       "ms_per_step_p90": q(ms_rep, 90), "value_median": q(rate_rep, 50),
                       "value_p10": q(rate_rep, 10), "value_p90": q(rate_rep, 90),
                       "protocol": "--steps is a minimum: the timed range is a whole number of rounds (streams x "
                                   "batches_per_call steps, >= --min-rounds) repeated until >= --min-seconds; every "
                                   "repetition is bracketed by barrier + synchronize; value = all edges / sum of the "
                                   "repetitions' max-over-ranks times"},
            "config": {"workload": wl_label +
                       f" N={n} E={eng0.n_edges} {'directed' if wl_directed else 'bidirectionalised'} D={d} "
                       f"{'fp32' if esz == 4 else 'fp16'} features, fanout={fanouts} B={B}/GPU GraphSAGE "
                       f"{d}->{hid}->{out_dim} (fp32 accumulate) inference step (sample+union+forward), sampler mode="
                       + args.mode,
                       "graph": "replica per GPU, roots sharded across ranks",
                       "streams": S, "batches_per_call": G, "fused_layers": bool(fused_layers),
                       "projected_input": (None if not projected else {
                           "precompute_s": round(pre_s, 4), "steps_per_pass": steps_per_pass,
                           "charged_ms_per_step": pre_per_step_s * 1e3,
                           "tflops_fp32_equiv": 2.0 * n * d * 2 * hid / max(pre_s, 1e-9) / 1e12,
                           "table_bytes": int(2 * n * hid * 4),
                           "note": "first layer = one reduction over X W_l^T rows + the destination's X W_r^T row + bias "
                                   "(gigl_sage_plan_set_projected_input); the table projection runs once per model and "
                                   "pass, its time / (N / B) is inside every step's time and inside `value`"}),
                       "sampled_edges_per_step": sampled_all / (steps_total * world),
                       "aggregated_edges_per_step": aggregated_all / (steps_total * world),
                       "reference_equivalent_aggregated_per_step": ref_equiv_all / (steps_total * world),
                       "sampled_edges_per_s": sampled_all / elapsed, "aggregated_edges_per_s": aggregated_all / elapsed,
                       "edge_counts_from": "every timed batch counted on the device (gigl_sage_plan_stats; sampling is "
                                           "deterministic)",
                       "setup_s": round(setup_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baseline_all_cores": cpu_baseline_all,
        }
    for p_ in plans:
        p_.close()
    for e in reversed(engines):
        e.close()
    torch.cuda.empty_cache()
    if world > 1 and wl_name in ("products", "small") and not args.no_sharded_sub:
        # the graph-larger-than-one-GPU path at this N (BASELINE configs[2]): the MAG240M-shaped graph hash-partitioned
        # over the ranks, through the library's sharded plan — a sub-record of the line, never its value
        # The headline above is complete; the sub-record must never cost it — its collectives (RCCL issued by the
        # library) have not run on a real multi-GPU node yet.  Every rank therefore runs it in a CHILD process (the
        # same script as the mag240m-sharded workload, the ranks' own process group on another port): a crash or a
        # hang there ends the child, not the line.  Rank 0 embeds the child's JSON line, or the reason there is none.
        import subprocess
        limit = float(os.environ.get("GIGL_BENCH_SUB_TIMEOUT", "600"))
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--workload", "mag240m-sharded",
               "--fanouts", "25,10", "--batch", "1024", "--min-seconds", str(args.min_seconds), "--min-reps",
               str(args.min_reps), "--min-rounds", str(args.min_rounds), "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--no-cpu-baseline",
               "--shard-group", str(args.shard_group), "--shard-hot-frac", str(args.shard_hot_frac), "--shard-plans",
               str(args.shard_plans), "--shard-scale", str(args.shard_scale), "--project-input", args.project_input,
               "--mode", args.mode] + (["--project-on-owner"] if args.project_on_owner else [])
        env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 17))
        # (under torchrun the parents' rendezvous store belongs to the elastic agent — TORCHELASTIC_USE_AGENT_STORE — and
        # nobody would host one on the children's port: without these variables the children's rank 0 hosts its own)
        for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
            env.pop(k)
        sub_err = None
        try:
            cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=limit)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if rank == 0:
                if cp.returncode == 0 and lines:
                    # N > 1: the line's value / config.workload ARE the north-star workload — BASELINE configs[2], the
                    # MAG240M-shaped graph hash-partitioned over the N ranks (weak scaling: N/8 of the graph, each GPU
                    # holding the share it has in the 8-GPU job), exchanges over RCCL; the replica run above (every rank
                    # a copy of the products-shaped graph, no data-path collective) becomes the sub-record
                    sub = json.loads(lines[-1])
                    rep_roof = {k: v for k, v in (line.get("roofline") or {}).items() if k not in ("groups", "by_kernel")}
                    sub["replicas"] = {**{k: line.get(k) for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup",
                                                                   "timing", "config")}, "roofline": rep_roof}
                    sub["headline_is"] = ("mag240m-sharded (BASELINE configs[2]) at shard scale N/8 over the N ranks' RCCL "
                                          "communicators; `replicas` = the products-shaped replica-per-GPU run of the same "
                                          "launch (no data-path collective)")
                    line = sub
                else:
                    sub_err = f"exit code {cp.returncode}: {cp.stderr.strip()[-400:]}"
            elif cp.returncode != 0:
                sub_err = f"rank {rank}: exit code {cp.returncode}"
        except subprocess.TimeoutExpired:
            sub_err = f"the sharded sub-record did not finish within {limit:.0f} s (GIGL_BENCH_SUB_TIMEOUT)"
        except Exception as ex:  # noqa: BLE001
            sub_err = f"{type(ex).__name__}: {str(ex)[:400]}"
        if rank == 0 and sub_err:
            line["sharded"] = {"error": sub_err}
            line["headline_is"] = ("FALLBACK: the mag240m-sharded run of this launch failed (see `sharded.error`); value / "
                                   "config are the products-shaped replica-per-GPU run, not the hash-partitioned workload")
    if rank == 0 and world == 1 and wl_name == "products" and not args.no_emulated_sub and not args.timed_only and \
            not os.environ.get("GIGL_BENCH_CHILD") and not under_profiler:
        # BASELINE configs[2] on the one GPU the driver's N=1 run has: the 8-rank hash-partitioned job emulated in one
        # process at a reduced scale (run_emulated_world) — measured per-rank bytes / fill / hit rate / compute, and the
        # 8-GPU step projected from them.  A child process with a time limit: it can never cost the headline.
        import subprocess
        limit = float(os.environ.get("GIGL_BENCH_SUB_TIMEOUT", "240"))
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "mag240m-sharded", "--emulate-world", "8",
               "--shard-scale", os.environ.get("GIGL_BENCH_EMULATE_SCALE", "0.08"), "--fanouts", "25,10", "--batch", "1024",
               "--shard-group", "16", "--steps", "256", "--no-cpu-baseline", "--no-live-pmc"]
        try:
            cp = subprocess.run(cmd, env=dict(os.environ, GIGL_BENCH_CHILD="1"), capture_output=True, text=True, timeout=limit)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode == 0 and lines:
                sub = json.loads(lines[-1])
                line["sharded_emulated"] = {k: sub.get(k) for k in ("emulated_world", "value", "value_is", "ms_per_step",
                                                                    "config", "emulated")}
            else:
                line["sharded_emulated"] = {"error": f"exit code {cp.returncode}: {cp.stderr.strip()[-300:]}"}
        except subprocess.TimeoutExpired:
            line["sharded_emulated"] = {"error": f"did not finish within {limit:.0f} s (GIGL_BENCH_SUB_TIMEOUT)"}
        except Exception as ex:  # noqa: BLE001
            line["sharded_emulated"] = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
    if rank == 0:
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_sharded(args, rank, world, local_rank, sub=False):
    """BASELINE.json configs[2]: MAG240M-shaped graph (N=244,160,499, E=1,728,364,232 directed RMAT, D=768 fp16,
    SURVEY.md 8(d) C3) hash-partitioned over the ranks: rank r holds the CSC rows and feature rows of the nodes
    with id % world == r.  A step = one batch of B roots per rank through the library's sharded plan
    (gigl_dist_plan_*, csrc/dist.hip): per hop one all-to-all of (node, K) requests to the owners,
    gigl_expand_frontier there, one all-to-all back; union graph locally; the UNIQUE node ids pulled from their owners
    (rows gathered — or projected by the first layer, --project-on-owner — straight into the send buffer); 2-layer
    GraphSAGE 768->256->256.  Every exchange is issued by the library over RCCL on the plan's stream and nothing in
    a step reads the device from the host.  Several plans (ctx + stream + communicator each) are in flight: one host
    thread issues their phases interleaved, in the same order on every rank, so one plan's exchange overlaps the
    other's expansion / forward.  sub=True: called at the end of the N > 1 headline run for its `sharded` sub-record —
    returns the record (rank 0) instead of printing it and leaves the process group alone."""
    import torch.distributed as dist
    from gigl_amd._lib import STATS, STATS_LEN
    from gigl_amd.dist import Comm, DistSagePlan
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE

    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():  # single rank: RCCL with itself (same code path, the exchange is a device copy)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    L = len(fanouts)
    B, K, W = args.batch, max(1, args.steps), max(0, args.warmup)
    if args.shard_scale <= 0.0:  # weak scaling: a rank's shard is 1/8 of MAG240M whatever the world size
        args.shard_scale = min(1.0, world / 8.0)
    n = max(int(244_160_499 * args.shard_scale), world * 1024)
    e_total = max(int(1_728_364_232 * args.shard_scale), 1)
    d, hid, out_dim = 768, 256, 256
    t0 = time.time()
    # ---- this rank's shard: every rank draws the same seeded edge chunks and keeps the edges it owns
    scale_bits = max(int(np.ceil(np.log2(n))), 10)
    chunk, keys = 1 << 26, []
    for ci, c0 in enumerate(range(0, e_total, chunk)):
        m = min(chunk, e_total - c0)
        src, dst = rmat_edges_gpu(scale_bits, m, seed=3 + 7919 * ci, device=dev)
        src = (src * 0x9E3779B1) % n
        dst = (dst * 0x9E3779B1) % n
        keep = (dst % world) == rank
        keys.append(((dst[keep] // world) << 32) | src[keep])
        del src, dst, keep
    key = torch.unique(torch.cat(keys))  # sorted by (local row, src), duplicates dropped
    del keys
    n_local = (n - rank + world - 1) // world
    rowptr = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(key >> 32, minlength=n_local), 0)
    col = (key & 0xFFFFFFFF).to(torch.int32)
    maxdeg = torch.tensor([int((rowptr[1:] - rowptr[:-1]).max())], dtype=torch.int64, device=dev)
    e_local = torch.tensor([int(col.numel())], dtype=torch.int64, device=dev)
    all_reduce(maxdeg, dist.ReduceOp.MAX)
    all_reduce(e_local, dist.ReduceOp.SUM)
    eng.load_csc(rowptr, col)
    # replicated hot rows (--shard-hot-frac): the nodes that occur most often as in-neighbours, the same set on every rank
    hot_ids = None
    hot_frac = float(args.shard_hot_frac)
    if hot_frac < 0.0:
        # auto (the default): hub-row replication ON whenever rows travel (world > 1) — the fraction of the nodes whose
        # replicated rows fit in 4 % of the HBM still free after the shard is loaded, at most 5 %
        if world == 1:
            hot_frac = 0.0
        else:
            free_b, _ = torch.cuda.mem_get_info(dev)
            free_b -= ((n + world - 1) // world) * d * 2  # (the rank's feature rows are loaded below)
            hot_frac = max(0.0, min(0.05, 0.04 * free_b / max(n * d * 2, 1)))
        fr = torch.tensor([hot_frac], dtype=torch.float64, device=dev)
        all_reduce(fr, dist.ReduceOp.MIN)  # the same set on every rank
        hot_frac = float(fr.item())
    n_hot = int(n * max(0.0, hot_frac))
    if n_hot > 0:
        occ = torch.bincount(col.to(torch.int64) & 0xFFFFFFFF, minlength=n).to(torch.int32)
        all_reduce(occ, dist.ReduceOp.SUM)
        hot_ids = torch.topk(occ.to(torch.int64) * (1 << 32) + (n - 1 - torch.arange(n, device=dev)), n_hot).indices
        hot_ids = hot_ids.to(torch.int32).contiguous()  # (ties broken by id: identical on every rank)
        del occ
    del key, rowptr, col
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x_local = torch.empty((n_local, d), device=dev, dtype=torch.float16)
    step_rows = max(1, (1 << 28) // d)
    for i in range(0, n_local, step_rows):
        x_local[i:i + step_rows] = torch.randn((min(step_rows, n_local - i), d), generator=g, device=dev).to(torch.float16)
    eng.load_features(x_local)
    hot_rows = None
    if hot_ids is not None:  # every rank contributes the rows it owns; the sum over ranks is the replicated table
        hi = hot_ids.to(torch.int64) & 0xFFFFFFFF
        mine_hot = (hi % world) == rank
        hot_rows = torch.zeros((n_hot, d), device=dev, dtype=torch.float16)
        hot_rows[mine_hot] = x_local[hi[mine_hot] // world]
        all_reduce(hot_rows, dist.ReduceOp.SUM)
    del x_local
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    gat = getattr(args, "shard_encoder", "sage") == "gat"
    if gat:
        from gigl_amd.models_attn import GAT
        hid, out_dim = 128, 128
        model = GAT(d, hid, out_dim, num_layers=L, heads=2).to(dev)
        hot_ids = None  # (replicated hot rows belong to the SAGE plan's dense bookkeeping)
        n_hot = 0
    else:
        model = GraphSAGE(d, hid, out_dim, num_layers=L).to(dev)
        w, bs = model.fused_params()
    # pre-projected rows: every rank projects ITS shard once ([W_l x | W_r x]); the pull then moves 1 KB W_l x rows
    # instead of 1.5 KB raw rows and no step projects anything.  Timed, and charged to every step as 1 / (this rank's
    # steps of a full pass over all nodes = N / (B * world)) of its duration
    pre_s, proj_table = 0.0, None
    if not gat and not args.project_on_owner and args.project_input != "off" and L == 2 and \
            (args.project_input == "on" or model.projected_input_pays(eng)):
        proj_table = torch.empty((n_local, 2 * hid), dtype=torch.float32, device=dev)
        proj_table.zero_()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        eng.project_features(w[0], out=proj_table)
        torch.cuda.synchronize()
        pre_s = time.perf_counter() - tp
        if hot_ids is not None:  # the replicas become W_l x rows: every rank contributes the rows it owns
            hi = hot_ids.to(torch.int64) & 0xFFFFFFFF
            mine_hot = (hi % world) == rank
            hot_rows = torch.zeros((n_hot, hid), device=dev, dtype=torch.float32)
            hot_rows[mine_hot] = proj_table[hi[mine_hot] // world, :hid]
            all_reduce(hot_rows, dist.ReduceOp.SUM)
    steps_per_pass = max(1, -(-n // (B * world)))
    pre_per_step_s = pre_s / steps_per_pass
    # every hash window ends below (hops+1)*n + seed*hops + maxdeg: lets the owners use the range table throughout
    bound = (L + 1) * n + 42 * L + int(maxdeg.item())
    mwe = bound if bound < (1 << 30) else -1
    # G consecutive batches travel together: one set of exchanges and launches per G steps; the union graph keeps the
    # batches apart (dedup within a batch only), so a step's edges are those of its batch
    G, S = max(1, args.shard_group), max(1, args.shard_plans)
    rnd = S * G
    K_rep = max(-(-K // rnd), args.min_rounds) * rnd
    Wp = -(-max(W, 1) // rnd) * rnd
    N_SEG = 2
    pool = Wp + N_SEG * K_rep
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    perm = torch.randint(0, n, (pool * world * B,), generator=gp)
    my = perm.view(pool * world, B)[rank::world].to(torch.int32).to(dev).contiguous().view(-1, G * B)  # [calls, G*B]

    class Slot:  # one plan in flight: ctx + stream + communicator + plan
        pass

    def make_slots(pull_cap, pull_cap_b=0):
        slots = []
        for si in range(S):
            sl = Slot()
            sl.stream = torch.cuda.Stream(device=dev)
            sl.eng = eng if si == 0 else HipEngine(local_rank)
            if si:
                sl.eng.share_resident(eng)
            sl.eng.bind_stream(sl.stream)
            sl.comm = Comm.from_torch(sl.eng)  # RCCL (nccl backend); the host-callback transport under gloo
            if gat:
                sl.plan = model.make_dist_plan(sl.comm, G * B, fanouts, group_roots=B, max_window_end=mwe, pull_cap=pull_cap)
            else:
                sl.plan = DistSagePlan(sl.comm, w, bs, G * B, fanouts, group_roots=B,
                                       project_on_owner=args.project_on_owner, pull_cap=pull_cap, max_window_end=mwe,
                                       projected=proj_table, pull_cap_b=pull_cap_b)
            sl.out = sl.plan.new_out()
            if hot_ids is not None:
                sl.plan.set_hot_rows(hot_ids, hot_rows)
            slots.append(sl)
        return slots

    def close_slots(slots):
        for sl in slots:
            sl.plan.close()
            sl.comm.close()
        for sl in reversed(slots[1:]):
            sl.eng.close()

    def run_calls(slots, lo, hi, acc=None):
        """library calls lo..hi-1 (G batches each), S at a time: the phases of the S plans are issued interleaved by
        this one thread — the same order on every rank"""
        lib = slots[0].plan._lib
        for c0 in range(lo, hi, S):
            live = [(slots[j], my[c0 + j]) for j in range(min(S, hi - c0))]
            nl = len(live)
            # (gigl_dist_plan_run_interleaved: phase 0 of every plan in flight, then phase 1 of every plan, ... from C++)
            pa = (C.c_void_p * nl)(*[sl.plan._plan for sl, _ in live])
            ra = (C.c_void_p * nl)(*[r_.data_ptr() for _, r_ in live])
            oa = (C.c_void_p * nl)(*[sl.out.data_ptr() for sl, _ in live])
            rc = lib.gigl_dist_plan_run_interleaved(pa, nl, ra, 42, oa)
            if rc != 0:
                from gigl_amd._lib import check
                check(rc, slots[0].eng._ctx)
            if acc is not None:
                for sl, _ in live:
                    sl.plan.stats(acc)
                    sl.plan.bucket_fill(fill_acc)

    def sync_all(slots):
        for sl in slots:
            sl.stream.synchronize()

    # ---- warm-up with default row buckets, then size them from what the warm-up saw (+10 %): rows are the bytes that
    # matter on the links, so the send buffers should not be padded more than that
    fill_acc = torch.zeros(4, dtype=torch.int64, device=dev)
    slots = make_slots(0)
    acc0 = torch.zeros(STATS_LEN, dtype=torch.int64, device=dev)
    run_calls(slots, 0, Wp // G, acc0)
    sync_all(slots)
    most = acc0[STATS["pull_bucket_max"]:STATS["pull_bucket_max"] + 1].clone()
    all_reduce(most, dist.ReduceOp.MAX)
    if int(acc0[STATS["overflow"]].item()):
        raise RuntimeError("bucket overflow during warm-up")
    pull_cap = int(int(most.item()) * 1.1) + 64
    most_b = fill_acc[2:3].clone()
    all_reduce(most_b, dist.ReduceOp.MAX)
    pull_cap_b = int(int(most_b.item()) * 1.1) + 64 if proj_table is not None else 0
    close_slots(slots)
    slots = make_slots(pull_cap, pull_cap_b)
    run_calls(slots, 0, Wp // G)
    sync_all(slots)
    setup_s = time.time() - t0

    def seg_calls(r):
        lo = (Wp + (r % N_SEG) * K_rep) // G
        return lo, lo + K_rep // G

    # ---- untimed: exact counts of the pool segments (sampling is deterministic)
    seg_acc = torch.zeros((N_SEG, STATS_LEN), dtype=torch.int64, device=dev)
    for sg in range(N_SEG):
        run_calls(slots, *seg_calls(sg), acc=seg_acc[sg])
    sync_all(slots)
    seg_stats = seg_acc.cpu().numpy().astype(np.float64)
    if seg_stats[:, STATS["overflow"]].any():
        raise RuntimeError("bucket overflow in a benchmark batch: rerun with a larger --shard-group slack")
    # ---- calibration repetition, then the timed region
    torch.cuda.synchronize()
    tc = time.perf_counter()
    run_calls(slots, *seg_calls(0))
    sync_all(slots)
    t_cal = time.perf_counter() - tc
    rr = torch.tensor([int(min(max(np.ceil(args.min_seconds / max(t_cal, 1e-6)), args.min_reps), 2000))],
                      dtype=torch.int64, device=dev)
    all_reduce(rr, dist.ReduceOp.MAX)
    reps = int(rr.item())
    rep_s = []
    moved0 = [sl.comm.traffic() for sl in slots]  # (bytes this rank's communicators put on the links so far)
    for r in range(reps):
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_calls(slots, *seg_calls(r))
        sync_all(slots)
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t1)
    dist.barrier()
    moved1 = [sl.comm.traffic() for sl in slots]
    moved_t = torch.tensor([sum(b_[0] - a_[0] for a_, b_ in zip(moved0, moved1)),
                            sum(b_[1] - a_[1] for a_, b_ in zip(moved0, moved1))], dtype=torch.float64, device=dev)
    moved_max = moved_t.clone()
    all_reduce(moved_t, dist.ReduceOp.SUM)
    all_reduce(moved_max, dist.ReduceOp.MAX)
    comm_ranks, comm_kind = slots[0].comm.world, slots[0].comm.kind  # (from the communicator: gigl_comm_info)
    # ---- per-kernel HIP-event times of one more (untimed) repetition: the library's timers on every plan's ctx; the
    # plans stay interleaved as in the timed region, so an interval includes what the other plans' kernels took from it
    prof_names = ["expand", "union_insert", "union_relax", "union_nodes", "union_edge_sort", "union_csr", "gather_mean",
                  "linear", "dist_prep", "dist_serve"]
    for sl in slots:
        sl.eng.profile_enable(prof_names, capacity=8192)
        sl.eng.profile_reset()
    run_calls(slots, *seg_calls(0))
    sync_all(slots)
    prof_sh = {k: [sum(x) for x in zip(*[sl.eng.profile_read(k) for sl in slots])] for k in prof_names}
    for sl in slots:
        sl.eng.profile_enable([], 0)
    dist.barrier()
    rep_t = torch.tensor(rep_s, dtype=torch.float64, device=dev)
    all_reduce(rep_t, dist.ReduceOp.MAX)
    seg_use = np.array([sum(1 for r in range(reps) if r % N_SEG == sg) for sg in range(N_SEG)], dtype=np.float64)
    tot = torch.tensor((seg_stats * seg_use[:, None]).sum(0), dtype=torch.float64, device=dev)
    all_reduce(tot, dist.ReduceOp.SUM)
    tot = tot.cpu().numpy()
    rep_np = rep_t.cpu().numpy() + K_rep * pre_per_step_s  # (+ every step's share of the shard projection, if any)
    elapsed = float(rep_np.sum())
    steps_total = reps * K_rep
    sampled_all, aggregated_all = float(tot[STATS["sampled"]]), float(tot[STATS["aggregated"]])
    pulled_all = float(tot[STATS["pulled_rows"]])
    row_bytes = hid * 4 if (args.project_on_owner or proj_table is not None) else d * 2
    ms_rep = rep_np / K_rep * 1e3
    if rank == 0:
        q = lambda a, p: float(np.percentile(a, p))
        sent_per_step = world * pull_cap * row_bytes * (2 if args.project_on_owner else 1) / G  # (approx. for B rows)
        line = {
            "metric": "sampled+aggregated edges/s", "value": (sampled_all + aggregated_all) / elapsed,
            "unit": "edges/s", "n_gpus": world, "steps": steps_total, "warmup": Wp,
            "ms_per_step": elapsed / steps_total * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "steps_requested": K,
            # the communicator's own view (gigl_comm_info / gigl_comm_traffic), not WORLD_SIZE: how many ranks the
            # exchanges of the timed region ran between, through which transport, and the bytes they put on the links
            "rccl_ranks": int(comm_ranks) if comm_kind == 0 else 0,
            "comm": {"ranks": int(comm_ranks),
                     "transport": {0: "rccl", 1: "in-process", 2: "host-callback"}.get(int(comm_kind), str(comm_kind)),
                     "xgmi_bytes_per_step_per_gpu_mean": float(moved_t[0].item()) / max(steps_total * world, 1),
                     "xgmi_bytes_per_step_busiest_gpu": float(moved_max[0].item()) / max(steps_total, 1),
                     "xgmi_bytes_per_step_had_blocks_been_full": float(moved_t[1].item()) / max(steps_total * world, 1),
                     "measured": "gigl_comm_traffic over the timed region: bytes sent to OTHER ranks by this rank's "
                                 "communicators (all plans in flight); 0 at one rank"},
            "timing": {"repetitions": reps, "steps_per_repetition": K_rep, "timed_region_s": round(elapsed, 3),
                       "ms_per_step_median": q(ms_rep, 50), "ms_per_step_p10": q(ms_rep, 10),
                       "ms_per_step_p90": q(ms_rep, 90)},
            "config": {"workload": f"MAG240M-shaped RMAT x{args.shard_scale:g}: N={n} E={int(e_local.item())} directed, "
                                   f"D={d} fp16 features, hash-partitioned over {world} rank(s) (owner = id % world), "
                                   f"fanout={fanouts} B={B}/GPU "
                                   f"{'GAT heads 2 ' if gat else 'GraphSAGE '}{d}->{hid}->{out_dim}, sampler mode=parity, "
                                   f"{G} batches per exchange, {S} plans in flight, "
                                   f"{'%.3g %% of the nodes replicated as hot rows, ' % (100 * hot_frac) if n_hot else ''}"
                                   + ("rows projected on the owner (256 fp32)" if args.project_on_owner else
                                      "rows pre-projected once per rank (256 fp32 W_l x rows pulled)" if proj_table is not None
                                      else "raw rows (768 fp16)"),
                       "projected_input": (None if proj_table is None else {
                           "precompute_s": round(pre_s, 4), "steps_per_pass_per_rank": steps_per_pass,
                           "charged_ms_per_step": pre_per_step_s * 1e3}),
                       "graph": "CSC rows + feature rows of the owned nodes per rank; per-hop all-to-all frontier "
                                "exchange and feature pull of the unique union-graph nodes, issued by the library "
                                "(gigl_dist_plan, RCCL)",
                       "sampled_edges_per_step": sampled_all / (steps_total * world),
                       "aggregated_edges_per_step": aggregated_all / (steps_total * world),
                       "pulled_feature_rows_per_step": pulled_all / (steps_total * world),
                       "pulled_feature_bytes_per_s": pulled_all * row_bytes / elapsed,
                       "row_bucket_rows_per_peer": pull_cap,
                       "row_bytes_sent_per_step_per_rank": sent_per_step,
                       "row_bucket_fill": pulled_all / (steps_total * world) / max(world * pull_cap / G, 1),
                       "setup_s": round(setup_s, 1)},
            "roofline": None, "cpu_baseline": None,
        }
        # ---- roofline of the dominant kernel group (this rank's segment-0 repetition, counted on the device): the byte
        # model of the single-GPU line; layer 0 reads pre-projected fp32 rows of `hid` columns when the
        # table was projected, stored fp16 rows of D otherwise
        if not gat:
            st0 = seg_stats[0]
            agg0, agg1 = st0[STATS["agg_layer0"]], st0[STATS["agg_layer0"] + 1]
            rows0, rows1 = st0[STATS["rows_layer0"]], st0[STATS["rows_layer0"] + 1]
            if proj_table is not None or args.project_on_owner:
                b_gather = agg0 * (4 + hid * 4) + rows0 * (8 + 2 * hid * 4)
            else:
                b_gather = agg0 * (4 + d * 2) + rows0 * (8 + d * 2 + 2 * d * 4)
            b_gather += agg1 * (4 + hid * 4) + rows1 * (8 + hid * 4)
            byk = {k: {"ms_per_step": round(v[0] / K_rep, 5), "launches": int(v[1])} for k, v in prof_sh.items() if v[0] > 0}
            if byk:
                dom = max(byk, key=lambda k: byk[k]["ms_per_step"])
                if "gather_mean" in byk:
                    gm = byk["gather_mean"]
                    gm.update(bound="hbm", achieved=round(b_gather / K_rep / (gm["ms_per_step"] * 1e-3) / 1e9, 1),
                              peak=HBM_PEAK_GBS, unit="GB/s")
                    gm["frac"] = round(gm["achieved"] / HBM_PEAK_GBS, 4)
                hk = byk.get("gather_mean", byk[dom])
                line["roofline"] = {
                    "bound": "hbm", "kernel": "gather_mean", "achieved": hk.get("achieved"), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hk.get("frac"), "traffic": None, "dominant": dom,
                    "alg_bytes_per_launch": round(b_gather / max(hk["launches"], 1)),
                    "avg_launch_us": round(hk["ms_per_step"] * K_rep / max(hk["launches"], 1) * 1e3, 2),
                    "launches": hk["launches"],
                    "timing": f"HIP events on the plans' streams over one untimed repetition of the timed calls ({S} plans "
                              "in flight: intervals include overlap with the other plans' kernels)",
                    "by_kernel": byk}
        # ---- second roofline: xGMI.  Bytes a rank puts on its links per step = what it sends to the
        # other world-1 ranks: per hop the request buckets (8 B per entry) and, as an owner, the answer buckets
        # (4*f B per entry); then the id buckets of the feature pull (4 B) and the row buckets.  Buckets travel whole
        # (fixed capacity, equal split), so `sent` counts padding; `payload` counts the requested entries only.
        step_s = elapsed / steps_total
        peers = world - 1
        m_k, hop_sent = G * B, 0.0
        for f in fanouts:
            cap_k = m_k if world <= 2 else min(m_k, int(1.5 * m_k / world) + 512)
            hop_sent += peers * cap_k * (8 + 4 * f)
            m_k *= f
        rows_sent = peers * pull_cap * (row_bytes + 4) * (2 if args.project_on_owner else 1) + \
            peers * pull_cap_b * (row_bytes + 4)
        sent_step = (hop_sent + rows_sent) / G
        payload_step = pulled_all / (steps_total * world) * (row_bytes + 4) + \
            sampled_all / (steps_total * world) * 4 * peers / max(world, 1)
        peak_gbs = 7 * 153.0
        line["roofline_xgmi"] = {
            "bound": "xgmi", "peak": peak_gbs, "unit": "GB/s per GPU (7 links x 153 GB/s)",
            "sent_bytes_per_step_per_gpu": sent_step, "payload_bytes_per_step_per_gpu": payload_step,
            "achieved": sent_step / step_s / 1e9, "frac": sent_step / step_s / 1e9 / peak_gbs,
            "payload_achieved": payload_step / step_s / 1e9, "links_in_use": min(peers, 7),
            "transport": ("RCCL ncclSend / ncclRecv groups issued by the library" if dist.get_backend() == "nccl"
                          else "host callback over " + dist.get_backend() + " (functional check, not xGMI)"),
            "ranks": world}
        if world == 1 and not gat and not args.no_cpu_baseline and not sub:
            # (one rank: the shard is the whole graph, so the single-GPU line's CPU port applies as it is — the oracle
            # sampler + collate and the fp32 CPU forward over full batches of this graph; at N > 1 no host holds the graph)
            one, allc = run_cpu_baseline(eng, model, my.view(-1, B), fanouts, Wp, n, d)
            line["cpu_baseline"], line["cpu_baseline_all_cores"] = one, allc
        if not sub:
            emit(line)
    else:
        line = None
    dist.barrier()
    close_slots(slots)
    if not sub:
        dist.destroy_process_group()
    eng.close()
    return line


def run_emulated_world(args, local_rank=0, sub=False):
    """BASELINE configs[2] without an 8-GPU node: all W ranks of the hash-partitioned job as ctxs of ONE process on one
    GPU (gigl_dist_init_local: the in-process transport the parity tests use — every exchange is a device copy), at the
    largest MAG240M-shaped scale the GPU holds.  The step's CODE is the multi-GPU step's (gigl_dist_plan_run_local issues
    every rank's phases in the order the ranks would), so what each rank would put on its links is MEASURED: pulled rows /
    bytes per rank and step, bucket fill (padding), what hub-row replication takes off the links, and the per-rank
    compute time (the W ranks' kernels share this GPU: time of a step of all ranks / W).  What is NOT measured is xGMI:
    `projection` combines the measured bytes with 7 links x 153 GB/s per GPU and says so."""
    from gigl_amd._lib import STATS, STATS_LEN
    from gigl_amd.dist import Comm, DistSagePlan
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE

    W = int(args.emulate_world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    fanouts = [int(v) for v in args.fanouts.split(",")]
    L = len(fanouts)
    B, G = args.batch, max(1, args.shard_group)
    d, hid, out_dim = 768, 256, 256
    t0 = time.time()
    free_b, _ = torch.cuda.mem_get_info(dev)
    # stored row + pre-projected row + the sampler's threshold table over (hops + 1) * N + graph + the plans' id-indexed pull
    # bookkeeping (8 B per node and plan); half of the free memory: workspaces and the generator's temporaries need the rest
    per_node = d * 2 + 2 * hid * 4 + 13 * (L + 1) + 64 + 8 * W
    scale = args.shard_scale if args.shard_scale > 0 else min(1.0, 0.5 * free_b / per_node / 244_160_499)
    n = max(int(244_160_499 * scale), W * 1024)
    e_total = max(int(1_728_364_232 * scale), 1)
    scale_bits = max(int(np.ceil(np.log2(n))), 10)
    # ---- every rank's shard (the generator of run_sharded: same seeded chunks, each edge to the owner of its destination)
    keys = [[] for _ in range(W)]
    chunk = 1 << 26
    for ci, c0 in enumerate(range(0, e_total, chunk)):
        m = min(chunk, e_total - c0)
        src, dst = rmat_edges_gpu(scale_bits, m, seed=3 + 7919 * ci, device=dev)
        src = (src * 0x9E3779B1) % n
        dst = (dst * 0x9E3779B1) % n
        for r in range(W):
            keep = (dst % W) == r
            keys[r].append(((dst[keep] // W) << 32) | src[keep])
        del src, dst
    engs, n_local, maxdeg, e_sum = [], [], 0, 0
    occ = torch.zeros(n, dtype=torch.int32, device=dev)
    for r in range(W):
        key = torch.unique(torch.cat(keys[r]))
        keys[r] = None
        nl = (n - r + W - 1) // W
        rowptr = torch.zeros(nl + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(key >> 32, minlength=nl), 0)
        col = (key & 0xFFFFFFFF).to(torch.int32)
        maxdeg = max(maxdeg, int((rowptr[1:] - rowptr[:-1]).max()))
        e_sum += int(col.numel())
        occ += torch.bincount(col.to(torch.int64) & 0xFFFFFFFF, minlength=n).to(torch.int32)
        eng = HipEngine(local_rank)
        eng.load_csc(rowptr, col)
        engs.append(eng)
        n_local.append(nl)
        del key, rowptr, col
    torch.manual_seed(0)
    model = GraphSAGE(d, hid, out_dim, num_layers=L).to(dev)
    w, bs = model.fused_params()
    hot_frac = 0.01 if args.shard_hot_frac < 0 else float(args.shard_hot_frac)
    n_hot = int(n * hot_frac)
    hot_ids = None
    if n_hot > 0:
        hot_ids = torch.topk(occ.to(torch.int64) * (1 << 32) + (n - 1 - torch.arange(n, device=dev)), n_hot).indices
        hot_ids = hot_ids.to(torch.int32).contiguous()
    del occ
    use_proj = args.project_input != "off" and L == 2
    proj, pre_s = [], 0.0
    hot_rows = torch.zeros((n_hot, hid if use_proj else d), device=dev, dtype=torch.float32 if use_proj else torch.float16) \
        if n_hot else None
    step_rows = max(1, (1 << 28) // d)
    for r in range(W):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + r)
        x_local = torch.empty((n_local[r], d), device=dev, dtype=torch.float16)
        for i in range(0, n_local[r], step_rows):
            x_local[i:i + step_rows] = torch.randn((min(step_rows, n_local[r] - i), d), generator=g, device=dev).to(torch.float16)
        engs[r].load_features(x_local)
        pt = None
        if use_proj:
            pt = torch.empty((n_local[r], 2 * hid), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            engs[r].project_features(w[0], out=pt)
            torch.cuda.synchronize()
            pre_s = max(pre_s, time.perf_counter() - tp)
        proj.append(pt)
        if n_hot:
            hi = hot_ids.to(torch.int64) & 0xFFFFFFFF
            mine = (hi % W) == r
            hot_rows[mine] = (pt[hi[mine] // W, :hid] if use_proj else x_local[hi[mine] // W])
        del x_local
    torch.cuda.empty_cache()
    bound = (L + 1) * n + 42 * L + maxdeg
    mwe = bound if bound < (1 << 30) else -1
    comms = Comm.local(engs)
    K = max(4, min(args.steps // G, 24))  # calls (G batches per rank each) per measurement
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    roots = torch.randint(0, n, ((K + 2) * W, G * B), generator=gp).to(torch.int32).to(dev)

    def make_plans(pull_cap, pull_cap_b, hot):
        plans = [DistSagePlan(comms[r], w, bs, G * B, fanouts, group_roots=B, pull_cap=pull_cap, max_window_end=mwe,
                              projected=proj[r], pull_cap_b=pull_cap_b) for r in range(W)]
        if hot and n_hot:
            for pl in plans:
                pl.set_hot_rows(hot_ids, hot_rows)
        return plans

    def run_calls(plans, lo, hi, accs=None, fills=None):
        outs = [pl.new_out() for pl in plans]
        for c in range(lo, hi):
            DistSagePlan.run_local(plans, [roots[c * W + r] for r in range(W)], outs)
            if accs is not None:
                for r, pl in enumerate(plans):
                    pl.stats(accs[r])
                    pl.bucket_fill(fills[r])
        torch.cuda.synchronize()

    traffic = []  # per rank: [moved, full-block] bytes per step of the last measurement
    ktime = {}    # kernel group -> [ms, launches] summed over the W ranks, last measurement
    EMU_PROF = ["expand", "expand_heavy", "union_insert", "union_relax", "union_nodes", "union_edge_sort", "union_csr",
                "gather_mean", "linear", "dist_prep", "dist_serve"]

    def measure(hot):
        # bucket capacities from two warm-up calls (+10 %), as the multi-process bench does
        acc0 = [torch.zeros(STATS_LEN, dtype=torch.int64, device=dev) for _ in range(W)]
        fill0 = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(W)]
        plans = make_plans(0, 0, hot)
        run_calls(plans, 0, 2, acc0, fill0)
        if any(int(a[STATS["overflow"]]) for a in acc0):
            raise RuntimeError("bucket overflow during the emulated world's warm-up")
        pull_cap = int(max(int(a[STATS["pull_bucket_max"]]) for a in acc0) * 1.1) + 64
        pull_cap_b = (int(max(int(f[2]) for f in fill0) * 1.1) + 64) if use_proj else 0
        for pl in plans:
            pl.close()
        plans = make_plans(pull_cap, pull_cap_b, hot)
        run_calls(plans, 0, 2)
        accs = [torch.zeros(STATS_LEN, dtype=torch.int64, device=dev) for _ in range(W)]
        fills = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(W)]
        run_calls(plans, 2, 2 + K, accs, fills)  # counted (untimed)
        torch.cuda.synchronize()
        tr0 = np.array([c.traffic() for c in comms], dtype=np.float64)
        t1 = time.perf_counter()
        run_calls(plans, 2, 2 + K)               # timed: all W ranks' steps on this one GPU
        dt = time.perf_counter() - t1
        # bytes each rank handed to the transport for OTHER ranks over the timed calls (gigl_comm_traffic): as moved —
        # the feature-row blocks at the size of their request counts — and as full-capacity blocks would have been
        traffic.clear()
        traffic.extend(((np.array([c.traffic() for c in comms], dtype=np.float64) - tr0) / (K * G)).tolist())
        # ---- the ranks' own KERNEL time (untimed repetition, every HIP-event timer of the library on): the W ranks share
        # one stream here, so an interval is its kernels' duration (+ the host's gap between the two event records when
        # the launch queue runs dry, which it does: an upper bound).  The in-process transport's device copies stand in
        # for RCCL and are not kernels of a rank: not counted.
        for e_ in engs:
            e_.profile_enable(EMU_PROF, capacity=(K + 2) * 64)
            e_.profile_reset()
        run_calls(plans, 2, 2 + K)
        kt = {k: [sum(x) for x in zip(*[e_.profile_read(k) for e_ in engs])] for k in EMU_PROF}
        for e_ in engs:
            e_.profile_enable([], 0)
        ktime.clear()
        ktime.update({k: v for k, v in kt.items() if v[0] > 0})
        st = np.stack([a.cpu().numpy().astype(np.float64) for a in accs])
        fl = np.stack([f.cpu().numpy().astype(np.float64) for f in fills])
        if st[:, STATS["overflow"]].any():
            raise RuntimeError("bucket overflow in the emulated world")
        for pl in plans:
            pl.close()
        return st, fl, dt, pull_cap, pull_cap_b

    row_bytes = hid * 4 if use_proj else d * 2
    steps = K * G  # steps per rank in a measurement
    res = {}
    for tag, hot in (("hot_rows", True), ("no_replication", False)) if n_hot else (("no_replication", False),):
        st, fl, dt, pull_cap, pull_cap_b = measure(hot)
        pulled = st[:, STATS["pulled_rows"]] / steps  # rows per step, per rank
        # what a rank sends as an OWNER (= what it receives as a requester, by symmetry of the measured totals): the row
        # buckets travel whole (fixed capacity, equal split): W - 1 peers x capacity x (row + id) per call
        sent_rows_bytes = (W - 1) * (pull_cap + pull_cap_b) * (row_bytes + 4) / G
        m_k, hop_bytes = G * B, 0.0
        for f in fanouts:
            cap_k = min(m_k, int(1.5 * m_k / W) + 512)
            hop_bytes += (W - 1) * cap_k * (8 + 4 * f)
            m_k *= f
        hop_bytes /= G
        payload = pulled * (row_bytes + 4)
        compute_ms = dt / (K * G * W) * 1e3
        # kernel time per rank-step by group (HIP events), and the byte model of the dominant one (as run_sharded's)
        kg = {k: round(v[0] / (K * G * W), 6) for k, v in ktime.items()}
        kernel_ms = float(sum(kg.values()))
        agg0 = st[:, STATS["agg_layer0"]].sum() / (steps * W)
        agg1 = st[:, STATS["agg_layer0"] + 1].sum() / (steps * W)
        rows0 = st[:, STATS["rows_layer0"]].sum() / (steps * W)
        rows1 = st[:, STATS["rows_layer0"] + 1].sum() / (steps * W)
        b_gather = (agg0 * (4 + hid * 4) + rows0 * (8 + 2 * hid * 4) if use_proj else
                    agg0 * (4 + d * 2) + rows0 * (8 + d * 2 + 2 * d * 4)) + agg1 * (4 + hid * 4) + rows1 * (8 + hid * 4)
        gm_ms = kg.get("gather_mean", 0.0)
        tr = np.array(traffic, dtype=np.float64)  # [W, 2]
        moved_step, full_step = float(tr[:, 0].max()), float(tr[:, 1].max())  # the busiest rank's
        per_link = moved_step / (W - 1)  # bytes per peer pair and step: one xGMI link each (W <= 8)
        link_ms = per_link / 153e9 * 1e3
        edges_step = (st[:, STATS["sampled"]] + st[:, STATS["aggregated"]]).sum() / (steps * W)
        res[tag] = {
            "pulled_rows_per_step_per_rank": [round(float(v), 1) for v in pulled],
            "pulled_rows_per_step_mean": float(pulled.mean()),
            "row_payload_bytes_per_step_per_rank": float(payload.mean()),
            "bytes_sent_per_step_per_rank": [round(float(v)) for v in tr[:, 0]],
            "bytes_sent_per_step_busiest_rank": moved_step,
            "bytes_sent_with_full_blocks_busiest_rank": full_step,
            "exchange_sizes": "measured by the transport (gigl_comm_traffic) over the timed calls: the feature-row blocks "
                              "travel at the size of their request counts (the counts ride with the id request), the id / "
                              "neighbour blocks of the hops and the id buckets at their fixed capacity",
            "row_bytes_at_full_capacity_per_step_per_rank": float(sent_rows_bytes),
            "hop_exchange_bytes_sent_per_step_per_rank": float(hop_bytes),
            "row_bucket_capacity_per_peer": [pull_cap, pull_cap_b],
            # occupied entries of the row buckets / their capacity, summed over the W - 1 peers (gigl_dist_plan_bucket_fill:
            # first pull, and the W_r x pull of a pre-projected plan): 1 - fill is padding that would travel over xGMI
            "row_bucket_fill": float((fl[:, 1].sum() + fl[:, 3].sum()) /
                                     max(K * W * (W - 1) * (pull_cap + pull_cap_b), 1)),
            "row_bucket_fill_fullest": float(max(fl[:, 0].max() / max(pull_cap, 1),
                                                 fl[:, 2].max() / max(pull_cap_b, 1) if pull_cap_b else 0.0)),
            "rows_in_buckets_per_step_per_rank": float((fl[:, 1].sum() + fl[:, 3].sum()) / (K * G * W)),
            "sampled_plus_aggregated_edges_per_step_per_rank": float(edges_step),
            "wall_ms_per_step_per_rank": compute_ms,
            "kernel_ms_per_step_per_rank": kernel_ms,
            "kernel_ms_by_group": kg,
            "sharded_only_kernel_share": round((kg.get("dist_prep", 0.0) + kg.get("dist_serve", 0.0)) / max(kernel_ms, 1e-12), 4),
            "roofline": {"bound": "hbm", "kernel": "gather_mean", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "achieved": round(b_gather / max(gm_ms * 1e-3, 1e-12) / 1e9, 1),
                         "frac": round(b_gather / max(gm_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
                         "alg_bytes_per_rank_step": round(b_gather), "ms_per_rank_step": gm_ms,
                         "timing": "HIP events around the rank's launches; one stream for all W ranks: kernels run alone"},
            "measured": "all of the above: counted on the device / timed on this GPU with the W ranks sharing it.  wall_ms = "
                        "host wall clock of eager launches, one plan per rank, phases of the W ranks issued in turn by one "
                        "host thread (host-bound: the GPU idles most of it); kernel_ms = the HIP-event time of the ranks' own "
                        "kernels (an upper bound of the kernel time: profiles/r05*_emulated_world8_kernel_time.txt has the "
                        "rocprofv3 figure of the same run); the in-process transport's copies stand in for RCCL and are in "
                        "neither",
            "projection": {
                "label": "PROJECTION, not a measurement: measured bytes over 7 x 153 GB/s xGMI links per GPU (one link per "
                         "peer at W = 8) against the measured per-rank KERNEL time with NO overlap between a rank's kernels "
                         "assumed (the single-rank bench overlaps three plans); exchanges assumed to overlap compute across "
                         "the plans in flight",
                "link_ms_per_step": link_ms, "bound": "xgmi" if link_ms > kernel_ms else "compute",
                "step_ms": max(link_ms, kernel_ms),
                "whole_node_edges_per_s": W * edges_step / (max(link_ms, kernel_ms) * 1e-3)}}
    if n_hot:
        a, b_ = res["no_replication"]["pulled_rows_per_step_mean"], res["hot_rows"]["pulled_rows_per_step_mean"]
        res["hot_row_hit_rate"] = {"replicated_fraction_of_nodes": hot_frac, "replica_bytes_per_rank": int(n_hot * row_bytes),
                                   "pulled_rows_without": a, "pulled_rows_with": b_, "rows_taken_off_the_links": 1.0 - b_ / max(a, 1.0)}
    best = res.get("hot_rows", res["no_replication"])
    line = {
        "metric": "sampled+aggregated edges/s", "value": best["projection"]["whole_node_edges_per_s"], "unit": "edges/s",
        "n_gpus": 1, "emulated_world": W, "steps": steps, "ms_per_step": best["projection"]["step_ms"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_is": "a PROJECTION for W GPUs from quantities measured on ONE GPU (see emulated.*.projection.label); the "
                    "measured quantities are in `emulated`",
        "config": {"workload": f"MAG240M-shaped RMAT x{scale:.3g}: N={n} E={e_sum} directed, D={d} fp16, hash-partitioned "
                               f"over {W} emulated ranks in one process (owner = id % {W}), fanout={fanouts} B={B}/rank, "
                               f"GraphSAGE {d}->{hid}->{out_dim}, {G} batches per exchange, "
                               + ("rows pre-projected once per rank (256 fp32 W_l x rows pulled)" if use_proj else "raw rows"),
                   "transport": "gigl_dist_init_local (in-process: every exchange is a device copy on this GPU)",
                   "projection_precompute_s_per_rank": round(pre_s, 4), "setup_s": round(time.time() - t0, 1)},
        "emulated": res, "roofline": best["roofline"], "cpu_baseline": None}
    for c in comms:
        c.close()
    for e in reversed(engs):
        e.close()
    torch.cuda.empty_cache()
    if not sub:
        emit(line)
    return line


def run_train(args, rank, world, local_rank):
    """--train: one TRAINING step per batch on the in-HBM route (gigl_amd/hbm.py, what Trainer.run drives): k-hop sample
    + batch union graph in HBM, GraphSAGE forward with autograd over the union graph (trimmed schedule), cross-entropy
    on the root rows, backward (scatter of the layer-1 input gradient by gigl_gather_reduce_backward: fp32 atomics; the
    projections' backward products) and the Adam update (lr 0.01, weight decay 5e-4: the reference spec's defaults,
    node_classification_modeling_task_spec.py:51-57,134-173).  One batch per step, one stream, launches eager (autograd
    drives them from Python).  Edges are counted like the inference line (sampled + the edges the FORWARD reductions
    consume); the backward scatter's edges are reported next to them."""
    import torch.nn.functional as F
    from gigl_amd._lib import GIGL_META_LEVEL0, KERNEL_IDS, MODE_FAST, MODE_SPARK_HASH
    from gigl_amd.engine import HipEngine
    from gigl_amd.hbm import ResidentGraph
    from gigl_amd.models import GraphSAGE

    torch.cuda.set_device(local_rank)
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    L = len(fanouts)
    B, K, W = args.batch, max(64, args.steps), max(8, args.warmup)
    t0 = time.time()
    n, d = build_workload(eng, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    esz = 4 if wl_dtype == torch.float32 else 2
    torch.manual_seed(0)
    model = GraphSAGE(d, hid, out_dim, num_layers=L).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
    model.train()
    mode = MODE_SPARK_HASH if args.mode == "parity" else MODE_FAST
    st = torch.cuda.Stream(device=dev)
    eng.bind_stream(st)
    resident = ResidentGraph.from_engine(eng, np.arange(n, dtype=np.int64), fanouts, mode=mode)
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    pool = W + K
    perm = torch.randperm(n, generator=gp)
    if perm.numel() < pool * world * B:
        perm = perm.repeat((pool * world * B + perm.numel() - 1) // perm.numel())
    my = perm[: pool * world * B].view(pool * world, B)[rank::world].to(torch.int32).to(dev).contiguous()
    labels = torch.randint(0, out_dim, (n,), generator=gp).to(dev)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    counts = torch.zeros(3, dtype=torch.int64, device=dev)  # sampled, forward-aggregated, backward-scattered edges

    def step(i, count=False):
        with torch.cuda.stream(st):
            roots = my[i]
            hb = resident.hip_batch(roots, train=True)
            out = model(hb)
            loss = F.cross_entropy(out[hb.root_local.long()], labels[roots.long() & 0xFFFFFFFF])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if count:
                u = hb.union
                rowlen = (u.rowend - u.rowptr).to(torch.int64)
                ar = torch.arange(rowlen.numel(), device=dev)
                per_layer = [(rowlen * (ar < u.meta[GIGL_META_LEVEL0 + (L - 1 - l)])).sum() for l in range(L)]
                counts.add_(torch.stack([sum(c.sum() for c in hb.tree.cnt).to(torch.int64), sum(per_layer),
                                         sum(per_layer[1:]) if L > 1 else per_layer[0] * 0]))
        return loss

    # (a counter-collection child of the library-plan line runs the plan's steps only: every kernel it counts is the plan's)
    plan_only_child = bool(args.timed_only and not os.environ.get("GIGL_BENCH_TRAIN_EAGER") and
                           not os.environ.get("GIGL_BENCH_TRAIN_AUTOGRAD"))
    for i in range(0 if plan_only_child else W):
        step(i)
    st.synchronize()
    # the step replayed as ONE HIP graph (gigl_amd.hbm.GraphedTrainStep: what the trainer's in-HBM route runs): the same
    # launches without the host between them.  The eager step above stays for the per-kernel timers and the counts.
    eager_step, graphed, driver = step, None, "eager launches from Python (torch autograd)"
    lib_plan = None
    if not os.environ.get("GIGL_BENCH_TRAIN_EAGER") and not os.environ.get("GIGL_BENCH_TRAIN_AUTOGRAD"):
        # the library's training step (gigl_sage_train_plan_*: what Trainer.run's in-HBM route runs for plain mean-GraphSAGE
        # encoders): the whole step is one captured library call, no torch kernel in it
        from gigl_amd.engine import SageTrainPlan
        try:
            torch.cuda.synchronize()
            lib_plan = SageTrainPlan(eng, model, B, fanouts, lr=0.01, weight_decay=5e-4)
            lab_pool = labels[my.long() & 0xFFFFFFFF]  # [pool, B]

            def step(i, count=False):  # noqa: F811
                if count:
                    return eager_step(i, True)
                # (the next batch's sampling + union overlap this batch's layers: gigl_sage_train_plan_prefetch)
                pf = os.environ.get("GIGL_BENCH_TRAIN_NO_PREFETCH")  # ("1": none, "2": one batch ahead only)
                nxt = my[i + 1] if i + 1 < my.shape[0] and pf != "1" else None
                nxt2 = my[i + 2] if i + 2 < my.shape[0] and not pf else None
                with torch.cuda.stream(st):
                    return lib_plan.step(my[i], lab_pool[i], sampling_seed=resident.seed, mode=mode, next_roots=nxt,
                                         next_roots2=nxt2)
            for i in range(min(W, 4)):  # (eager step, capture, replays)
                step(i)
            st.synchronize()
            driver = "gigl_sage_train_plan_step: one library call per step, replayed as one hipGraph"
        except NotImplementedError as exc:
            print(f"train: library training plan not applicable ({exc})", file=sys.stderr)
            lib_plan, step = None, eager_step
    if lib_plan is None and not os.environ.get("GIGL_BENCH_TRAIN_EAGER") and not args.timed_only:
        from gigl_amd.hbm import GraphedTrainStep
        try:
            graphed = GraphedTrainStep(resident, model, opt, B, my[0], labels[my[0].long() & 0xFFFFFFFF])
            lab_pool = labels[my.long() & 0xFFFFFFFF]  # [pool, B]

            def step(i, count=False):  # noqa: F811
                if count:
                    return eager_step(i, True)
                return graphed.step(my[i], lab_pool[i])
            st = graphed.stream
            driver = "one HIP graph per step (GraphedTrainStep), replayed over static inputs"
        except Exception as exc:  # noqa: BLE001 — the eager loop is the same step, only slower
            print(f"train: graph capture unavailable ({type(exc).__name__}: {str(exc)[:300]})", file=sys.stderr)
            eng.bind_stream(st)
    if args.timed_only:  # counter-collection runs
        t1 = time.perf_counter()
        for i in range(W, W + K):
            step(i)
        st.synchronize()
        print(json.dumps({"timed_only": True, "train": True, "steps": K, "workload": wl_name, "batches_per_call": 1,
                          "streams": 1, "ms_per_step": (time.perf_counter() - t1) / K * 1e3,
                          "steps_executed": K + (min(W, 4) if lib_plan is not None else (0 if plan_only_child else W))}))
        eng.close()
        return
    # ---- untimed: exact counts of the timed batches, then every library kernel group's own time (HIP events)
    for i in range(W, W + K):
        step(i, count=True)
    st.synchronize()
    cnt = counts.cpu().numpy().astype(np.float64)
    names = list(KERNEL_IDS)
    P = min(K, 64)
    eng.profile_enable(names, capacity=P * 64)
    for i in range(W, W + P):
        eager_step(i)  # (timed launches must be eager: events inside a captured graph cannot be read)
    torch.cuda.synchronize()
    prof = {k: eng.profile_read(k) for k in names}
    eng.profile_enable([], 0)
    # ---- timed region
    reps = []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < args.min_seconds or len(reps) < 3:
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(W, W + K):
            step(i)
        torch.cuda.synchronize()
        reps.append(time.perf_counter() - t1)
    rep_np = np.array(reps)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor(rep_np, dtype=torch.float64, device=dev)
        all_reduce(tt, dist.ReduceOp.MAX)
        rep_np = tt.cpu().numpy()
    elapsed, steps_total = float(rep_np.sum()), K * len(rep_np)
    sampled, agg, bwd = cnt[0] / K, cnt[1] / K, cnt[2] / K  # per step (this rank)
    # ---- rooflines of the library kernels (single stream: the intervals are the kernels' own)
    dims = [d] + [hid] * (L - 1)
    by_kernel = {}
    for k, (ms, nl) in prof.items():
        if ms <= 0:
            continue
        e = {"ms_per_step": round(ms / P, 5), "launches_per_step": nl / P}
        if k == "gather_bwd":  # per scattered edge: 4 B index + D*4 read-modify-write (atomic) + the row's gradient read
            by = bwd * (4 + 2 * hid * 4) + B * (8 + 3 * hid * 4)
            e.update(bound="hbm", achieved=round(by / (ms / P * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(by / (ms / P * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), alg_bytes_per_step=by)
        by_kernel[k] = e
    lib_ms = sum(v[0] for v in prof.values()) / P
    step_ms = elapsed / steps_total * 1e3
    dominant = max(prof, key=lambda k: prof[k][0])
    flops_fwd = sum(2.0 * (B * sum(int(np.prod(fanouts[:j])) for j in range(L - l))) * 2 * dims[l] *
                    (hid if l < L - 1 else out_dim) for l in range(L))  # (row CAPACITIES: an upper bound)
    roofline = {"bound": by_kernel.get(dominant, {}).get("bound", "latency"), "kernel": dominant,
                "achieved": by_kernel.get(dominant, {}).get("achieved"), "peak": by_kernel.get(dominant, {}).get("peak"),
                "unit": by_kernel.get(dominant, {}).get("unit"), "frac": by_kernel.get(dominant, {}).get("frac"),
                "traffic": None, "dominant": dominant,
                "library_kernel_ms_per_step": round(lib_ms, 5), "step_ms": round(step_ms, 5),
                "library_kernel_share_of_step": round(lib_ms / step_ms, 4),
                "note": "one batch per step, launches driven by torch autograd from Python on one stream: the step is "
                        "bound by launch / host overhead between kernels, not by a kernel (library_kernel_share_of_step); "
                        "the backward scatter (gather_bwd, fp32 atomics) has its own HBM line in by_kernel",
                "by_kernel": by_kernel}
    # ---- HBM traffic of the step by the counters: rocprofv3 PMC passes of a child run of this same command (eager
    # launches of the library plan's kernels: GIGL_TRAIN_PLAN_EAGER), all kernels launched at least once per step
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if rank == 0 and world == 1 and not args.no_live_pmc and not under_profiler and not os.environ.get("GIGL_BENCH_CHILD"):
        passthrough = ["--train", "--workload", args.workload, "--batch", str(B), "--fanouts",
                       ",".join(str(f) for f in fanouts), "--mode", args.mode] + (["--small"] if args.small else [])
        torch.cuda.synchronize()
        doc_, note_ = collect_live_pmc(passthrough, env_extra={"GIGL_TRAIN_PLAN_EAGER": "1"})
        if doc_ is not None:
            n_exec = int(doc_.get("steps_executed") or 68)  # (the child's timed steps + the plan's warm-up steps)
            by_step, per_k = step_traffic_of(doc_, n_exec)
            roofline["step"] = {"bound": "hbm", "traffic_bytes_per_step": round(by_step),
                                "achieved": round(by_step / (step_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(by_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a child run of this command "
                                          f"({n_exec} steps, eager launches), kernels launched at least once per step",
                                "by_kernel_bytes_per_step": {k: round(v) for k, v in sorted(per_k.items(), key=lambda kv: -kv[1])[:12]}}
            dom_pfx = {"linear": ["linear_split_kernel", "linear_weight_grad"], "gather_mean": ["gather_mean_kernel"],
                       "gather_bwd": ["gather_mean_backward_kernel", "gather_reduce_backward"],
                       "expand": ["plan_rows_kernel", "expand_rows_kernel"]}.get(dominant)
            if dom_pfx:
                roofline["traffic"] = round(step_traffic_of(doc_, n_exec, dom_pfx)[0])
                roofline["traffic_unit"] = "HBM bytes per step of the dominant group's kernels (same source as roofline.step)"
            roofline["live_pmc"] = "collected"
        else:
            roofline["live_pmc"] = note_
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_train_baseline(eng, model, my, labels, fanouts, W, out_dim)
    if rank == 0:
        q = lambda a, p: float(np.percentile(a, p))
        ms_rep = rep_np / K * 1e3
        line = {
            "metric": "sampled+aggregated edges/s", "value": (sampled + agg) * world * steps_total / elapsed,
            "unit": "edges/s", "n_gpus": world, "steps": steps_total, "warmup": W, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timing": {"repetitions": len(rep_np), "steps_per_repetition": K, "timed_region_s": round(elapsed, 3),
                       "ms_per_step_median": q(ms_rep, 50), "ms_per_step_p10": q(ms_rep, 10), "ms_per_step_p90": q(ms_rep, 90)},
            "config": {"workload": wl_label + f" N={n} E={eng.n_edges} D={d} {'fp32' if esz == 4 else 'fp16'} features, "
                                            f"fanout={fanouts} B={B}/GPU GraphSAGE {d}->{hid}->{out_dim}: TRAINING step "
                                            "(sample + union in HBM, forward with autograd, cross-entropy, backward, Adam), "
                                            "sampler mode=" + args.mode,
                       "entry": ("engine.SageTrainPlan (gigl_sage_train_plan_*): HipGraphSageNodeClassificationSpec.train on the "
                                 "in-HBM route" if lib_plan is not None else
                                 "ResidentGraph.hip_batch(train=True) -> GraphSAGE._forward_union_autograd (gigl_amd/hbm.py)"),
                       "driver": driver,
                       "sampled_edges_per_step": sampled, "aggregated_edges_per_step": agg,
                       "backward_scattered_edges_per_step": bwd, "setup_s": round(setup_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        emit(line)
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_lp_train(args, rank, world, local_rank):
    """--train --train-task lp: the LINK-PREDICTION training step of the reference's default trainer
    (node_anchor_based_link_prediction_modeling_task_spec.py:334-451: GraphSAGE encoder with L2-normalised output,
    inner-product decoder, Retrieval loss with temperature 0.07 and accidental-hit removal, Adam lr 5e-3 wd 1e-6,
    main_sample_batch_size 2048 anchors with one positive each, 512 random negatives) as ONE library call per step
    (gigl_nablp_train_plan_*: both encodes, the head, the backward of both, the update — replayed as one hipGraph).
    Edges are counted like the inference line, over both encodes (sampled + consumed by the forward reductions)."""
    from gigl_amd._lib import GIGL_META_LEVEL0, MODE_SPARK_HASH
    from gigl_amd.engine import HipEngine, NablpTrainPlan
    from gigl_amd.models import GraphSAGE

    torch.cuda.set_device(local_rank)
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    L = len(fanouts)
    B = args.batch if args.batch != 1024 else 2048  # (the spec's main_sample_batch_size)
    P, NRN, K, W = 1, 512, max(32, args.steps), max(4, args.warmup)
    t0 = time.time()
    n, d = build_workload(eng, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    if wl_directed:
        raise SystemExit("--train-task lp: the synthetic undirected workloads only (positives = sampled out-neighbours)")
    eng._graph_out = eng._graph  # (bidirectionalised: a node's out-neighbours are its in-neighbours)
    torch.manual_seed(0)
    emb = 128
    model = GraphSAGE(d, hid, emb, num_layers=L, should_l2_normalize_embedding_layer_output=True).to(dev)
    st = torch.cuda.Stream(device=dev)
    eng.bind_stream(st)
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    pool = W + K
    perm = torch.randperm(n, generator=gp)
    anchors = perm[: pool * B].view(pool, B).to(torch.int32).to(dev)
    rns = torch.randint(0, n, (pool, NRN), generator=gp).to(torch.int32).to(dev)
    ar = torch.arange(P, device=dev).view(1, P)
    batches = []
    with torch.cuda.stream(st):
        for i in range(pool):
            pos, cnt = eng.sample_positives(anchors[i], P, sampling_seed=42)
            a2 = anchors[i].view(-1, 1)
            roots = torch.cat([a2, torch.where(ar < cnt.view(-1, 1), pos.view(-1, P), a2.expand(-1, P))], dim=1).reshape(-1)
            batches.append((roots.contiguous(), cnt.to(torch.int32).contiguous(), rns[i].contiguous()))
    st.synchronize()
    setup_s = time.time() - t0
    plan = NablpTrainPlan(eng, model, B, P, NRN, fanouts, temperature=0.07, remove_accidental_hits=True, lr=5e-3,
                          weight_decay=1e-6)
    losses = []
    with torch.cuda.stream(st):
        # (eager once, captured on the second step, replayed from then on; --no-train-prefetch: every step samples its own
        # batch first instead of finding it prefetched beside the previous step's layers)
        prefetch = not getattr(args, "no_train_prefetch", False)
        nxt = lambda i, hi: (batches[i + 1][0], batches[i + 1][2]) if prefetch and i + 1 < hi else None
        for i in range(W):
            losses.append(plan.step(*batches[i], next_roots=nxt(i, W)).clone())
    st.synchronize()
    # ---- untimed: exact edge counts of the timed batches (both encodes), through the separate entry points
    counts = np.zeros(2, dtype=np.float64)
    with torch.cuda.stream(st):
        for i in range(W, W + K):
            for r in (batches[i][0], batches[i][2]):
                tree = eng.sample_khop(r, fanouts)
                u = eng.union_build(tree)
                rowlen = (u.rowend - u.rowptr).to(torch.int64)
                a_ = torch.arange(rowlen.numel(), device=dev)
                agg = sum((rowlen * (a_ < u.meta[GIGL_META_LEVEL0 + (L - 1 - l)])).sum() for l in range(L))
                counts += np.array([float(sum(c.sum() for c in tree.cnt)), float(agg)])
    st.synchronize()
    reps = []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < args.min_seconds or len(reps) < 3:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(st):
            for i in range(W, W + K):
                last = plan.step(*batches[i], next_roots=nxt(i, W + K))
        st.synchronize()
        reps.append(time.perf_counter() - t1)
    rep_np = np.array(reps)
    elapsed, steps_total = float(rep_np.sum()), K * len(reps)
    sampled, agg = counts[0] / K, counts[1] / K
    first, lastv = float(losses[0][0]), float(last[0])
    # ---- the same step driven by torch autograd over in-HBM batches (what the trainer ran before the plan: HipBatch forward
    # with autograd, torch ops for the head, the fused retrieval loss, torch.optim.Adam), a few steps, for the ratio
    autograd_ms = None
    try:
        import copy
        from gigl_amd.link_prediction import RetrievalLoss
        from gigl_amd.models import HipBatch
        ref = copy.deepcopy(model).train()
        opt = torch.optim.Adam(ref.parameters(), lr=5e-3, weight_decay=1e-6)
        lossf = RetrievalLoss(temperature=0.07, remove_accidental_hits=True)
        T = 1 + P

        def autograd_step(i):
            roots, cnt, rn = batches[i]
            embs = []
            for r in (roots, rn):
                tree = eng.sample_khop(r, fanouts)
                u = eng.union_build(tree)
                embs.append(ref(HipBatch(eng, tree, u, train=True))[u.root_local[: r.numel()].long()])
            ok = (torch.arange(P, device=dev).view(1, P) < cnt.view(-1, 1)).reshape(-1)
            q_rows = (torch.arange(B, device=dev) * T).repeat_interleave(P)[ok]
            p_rows = (torch.arange(B, device=dev).view(-1, 1) * T + 1 + torch.arange(P, device=dev).view(1, P)).reshape(-1)[ok]
            ids = roots.to(torch.int64) & 0xFFFFFFFF
            cand = torch.cat([embs[0][p_rows], embs[1]])
            scores = embs[0][q_rows] @ cand.T
            loss = lossf.calculate_batch_retrieval_loss(scores, None, ids[q_rows], torch.cat([ids[p_rows], rn.to(torch.int64) & 0xFFFFFFFF]),
                                                        device=dev) / max(int(q_rows.numel()), 1)
            opt.zero_grad()
            loss.backward()
            opt.step()
        with torch.cuda.stream(st):
            for i in range(2):
                autograd_step(i)
            st.synchronize()
            t1 = time.perf_counter()
            for i in range(W, W + 8):
                autograd_step(i)
            st.synchronize()
            autograd_ms = (time.perf_counter() - t1) / 8 * 1e3
    except Exception as exc:  # noqa: BLE001 — a comparison figure only
        print(f"lp train: autograd comparison unavailable ({type(exc).__name__}: {str(exc)[:200]})", file=sys.stderr)
    line = {
        "metric": "sampled+aggregated edges/s (link-prediction training step)", "value": (sampled + agg) * steps_total / elapsed,
        "unit": "edges/s", "n_gpus": 1, "steps": steps_total, "warmup": W, "ms_per_step": elapsed / steps_total * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing": {"repetitions": len(reps), "steps_per_repetition": K, "timed_region_s": round(elapsed, 3),
                   "ms_per_step_median": float(np.median(rep_np) / K * 1e3)},
        "config": {"workload": wl_label + f" N={n} E={eng.n_edges} D={d} fp32, fanout={fanouts}: link-prediction TRAINING step, "
                                        f"{B} anchors x (1 + {P}) rooted trees + {NRN} random negatives per step, GraphSAGE "
                                        f"{d}->{hid}->{emb} L2-normalised, inner-product scores [{B * P} x {B * P + NRN}], "
                                        "retrieval loss (temperature 0.07, same-query + accidental-hit masks), backward of "
                                        "both encodes, Adam(lr 5e-3, wd 1e-6)",
                   "driver": "gigl_nablp_train_plan_step2: ONE library call per step (the next batch's sample + union on a side "
                             "stream beside this step's layers when prefetch is on), replayed as hipGraphs; no torch kernel "
                             "inside a step",
                   "prefetch": prefetch,
                   "sampled_edges_per_step": sampled, "aggregated_edges_per_step": agg,
                   "loss_first_step": first, "loss_last_step": lastv,
                   "autograd_driven_ms_per_step": autograd_ms, "setup_s": round(setup_s, 1)},
        "roofline": None, "cpu_baseline": None,
        "note": "secondary line; the per-kernel picture of a step is the rocprofv3 summary under profiles/ (the plan's launches "
                "run on a private ctx: no per-group HIP-event timers)",
    }
    if not (np.isfinite(first) and np.isfinite(lastv)):
        raise RuntimeError("non-finite training loss")
    emit(line)
    plan.close()
    eng._graph_out = None  # (an alias of the main graph: freed once)
    eng.close()


def run_cpu_train_baseline(eng, model, my, labels, fanouts, W, out_dim):
    """the CPU port of the training step on one host core: oracle sampler + collate (C), fp32 torch forward over the WHOLE
    union graph with autograd (the reference's execution order), cross-entropy on the roots, backward, Adam — full
    batches of the same B roots; counted in the GPU line's unit (sampled + trimmed forward-aggregated edges)"""
    import torch.nn.functional as F

    import oracle
    from oracle import gnn_ref
    rowptr, col = eng.graph_to_host()
    L, B = len(fanouts), int(my.shape[1])
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=0.01, weight_decay=5e-4)
    lab = labels.cpu()
    torch.set_num_threads(1)
    budget_s, t_used, edges, batches = 20.0, 0.0, 0, 0
    i = W
    while t_used < budget_s and batches < 64:
        roots = my[i % my.shape[0]].cpu().numpy().view(np.uint32)
        t0 = time.perf_counter()
        nbr, cnt = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
        u = oracle.union_build(roots, fanouts, nbr)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        t_used += time.perf_counter() - t0
        ids = torch.from_numpy(u["nodes"].astype(np.int64)).to(torch.int32).to(eng.device)
        xs = eng.gather_rows(ids, torch.tensor([ids.numel()], dtype=torch.int32, device=eng.device), int(ids.numel())).cpu()
        t0 = time.perf_counter()
        out = gnn_ref.graphsage_forward(xs, ei, params, L)
        loss = F.cross_entropy(out[torch.from_numpy(u["root_local"].astype(np.int64))],
                               lab[torch.from_numpy(roots.astype(np.int64))])
        opt.zero_grad()
        loss.backward()
        opt.step()
        t_used += time.perf_counter() - t0
        meta, rp = u["meta"], u["rowptr"].astype(np.int64)
        edges += int(sum(int(c.sum()) for c in cnt)) + sum(int(rp[int(meta[2 + (L - 1 - l)])]) for l in range(L))
        batches += 1
        i += 1
    return {"value": edges / t_used, "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": f"{batches} full training batches of {B} roots of the same graph / fanout, {t_used:.1f} s: oracle "
                      "sampler + collate (oracle/gigl_oracle.c, 1 thread), fp32 torch CPU forward over the whole union graph "
                      "with autograd, cross-entropy, backward, Adam (1 thread); edges in the GPU line's unit"}


def run_entry_inferencer(args, rank, world, local_rank):
    """--entry inferencer: the workload's full inference pass (every node a root, batches of B in the TFRecord route's
    order) through the drop-in entry point's own code: Inferencer.infer_resident -> plugin.infer_batch(HbmRootBatch) ->
    ResidentGraph.encode -> gigl_sage_plan_run, rows handed to the exporter (Avro encoded on the device, written out by
    its writer thread).  The graph is built in HBM by this script (ResidentGraph.from_engine) instead of being read from
    preprocessor tables — ingest is one-time work outside the step.  A replica per GPU at N > 1; a secondary line."""
    import shutil
    import tempfile
    from gigl_amd._lib import MODE_FAST, MODE_SPARK_HASH, STATS, STATS_LEN
    from gigl_amd.engine import HipEngine
    from gigl_amd.hbm import ResidentGraph
    from gigl_amd.inferencer import Inferencer, _RowWriter
    from gigl_amd.task_specs import HipGraphSageNodeClassificationSpec

    torch.cuda.set_device(local_rank)
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    B, G = args.batch, max(1, args.group)
    t0 = time.time()
    n, d = build_workload(eng, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    torch.manual_seed(0)
    spec = HipGraphSageNodeClassificationSpec(out_dim=out_dim, hid_dim=hid, num_layers=len(fanouts))
    from gigl_amd.models import GraphSAGE
    spec.model = GraphSAGE(d, hid, out_dim, num_layers=len(fanouts)).to(dev)
    mode = MODE_SPARK_HASH if args.mode == "parity" else MODE_FAST
    resident = ResidentGraph.from_engine(eng, np.arange(n, dtype=np.int64), fanouts, node_type="paper", mode=mode)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    scratch = tempfile.mkdtemp(prefix="gigl_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)

    class _NullWriter:  # rows stay in HBM
        n_rows = 0

        def add(self, ids, emb, pred, ids_dev=None):
            self.n_rows += int(ids.size)

        def close(self):
            pass

    inf = Inferencer()

    def one_pass(sink):
        w = (_RowWriter({"embeddings": os.path.join(scratch, "emb") + "/"}, "paper", keep_on_device=sink == "avro-device")
             if sink != "none" else _NullWriter())
        inf.infer_resident(spec, dev, resident, w, B, groups=G)
        w.close()
        torch.cuda.synchronize()
        return w

    try:
        one_pass("none")  # warm-up: hash table, plan, allocator
        # exact edge counts of the pass (untimed; sampling is deterministic)
        acc = torch.zeros(STATS_LEN, dtype=torch.int64, device=dev)
        n_steps = 0
        ids = resident.inference_root_order()
        for hb in resident.root_batches(ids, B, G):
            plan = resident._plan_for(spec.model, B, G)
            plan.run(hb.roots, sampling_seed=resident.seed, mode=mode)
            plan.stats(hb.roots, acc)
        torch.cuda.synchronize()
        st = acc.cpu().numpy().astype(np.float64)
        n_steps = -(-n // B)
        # the padding batches of the last call (one repeated root each) are part of the pass; their few edges are in `st`
        edges_pass = float(st[STATS["sampled"]] + st[STATS["aggregated"]])
        res, sink_trace = {}, None
        for sink in ("none", "avro-device", "avro-files"):
            if os.environ.get("GIGL_BENCH_PROFILE") == sink:
                import cProfile
                import pstats
                one_pass(sink)
                pr = cProfile.Profile()
                pr.enable()
                one_pass(sink)
                pr.disable()
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(25)
            one_pass(sink)
            shutil.rmtree(os.path.join(scratch, "emb"), ignore_errors=True)
            reps = []
            t_all = time.perf_counter()
            while time.perf_counter() - t_all < args.min_seconds or len(reps) < 3:
                if world > 1:
                    import torch.distributed as dist
                    dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                w = one_pass(sink)
                reps.append(time.perf_counter() - t1)
                if sink == "avro-files":
                    sink_trace = dict(w.exporter.trace, bytes=w.exporter.bytes_written)
                    shutil.rmtree(os.path.join(scratch, "emb"), ignore_errors=True)  # (untimed)
            res[sink] = np.array(reps)
        # plan level on the same roots and call shape, driven directly (no entry-point code, no rows consumed)
        plan = resident._plan_for(spec.model, B, G)
        batches = list(resident.root_batches(ids, B, G))
        out = torch.empty((G * B, out_dim), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for hb in batches:
            plan.run(hb.roots, out=out, sampling_seed=resident.seed, mode=mode)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t1
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    head = res[args.entry_sink]
    t_med = float(np.median(head))
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([t_med], dtype=torch.float64, device=dev)
        all_reduce(tt, dist.ReduceOp.MAX)
        t_med = float(tt.item())
    if rank == 0:
        line = {
            "metric": "sampled+aggregated edges/s", "value": edges_pass * world / t_med, "unit": "edges/s",
            "n_gpus": world, "steps": int(n_steps * len(head)), "warmup": n_steps, "ms_per_step": t_med / n_steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_label + f" N={n} E={eng.n_edges} D={d} fanout={fanouts} B={B} GraphSAGE "
                                            f"{d}->{hid}->{out_dim}: FULL inference pass over every node through "
                                            "Inferencer.infer_resident (drop-in entry point, in-HBM route), sampler "
                                            "mode=" + args.mode,
                       "entry": "gigl_amd.inferencer.Inferencer.infer_resident -> HipGraphSageNodeClassificationSpec."
                                "infer_batch(HbmRootBatch) -> ResidentGraph.encode -> gigl_sage_plan_run",
                       "sink": {"avro-device": "Avro data blocks encoded on the device, left in HBM (outputs resident)",
                                "avro-files": "Avro shards: device-encoded, copied out and written to tmpfs by the "
                                              "exporter's writer thread (PCIe + file inclusive)",
                                "none": "bare rows, left in HBM"}[args.entry_sink],
                       "pcie_and_file_inclusive_pass_s_median": float(np.median(res["avro-files"])),
                       "pcie_and_file_inclusive_roots_per_s": n * world / float(np.median(res["avro-files"])),
                       "batches_per_call": G, "roots_per_s": n * world / t_med,
                       "pass_s_median": t_med, "pass_s_all": [round(float(v), 4) for v in head],
                       "compute_only_pass_s_median": float(np.median(res["none"])),
                       "compute_only_ms_per_step": float(np.median(res["none"])) / n_steps * 1e3,
                       "plan_level_pass_s": plan_s, "plan_level_ms_per_step": plan_s / n_steps * 1e3,
                       "entry_over_plan": t_med / plan_s, "sink_trace_last_pass": sink_trace,
                       "sampled_edges_per_step": float(st[STATS["sampled"]]) / n_steps,
                       "aggregated_edges_per_step": float(st[STATS["aggregated"]]) / n_steps,
                       "setup_s": round(setup_s, 1)},
            "roofline": None, "cpu_baseline": None,
        }
        emit(line)
    resident.close()
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_entry_sampler(args, rank, world, local_rank):
    """--entry sampler: the Subgraph Sampler job's step on the workload's graph — a batch of B roots sampled k hops
    (gigl_sample_khop, parity mode) and encoded as framed RootedNodeNeighborhood TFRecords on the device
    (gigl_records_encode: per-root dedup, hydration from the resident table, proto3 + TFRecord framing with both
    CRC-32C words), SGSPureSparkV1Task.scala:313-820 + TFRecordIO.scala:53-69.  Inputs and outputs resident in HBM (the
    job's device-to-host copy of finished frames is the PCIe-inclusive figure of scripts/micro_records.py).  Calls are
    issued back to back on the engine's stream into one output buffer; a replica per GPU at N > 1; a secondary line."""
    import ctypes as C
    from gigl_amd import _lib
    from gigl_amd.engine import HipEngine

    torch.cuda.set_device(local_rank)
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    B = args.batch
    t0 = time.time()
    n, d = build_workload(eng, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    g = torch.Generator().manual_seed(42)
    perm = torch.randperm(n, generator=g)
    n_batches = max(8, min(64, n // B // max(world, 1)))
    pool = [perm[(rank + world * i) * B:(rank + world * i + 1) * B].to(torch.int32).to(dev) for i in range(n_batches)]
    trees = [eng.alloc_tree(B, fanouts) for _ in range(2)]
    # sizes and content once, through the public entry (also builds the per-row CRC table: one-time, reported)
    t1 = time.time()
    tbl = C.c_void_p()
    _lib.check(eng._lib.gigl_features_row_crc(eng._ctx, eng._feat, C.byref(tbl)), eng._ctx)
    eng._stream.synchronize()
    row_crc_s = time.time() - t1
    sizes, edges_b, nodes_b = [], [], []
    for r in pool:
        tree = eng.sample_khop(r, fanouts, out=trees[0])
        buf, off = eng.encode_records(tree)
        sizes.append(int(buf.numel()))
        edges_b.append(int(sum(int((t_ != -1).sum().item()) for t_ in tree.nbr)))
    from gigl_amd import wire
    head = buf[: int(off[4].item())].cpu().numpy().tobytes()
    n_ok = sum(1 for _ in wire.iter_tfrecords(head))  # (the reader verifies both CRC words of every frame)
    assert n_ok == 4
    cap = max(sizes) + 4096
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rec_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    o = _lib.GiglRecordOpts()
    o.kind, o.trees_per_record, o.tfrecord_frame = _lib.REC_ROOTED_NODE_NEIGHBORHOOD, 1, 1
    o.condensed_node_type = o.condensed_edge_type = 0
    fo = (C.c_int32 * len(fanouts))(*fanouts)

    def step(i, encode=True, sample=True):
        tree = trees[i & 1]
        r = pool[i % n_batches]
        if sample:
            tree.roots = r
            _lib.check(eng._lib.gigl_sample_khop(eng._ctx, eng._graph, C.c_void_p(r.data_ptr()), B, fo, len(fanouts), 42,
                                                 _lib.MODE_SPARK_HASH if args.mode == "parity" else _lib.MODE_FAST,
                                                 C.byref(tree.c_struct)), eng._ctx)
        if encode:
            _lib.check(eng._lib.gigl_records_encode(eng._ctx, C.c_void_p(r.data_ptr()), C.byref(tree.c_struct), eng._feat,
                                                    C.byref(o), B, C.c_void_p(out.data_ptr()), cap,
                                                    C.c_void_p(rec_off.data_ptr()), C.c_void_p(status.data_ptr())),
                       eng._ctx)

    for tr in trees:
        tr.c_struct.hops, tr.c_struct.b = len(fanouts), B
        for k, f in enumerate(fanouts):
            tr.c_struct.fanouts[k] = f
    for i in range(max(4, args.warmup // 8)):
        step(i)
    eng._stream.synchronize()
    assert int(status.item()) == 0
    setup_s = time.time() - t0
    K_rep = max(n_batches, -(-max(1, args.steps // 8) // n_batches) * n_batches)

    def timed(reps, **kw):
        ts = []
        for _ in range(reps):
            if world > 1:
                import torch.distributed as dist
                dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_w = time.perf_counter()
            e0.record(eng._stream)
            for i in range(K_rep):
                step(i, **kw)
            e1.record(eng._stream)
            e1.synchronize()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t_w, e0.elapsed_time(e1) * 1e-3))
        return np.array(ts)

    reps = args.min_reps
    while True:
        full = timed(reps)
        if full[:, 0].sum() >= args.min_seconds or reps >= 4096:
            break
        reps *= 2
    enc_only = timed(max(3, reps // 4), sample=False)  # the encoder's share: HIP events on its stream, same calls
    wall = torch.tensor(full[:, 0], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce(wall, torch.distributed.ReduceOp.MAX)
    wall = wall.cpu().numpy()
    elapsed = float(wall.sum())
    steps_total = reps * K_rep
    sampled_per_step = float(np.mean(edges_b))
    bytes_per_step = float(np.mean(sizes))
    # algorithmic bytes of the encoder per call: the finished record bytes written + 4*D read per
    # DISTINCT node of every record (what the payloads are copied from) + the tree slots read once
    slots = 1 + sum(int(np.prod(fanouts[:k + 1])) for k in range(len(fanouts)))
    # node fields of a step, from the record sizes: bytes = fields * (4 D + ~10 header bytes) + edges * ~12.5 + ~30 / record
    fields_per_step = max(0.0, (bytes_per_step - 12.5 * sampled_per_step - 30.0 * B) / (4 * d + 10))
    enc_ms = float(np.median(enc_only[:, 1])) / K_rep * 1e3
    alg_bytes = bytes_per_step + min(fields_per_step, B * slots) * 4 * d + 4.0 * slots * B
    achieved = alg_bytes / (enc_ms * 1e-3) / 1e9
    # HBM traffic of one encode call from the committed counter summary (scripts/pmc_records.sh: FETCH_SIZE / WRITE_SIZE
    # in separate rocprofv3 passes over the same call shape — products-shaped graph, [25,10], 4,096 records): KB units;
    # fetches of 16-byte-per-lane reads are tallied at half their bytes on gfx950 (MI355X_MICROARCH.md), hence x2
    traffic, traffic_src = None, None
    if wl_name == "products" and B == 4096 and fanouts == [25, 10]:
        import glob
        for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_encoder_pmc.json")))[::-1]:
            try:
                c = json.load(open(f))
                traffic = sum(1024.0 * c[k + ".WRITE_SIZE"]["mean"] + 2048.0 * c[k + ".FETCH_SIZE"]["mean"]
                              for k in ("record_plan", "record_write"))
                traffic_src = os.path.basename(f)
                break
            except Exception:  # noqa: BLE001 — another layout: no traffic figure
                continue
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_records_baseline(eng, pool[0], fanouts, d)
    if rank == 0:
        q = lambda a, p: float(np.percentile(a, p))
        ms_rep = wall / K_rep * 1e3
        line = {
            "metric": "sampled edges/s (sampler job step: sample + encode records)",
            "value": sampled_per_step * steps_total * world / elapsed, "unit": "edges/s", "n_gpus": world,
            "steps": steps_total, "warmup": max(4, args.warmup // 8), "ms_per_step": elapsed / steps_total * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "timing": {"repetitions": reps, "steps_per_repetition": K_rep, "timed_region_s": round(elapsed, 3),
                       "ms_per_step_median": q(ms_rep, 50), "ms_per_step_p10": q(ms_rep, 10),
                       "ms_per_step_p90": q(ms_rep, 90)},
            "config": {"workload": wl_label + f" N={n} E={eng.n_edges} D={d} fp32 features, fanout={fanouts}, B={B} roots "
                                            "per step: k-hop sample (sampler mode=" + args.mode + ") + framed "
                                            "RootedNodeNeighborhood TFRecords encoded on the device, records left in HBM",
                       "entry": "gigl_sample_khop + gigl_records_encode (what SubgraphSampler.run issues per batch)",
                       "records_per_s": B * steps_total * world / elapsed,
                       "record_bytes_per_s": bytes_per_step * steps_total * world / elapsed,
                       "bytes_per_record": bytes_per_step / B, "sampled_edges_per_step": sampled_per_step,
                       "encode_only_ms_per_step": enc_ms, "row_crc_table_build_s": round(row_crc_s, 4),
                       "setup_s": round(setup_s, 1)},
            "roofline": {"bound": "hbm", "kernel": "gigl_records_encode (record_plan + record_scan + record_write)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None if traffic is None else round(traffic),
                         "traffic_source": traffic_src,
                         "alg_bytes_per_launch": round(alg_bytes), "avg_launch_us": round(enc_ms * 1e3, 1),
                         "launches": int(max(3, reps // 4) * K_rep),
                         "timing": "HIP events on the engine's stream around back-to-back encode calls (no sampling "
                                   "in between), median over repetitions",
                         "node_fields_per_step": round(fields_per_step),
                         "bytes": "record bytes written + 4*D read per node field (fields estimated from the record "
                                  "sizes) + 4 B per tree slot read"},
            "cpu_baseline": cpu_baseline,
        }
        emit(line)
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_cpu_records_baseline(eng, roots, fanouts, d, budget_s=15.0):
    """the oracle's sampler + its restatement of the job's output stage (oracle/records.py: per-root assembly, proto3
    encoding, TFRecord framing with CRC-32C — numpy / pure Python, one core) on a bounded sample of the same roots"""
    import oracle
    from oracle import records as R
    rowptr, col = eng.graph_to_host()
    r_np = roots.cpu().numpy().view(np.uint32)

    class _Rows:  # feature rows of the sampled nodes on demand (the table stays in HBM: 1 GB; untimed fetches)
        def __init__(self):
            self.cache = {}

        def prefetch(self, ids):
            ids = np.unique(np.asarray(ids, dtype=np.int64))
            t = torch.from_numpy(ids).to(torch.int32).to(eng.device)
            n_dev = torch.tensor([t.numel()], dtype=torch.int32, device=eng.device)
            rows = eng.gather_rows(t, n_dev, int(t.numel())).cpu().numpy()
            self.cache = {int(i): rows[k] for k, i in enumerate(ids.tolist())}

        def __getitem__(self, v):
            return self.cache[int(v)]
    feats = _Rows()
    done, edges, used = 0, 0, 0.0
    chunk = 16
    while used < budget_s and done < r_np.size:
        rr = r_np[done:done + chunk]
        t1 = time.perf_counter()
        nbr, _cnt = oracle.sample_khop(rowptr, col, rr, fanouts, canonical=True)
        trees_ = R.tree_edges(rr, fanouts, nbr)
        used += time.perf_counter() - t1
        feats.prefetch(np.concatenate([rr.astype(np.int64)] + [s_ for s_, _ in trees_]))
        t1 = time.perf_counter()
        for root, (s_, d_) in zip(rr.tolist(), trees_):
            R.tfrecord_frame(R.rooted_node_neighborhood_record(root, s_, d_, feats, 0, 0))
            edges += int(s_.size)
        used += time.perf_counter() - t1
        done += rr.size
    dt = used
    return {"value": edges / dt, "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": f"{done} roots of the same batch in {dt:.1f} s: oracle/gigl_oracle.c sampler (1 thread) + "
                      "oracle/records.py assembly, proto3 encoding and TFRecord framing (numpy / pure Python, 1 thread); "
                      f"{done / dt:.1f} records/s"}


def run_typed(args, rank, world, local_rank):
    """--workload typed-dblp (the SamplingOp-DAG sampler + HGT over typed graphs): a DBLP-shaped typed
    graph resident in HBM (2 M authors x 64 floats, 4 M papers x 128 floats, 40 M writes / written_by edges, skewed
    authors), a step = one batch of B paper roots through the one-call typed plan (gigl_typed_plan_*: the DAG
    [authors of the paper: f0] -> [papers of those authors: f1], the distinct nodes per type, the distinct edges per edge
    type) + feature rows + a 2-layer HGT (hidden 64, heads 2; the last layer on the roots only) -> the roots' rows.
    Edges: sampled = the ops' sampled neighbours; aggregated = the edges the two HGT layers reduce over (all distinct
    edges of the batch graph, then those into the roots).  A replica per GPU at N > 1; a secondary line."""
    from gigl_amd.graphdb_sampler import INCOMING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG
    from gigl_amd.models_hetero import HGT

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    f0, f1 = [int(v) for v in args.fanouts.split(",")][:2]
    B = args.batch
    na, npp, ne = (20_000, 40_000, 400_000) if args.small else (2_000_000, 4_000_000, 40_000_000)
    t0 = time.time()
    rng = np.random.default_rng(0)
    a2p, p2a = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
    src = (na * rng.random(ne) ** 2).astype(np.int64)  # skewed authors
    dst = rng.integers(0, npp, ne)
    edges = {a2p: (src.astype(np.uint32), dst.astype(np.uint32)), p2a: (dst.astype(np.uint32), src.astype(np.uint32))}
    feats = {"author": rng.standard_normal((na, 64)).astype(np.float32), "paper": rng.standard_normal((npp, 128)).astype(np.float32)}
    smp = HipGraphDBSampler({"author": 0, "paper": 1}, {"author": na, "paper": npp}, edges, {a2p: 0, p2a: 1}, feats,
                            device=local_rank)
    eng = smp.engine
    ops = [SamplingOp("h1", a2p, f0, [], INCOMING), SamplingOp("h2", p2a, f1, ["h1"], INCOMING)]
    dag = SamplingOpDAG.from_ops(ops)
    torch.manual_seed(0)
    ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
    model = HGT({"author": 64, "paper": 128}, {e: 0 for e in ets}, hid_dim=64, out_dim=64, num_layers=2, num_heads=2).to(dev).eval()
    model.engine = eng
    n_batches = 16
    g = torch.Generator().manual_seed(42)
    pool = [torch.randperm(npp, generator=g)[:B].numpy().astype(np.int64) for _ in range(n_batches)]
    setup_s = time.time() - t0

    def step(i):
        graph, ri, _ = smp.batch_graph_plan(pool[i % n_batches], "paper", dag, b_max=B,
                                            edge_type_ids=model.convs[0].edge_types_map)
        with torch.no_grad():
            return graph, ri, model(graph, ["paper"], row_subset={"paper": ri})["paper"]

    # the step as ONE library call (gigl_hgt_infer_*: plan -> typed batch graph at capacity prefixes -> HGT over composed
    # weights -> the roots' rows; replayed as a hipGraph) — what the typed in-HBM inference route runs for HGT encoders;
    # GIGL_BENCH_TYPED_STAGED=1 keeps the staged launches from Python
    one_call = None
    if not os.environ.get("GIGL_BENCH_TYPED_STAGED"):
        from gigl_amd.models_hetero import HgtInferPlan
        one_call = HgtInferPlan(model, smp, "paper", dag, B)
        if os.environ.get("GIGL_BENCH_NO_GRAPH"):
            one_call.use_graph(False)
        roots_dev = [torch.from_numpy(p_.astype(np.uint32).view(np.int32)).to(dev) for p_ in pool]
        for i in (0, 0, 1):  # (eager, captured, replayed) — and the same rows as the staged forward
            got = one_call.run(roots_dev[i])
            smp.engine.synchronize()
            want = step(i)[2]
            torch.cuda.synchronize()
            assert torch.allclose(got, want, rtol=1e-4, atol=1e-4), float((got - want).abs().max())

    def run_pass():
        """the pool's batches as the typed in-HBM inference route runs them (Inferencer._typed_run_hbm): batch i+1's
        sampling is enqueued before the model over batch i is launched"""
        if one_call is not None:  # (batch i + 1's graph part is announced: it is built under batch i's layers)
            for i in range(n_batches):
                one_call.run(roots_dev[i], roots_dev[i + 1] if i + 1 < n_batches else None)
            return
        issue = lambda i: smp.batch_graph_plan_issue(pool[i % n_batches], "paper", dag, b_max=B,
                                                     edge_type_ids=model.convs[0].edge_types_map)
        tk = issue(0)
        for i in range(n_batches):
            graph, ri, _ = smp.batch_graph_plan_finish(tk)
            tk = issue(i + 1) if i + 1 < n_batches else None
            with torch.no_grad():
                model(graph, ["paper"], row_subset={"paper": ri})

    # exact edge counts of every batch of the pool (the same batches are timed)
    sampled, agg = [], []
    for i in range(n_batches):
        graph, ri, out = step(i)
        res = smp.run_dag(torch.from_numpy(pool[i]).to(torch.int32), dag)
        sampled.append(sum(int(r.cnt.sum().item()) for r in res.values()))
        e_all = sum(int(v.shape[1]) for v in graph.edge_index_dict.values())
        is_root = torch.zeros(int(graph.x_dict["paper"].shape[0]), dtype=torch.bool, device=dev)
        is_root[ri] = True
        e_root = sum(int(is_root[v[1]].sum().item()) for k, v in graph.edge_index_dict.items() if k[2] == "paper")
        agg.append(e_all + e_root)
        assert bool(torch.isfinite(out).all()) and out.shape[0] == B
    torch.cuda.synchronize()
    reps, rep_s = 0, []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < args.min_seconds or reps < args.min_reps:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_pass()
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t1)
        reps += 1
    elapsed = float(sum(rep_s))
    steps = reps * n_batches
    # per-kernel times of one more pass (the library's HIP-event timers; the typed graphs' segmented reduce
    # gigl_hgt_aggregate is timed as gather_mean, the projections as linear, the ops of the DAG as expand)
    names = ["expand", "gather_mean", "linear"]
    eng.profile_enable(names, capacity=n_batches * 256)
    eng.profile_reset()
    run_pass()
    torch.cuda.synchronize()
    prof = {k: eng.profile_read(k) for k in names}
    eng.profile_enable([], 0)
    by_kernel = {k: {"ms_per_step": round(v[0] / n_batches, 5), "launches": int(v[1])} for k, v in prof.items() if v[0] > 0}
    step_ms = elapsed / steps * 1e3
    Fo, H = 64, 2
    # hgt_aggregate per edge: one k row + one v row of Fo floats; per destination: its q row and its output row
    graph, ri, _ = step(0)
    n_dst_all = sum(int(x.shape[0]) for x in graph.x_dict.values())
    b_agg = (float(np.mean(agg)) * (2 * Fo * 4 + 8) + (n_dst_all + B) * 2 * Fo * 4)
    roofline = None
    if "gather_mean" in by_kernel:
        gm = by_kernel["gather_mean"]
        ach = b_agg / (gm["ms_per_step"] * 1e-3) / 1e9
        gm.update(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
        dom = max(by_kernel, key=lambda k: by_kernel[k]["ms_per_step"])
        roofline = {"bound": "hbm", "kernel": "gigl_hgt_aggregate (timed as gather_mean)", "achieved": gm["achieved"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gm["frac"], "traffic": None, "dominant": dom,
                    "alg_bytes_per_launch": round(b_agg * n_batches / max(gm["launches"], 1)),
                    "avg_launch_us": round(gm["ms_per_step"] * n_batches / max(gm["launches"], 1) * 1e3, 2),
                    "launches": gm["launches"],
                    "library_kernel_share_of_step": round(sum(v["ms_per_step"] for v in by_kernel.values()) / step_ms, 3),
                    "note": ("one library call per step replayed as a hipGraph: bound by its kernels (the typed aggregate, the "
                             "per-type projections, the plan's sorts: launch latency at ~10^5 keys), not by the host"
                             if one_call is not None else
                             "the step is bound by the host issuing its ~150 small launches (typed projections per "
                             "node / edge type over composed weights, the plan's sorts), not by a kernel"),
                    "timing": "HIP events on the engine's stream over one untimed pass of the timed batches",
                    "by_kernel": by_kernel}
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_typed_baseline(edges, feats, ops, model, pool[0], B)
    line = {
        "metric": "sampled+aggregated edges/s", "value": (float(np.mean(sampled)) + float(np.mean(agg))) * steps / elapsed,
        "unit": "edges/s", "n_gpus": 1, "steps": steps, "warmup": n_batches, "ms_per_step": step_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing": {"repetitions": reps, "steps_per_repetition": n_batches, "timed_region_s": round(elapsed, 3)},
        "config": {"workload": f"DBLP-shaped typed graph ({na} authors x 64, {npp} papers x 128, {ne} edges per edge type), "
                               f"SamplingOp DAG [{f0},{f1}] over {B} paper roots per step through the one-call typed plan + "
                               "2-layer HGT (hidden 64, heads 2, last layer on the roots)",
                   "entry": ("models_hetero.HgtInferPlan.run (gigl_hgt_infer_run: gigl_typed_plan_run + merged CSR at capacity "
                             "prefixes + HGT over composed weights, one library call per step: two captured parts, the next batch's graph part "
                             "under this batch's layers)"
                             if one_call is not None else
                             "HipGraphDBSampler.batch_graph_plan_issue / _finish (gigl_typed_plan_run + gigl_typed_plan_merged_csr; "
                             "batch i+1 enqueued before the model over batch i) -> HGT.forward(row_subset) over composed weights"),
                   "roots_per_s": B * steps / elapsed, "sampled_edges_per_step": float(np.mean(sampled)),
                   "aggregated_edges_per_step": float(np.mean(agg)),
                   "distinct_nodes_per_step": n_dst_all, "setup_s": round(setup_s, 1)},
        "roofline": roofline, "cpu_baseline": cpu_baseline,
    }
    if world > 1:
        import torch.distributed as dist
        v = torch.tensor([line["value"]], dtype=torch.float64, device=dev)
        t = torch.tensor([line["ms_per_step"]], dtype=torch.float64, device=dev)
        all_reduce(v, dist.ReduceOp.SUM)
        all_reduce(t, dist.ReduceOp.MAX)
        line.update(value=float(v.item()), ms_per_step=float(t.item()), n_gpus=world)
    if rank == 0:
        emit(line)
    if one_call is not None:
        one_call.close()
    smp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_cpu_typed_baseline(edges, feats, ops, model, roots, B, budget_s=15.0):
    """the CPU restatement of the typed step on a bounded sample of the same roots: oracle/dag_sampler.py (the per-root
    GraphDBSampler restatement, pure Python) for the DAG, the union of the samples as the batch graph, fp32 torch CPU
    HGT (oracle/gnn_ref.hgt_conv, one thread) over it; edges counted in the GPU line's unit"""
    import torch.nn.functional as F
    from oracle import dag_sampler, gnn_ref
    torch.set_num_threads(1)
    nbrs = dag_sampler.neighbour_lists(edges)
    node_types = {"author": 0, "paper": 1}
    cet = {et: i for i, et in enumerate(edges)}
    by_c = {v: k for k, v in node_types.items()}
    et_of = {i: (et.src_node_type, et.relation, et.dst_node_type) for et, i in cet.items()}
    ets = list(et_of.values())
    from gigl_amd.models_hetero import HGT
    cpu = HGT({"author": 64, "paper": 128}, {e: 0 for e in ets}, hid_dim=64, out_dim=64, num_layers=2, num_heads=2)
    cpu.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    done, units, used = 0, 0, 0.0
    chunk = B  # (the GPU line's batch: the union graph, and with it the work per root, depends on the batch size)
    while used < budget_s and done < min(len(roots), B):
        rr = roots[done:done + chunk]
        t1 = time.perf_counter()
        e_all, n_all = set(), set()
        for r in rr.tolist():
            e_, n_ = dag_sampler.sample_for_root(int(r), ops, nbrs, node_types, cet, "paper")
            e_all |= e_
            n_all |= n_
        ids = {t: np.array(sorted(v for v, c in n_all if by_c[c] == t), dtype=np.int64) for t in node_types}
        pos = {t: {int(v): i for i, v in enumerate(ids[t].tolist())} for t in node_types}
        ei = {}
        for c, triple in et_of.items():
            pr = [(pos[triple[0]][s_], pos[triple[2]][d_]) for s_, d_, cc in e_all if cc == c]
            ei[triple] = torch.tensor(pr, dtype=torch.int64).t().reshape(2, -1)
        xd = {t: torch.from_numpy(feats[t][ids[t]]) for t in node_types if ids[t].size}
        with torch.no_grad():
            h = {t: torch.relu(F.linear(x, cpu.lin_dict[t].weight, cpu.lin_dict[t].bias)) for t, x in xd.items()}
            for conv in cpu.convs:
                pr = dict(kqv={t: (conv.kqv_lin.lins[t].weight, conv.kqv_lin.lins[t].bias) for t in xd},
                          out={t: (conv.out_lin.lins[t].weight, conv.out_lin.lins[t].bias) for t in xd},
                          k_rel=conv.k_rel.weight, v_rel=conv.v_rel.weight, skip={t: conv.skip[t] for t in xd},
                          p_rel={e: conv.p_rel["__".join(e)] for e in ets}, edge_types=ets)
                h = gnn_ref.hgt_conv(h, {k: v for k, v in ei.items() if v.numel()}, pr, 2)
            F.linear(h["paper"], cpu.lin.weight, cpu.lin.bias)
        used += time.perf_counter() - t1
        root_set = set(int(v) for v in rr.tolist())
        units += len(e_all) + len(e_all) + sum(1 for s_, d_, c in e_all if et_of[c][2] == "paper" and d_ in root_set)
        done += len(rr)
    return {"value": units / max(used, 1e-9), "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": f"{done} roots of one batch in {used:.1f} s: oracle/dag_sampler.py (pure Python, per root) + fp32 torch "
                      "CPU HGT over the union of the samples (oracle/gnn_ref.hgt_conv, both layers over the whole graph: "
                      "the reference's execution order), 1 thread; edges in the GPU line's unit (distinct sampled edges + "
                      f"the edges the trimmed layers reduce over); {done / max(used, 1e-9):.1f} roots/s"}


def _lib_stats_len():
    from gigl_amd._lib import STATS_LEN
    return STATS_LEN


def gat_lp_world(args, local_rank, want_out_degree=False):
    """the gat-lp workload in HBM: graph (CSR by destination + CSR by source: the positives' graph), the fp16 table and
    a 2-layer GAT; -> dict of the names run_gat_lp / run_gat_lp_train use"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models_attn import GAT
    torch.cuda.set_device(local_rank)
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    L = len(fanouts)
    B, n_neg = args.batch, 512
    scale = args.shard_scale if 0.0 < args.shard_scale < 1.0 else 0.125
    n = int(244_160_499 * scale)
    e_total = int(1_728_364_232 * scale)
    d, hid, out_dim, heads = 768, 128, 128, 2
    t0 = time.time()
    bits = int(np.ceil(np.log2(n)))
    parts = []
    for ci, c0 in enumerate(range(0, e_total, 1 << 27)):
        a_, b_ = rmat_edges_gpu(bits, min(1 << 27, e_total - c0), seed=3 + 7919 * ci, device=dev)
        parts.append((((a_ * 0x9E3779B1) % n).to(torch.int32), ((b_ * 0x9E3779B1) % n).to(torch.int32)))
    src, dst = torch.cat([q[0] for q in parts]), torch.cat([q[1] for q in parts])
    del parts
    eng.build_from_coo(n, src, dst, is_directed=True)
    eng.build_from_coo(n, dst, src, is_directed=True, out_graph=True)  # CSR by source: the positives' graph
    has_out = (torch.bincount(src.long(), minlength=n) > 0) if want_out_degree else None
    del src, dst
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    x = torch.empty((n, d), device=dev, dtype=torch.float16)
    step_rows = max(1, (1 << 28) // d)
    for i in range(0, n, step_rows):
        x[i:i + step_rows] = torch.randn((min(step_rows, n - i), d), generator=g, device=dev).to(torch.float16)
    eng.load_features(x)
    del x
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    model = GAT(d, hid, out_dim, num_layers=L, heads=heads).to(dev)
    if os.environ.get("GIGL_BENCH_GAT_FIRST_LAYER"):  # (A/B knob: "fused" | "0" = projection first)
        v = os.environ["GIGL_BENCH_GAT_FIRST_LAYER"]
        model.input_side_first_layer = False if v == "0" else v
    return dict(eng=eng, dev=dev, fanouts=fanouts, L=L, B=B, n_neg=n_neg, scale=scale, n=n, d=d, hid=hid, out_dim=out_dim,
                heads=heads, t0=t0, model=model, has_out=has_out)


def run_gat_lp_train(args, rank, world, local_rank):
    """--workload gat-lp --train: the link-prediction TRAINING step of the GAT encoder on the in-HBM route, as
    HipNodeAnchorLinkPredictionSpec.train issues it (node_anchor_based_link_prediction_modeling_task_spec.py:334-451):
    per step, B anchors + one sampled positive each (ResidentGraph.nablp_batches) and 512 random negatives are sampled
    and united in HBM, the batch graphs are handed to the encoder as device-built GraphData (ResidentGraph.graph_data:
    the GAT layers' autograd functions run HIP forward AND backward kernels), inner-product scores + the fused
    retrieval loss (nablp_spec._infer_task_inputs_hbm + Retrieval), backward, Adam (lr 5e-3, weight decay 1e-6: the
    spec's defaults).  Launches are driven by torch autograd from Python, one batch per step, one stream — the step is
    NOT a library plan (the node-classification step is: --train); a secondary line."""
    from gigl_amd._lib import GIGL_META_LEVEL0
    from gigl_amd.hbm import HbmTrainBatch, ResidentGraph
    from gigl_amd.link_prediction import DecoderType, LinkPredictionDecoder, LinkPredictionGNN
    from gigl_amd.nablp_spec import NodeAnchorBasedLinkPredictionTasks, Retrieval, _infer_task_inputs_hbm

    w_ = gat_lp_world(args, local_rank, want_out_degree=True)
    eng, dev, fanouts, L, B, n_neg, scale, n = (w_[k] for k in ("eng", "dev", "fanouts", "L", "B", "n_neg", "scale", "n"))
    d, hid, out_dim, heads, t0, enc = (w_[k] for k in ("d", "hid", "out_dim", "heads", "t0", "model"))
    model = LinkPredictionGNN(encoder=enc, decoder=LinkPredictionDecoder(DecoderType.inner_product)).to(dev)
    model.encoder.engine = eng
    model.decoder.engine = eng
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=1e-6)
    tasks = NodeAnchorBasedLinkPredictionTasks()
    tasks.add_task(Retrieval(temperature=0.07, remove_accidental_hits=True), weight=1.0)
    K, W = max(args.steps if args.steps != 960 else 64, 8), max(min(args.warmup, 8), 2)
    pool = K + W + 8
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    # anchors with at least one out-edge (the main samples of the reference's job are positive-edge endpoints)
    cand = torch.nonzero(w_["has_out"]).view(-1)
    pick = torch.randint(0, cand.numel(), (pool * B,), generator=gp).to(dev)
    anchors = cand[pick].cpu().numpy().astype(np.int64)
    del cand, pick, w_["has_out"]
    negs_cpu = torch.randint(0, n, (pool, n_neg), generator=gp)
    negs_host = negs_cpu.numpy().astype(np.int64)  # (the ids as the random-negative stream hands them out: host arrays)
    negs = negs_cpu.to(torch.int32).to(dev)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=dev)
    eng.bind_stream(st)
    resident = ResidentGraph.from_engine(eng, np.arange(n, dtype=np.int64), fanouts)
    resident.train_as_graph_data = True  # (GAT trains over a PyG-shaped batch: hbm.encoder_trains_over_hip_batches)
    resident.defer_x = True              # (... whose first layer reads the stored rows in place: no dense x per batch)
    setup_s = time.time() - t0
    main_it = resident.nablp_batches(anchors, np.ones(anchors.size, dtype=np.int64), B, 1, loop=True)

    def step(i):
        with torch.cuda.stream(st):
            mb = next(main_it)
            g, ri = resident.train_graph(negs[i % pool])
            rb = HbmTrainBatch(graph=g, root_node_indices=ri, root_node_labels=None, root_ids=negs_host[i % pool])
            opt.zero_grad(set_to_none=True)
            ti = _infer_task_inputs_hbm(model, mb, rb, False, dev)
            loss, _ = tasks.calculate_losses(ti, None, should_eval=False, device=dev)
            loss.backward()
            opt.step()
        return loss.detach()

    hist = [step(i) for i in range(W)]
    st.synchronize()
    # edges per step, counted on the device over untimed batches of the same shape (sampled + consumed by the FORWARD
    # attention reductions of both encodes, like the inference line)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    lvl = [GIGL_META_LEVEL0 + (L - 1 - l) for l in range(L)]
    n_count = 8
    with torch.cuda.stream(st), torch.no_grad():
        for i in range(n_count):
            a = torch.from_numpy(anchors[i * B:(i + 1) * B].astype(np.uint32).view(np.int32)).to(dev)
            pos, cnt = eng.sample_positives(a, 1)
            for roots in (torch.cat([a.view(-1, 1), pos.view(-1, 1)], dim=1).reshape(-1).contiguous(), negs[i]):
                tree = eng.sample_khop(roots, fanouts)
                u = eng.union_build(tree)
                rowlen = (u.rowend - u.rowptr).to(torch.int64)
                ar = torch.arange(rowlen.numel(), device=dev)
                agg = sum((rowlen * (ar < u.meta[j])).sum() for j in lvl)
                acc.add_(torch.stack([sum(c.sum() for c in tree.cnt).to(torch.int64), agg.to(torch.int64)]))
    st.synchronize()
    per_step = acc.cpu().numpy().astype(np.float64) / n_count
    rep_s, steps, i = [], 0, W
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < args.min_seconds or len(rep_s) < min(args.min_reps, 3):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(K):
            hist.append(step(i))
            i += 1
        torch.cuda.synchronize(dev)
        rep_s.append(time.perf_counter() - t1)
        steps += K
    elapsed = float(sum(rep_s))
    losses = torch.stack(hist).cpu().numpy().astype(np.float64)
    assert np.isfinite(losses).all(), "the training loss went non-finite"
    # where the step's library kernel time goes: HIP-event timers over a few untimed steps (eager launches on one stream)
    names = ["expand", "union_insert", "union_relax", "union_nodes", "union_edge_sort", "union_csr", "gather_mean",
             "gather_bwd", "linear"]
    n_prof = 8
    eng.profile_enable(names, capacity=4096)
    eng.profile_reset()
    for _ in range(n_prof):
        step(i)
        i += 1
    st.synchronize()
    prof = {k: eng.profile_read(k) for k in names}
    eng.profile_enable([], 0)
    by_kernel = {k: {"ms_per_step": round(v[0] / n_prof, 5), "launches_per_step": round(v[1] / n_prof, 1)}
                 for k, v in prof.items() if v[0] > 0}
    lib_ms = sum(v["ms_per_step"] for v in by_kernel.values())
    step_ms = elapsed / steps * 1e3
    ms_rep = np.array(rep_s) / K * 1e3
    q_ = lambda a, p: float(np.percentile(a, p))
    line = {
        "metric": "sampled+aggregated edges/s", "value": float(per_step.sum()) * steps / elapsed, "unit": "edges/s",
        "n_gpus": 1, "steps": steps, "warmup": W, "ms_per_step": step_ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing": {"repetitions": len(rep_s), "steps_per_repetition": K, "timed_region_s": round(elapsed, 3),
                   "ms_per_step_median": q_(ms_rep, 50), "ms_per_step_p10": q_(ms_rep, 10), "ms_per_step_p90": q_(ms_rep, 90)},
        "config": {"workload": f"MAG240M-shaped RMAT x{scale:g} (N={n}, E={eng.n_edges} directed, D={d} fp16), link-prediction "
                               f"TRAINING step: {B} anchors + 1 positive each + {n_neg} random negatives, fanout={fanouts}, "
                               f"2-layer GAT heads={heads} hid={hid} out={out_dim}, inner-product scores + fused retrieval "
                               "loss, backward, Adam",
                   "entry": "ResidentGraph.nablp_batches / train_graph -> nablp_spec._infer_task_inputs_hbm -> Retrieval -> "
                            "backward -> Adam: the step of HipNodeAnchorLinkPredictionSpec.train on the in-HBM route",
                   "driver": "torch autograd from Python, one batch per step, one stream, eager launches",
                   "sampled_edges_per_step": float(per_step[0]), "aggregated_edges_per_step": float(per_step[1]),
                   "loss_first": float(losses[0]), "loss_last_mean": float(losses[-8:].mean()),
                   "setup_s": round(setup_s, 1)},
        "roofline": {"bound": "latency", "kernel": max(by_kernel, key=lambda k: by_kernel[k]["ms_per_step"]) if by_kernel else None,
                     "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                     "library_kernel_ms_per_step": round(lib_ms, 5), "step_ms": round(step_ms, 5),
                     "library_kernel_share_of_step": round(lib_ms / step_ms, 4),
                     "note": "launches driven by torch autograd from Python on one stream, one host read per batch graph "
                             "(its node / edge counts): the step is bound by the host between kernels where "
                             "library_kernel_share_of_step is well under 1",
                     "by_kernel": by_kernel},
        "cpu_baseline": None,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        torch.cuda.synchronize(dev)
        eng.bind_stream(torch.cuda.current_stream(dev))  # (the baseline's device helpers run on torch's current stream)
        a_dev = torch.from_numpy(anchors[: 4 * B].astype(np.uint32).view(np.int32)).view(4, B).to(dev)
        line["cpu_baseline"] = run_cpu_gat_lp_baseline(eng, enc, a_dev, negs[:4], fanouts, heads, L, budget_s=20.0, train=True)
    if rank == 0:
        emit(line)
    eng.close()


def run_gat_lp_train_plan(args, rank, world, local_rank):
    """--workload gat-lp --train: the link-prediction TRAINING step of configs[4]'s encoder (two-layer GAT, heads 2, hid 128,
    out 128, over the MAG240M-shaped share with 768-wide fp16 rows) as ONE library call per step
    (gigl_gat_nablp_train_plan_*: sample + union of the main batch — B anchors with one positive each — and of 512 random
    negatives, the GAT forward of both from the input side, inner-product scores, retrieval loss, the backward of both
    encodes, Adam; the next batch's graph part on a side stream; replayed as hipGraphs; no torch kernel inside a step) —
    what HipNodeAnchorLinkPredictionSpec.train runs for this encoder (node_anchor_based_link_prediction_modeling_task_spec.py:
    334-451).  The autograd-driven step over the same kind of batches (round 4's line, --gat-train-autograd) is timed beside
    it for a few steps.  A secondary line."""
    from gigl_amd._lib import GIGL_META_LEVEL0
    from gigl_amd.engine import GatNablpTrainPlan
    from gigl_amd.hbm import HbmTrainBatch, ResidentGraph
    from gigl_amd.link_prediction import DecoderType, LinkPredictionDecoder, LinkPredictionGNN
    from gigl_amd.nablp_spec import NodeAnchorBasedLinkPredictionTasks, Retrieval, _infer_task_inputs_hbm
    import copy

    w_ = gat_lp_world(args, local_rank, want_out_degree=True)
    eng, dev, fanouts, L, B, n_neg, scale, n = (w_[k] for k in ("eng", "dev", "fanouts", "L", "B", "n_neg", "scale", "n"))
    d, hid, out_dim, heads, t0, enc = (w_[k] for k in ("d", "hid", "out_dim", "heads", "t0", "model"))
    K, W = max(args.steps if args.steps != 960 else 64, 8), max(min(args.warmup, 8), 2)
    pool = K + W
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    cand = torch.nonzero(w_["has_out"]).view(-1)  # anchors with at least one out-edge
    pick = torch.randint(0, cand.numel(), (pool * B,), generator=gp).to(dev)
    anchors = cand[pick].to(torch.int32).view(pool, B)
    del cand, pick, w_["has_out"]
    negs = torch.randint(0, n, (pool, n_neg), generator=gp).to(torch.int32).to(dev)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    batches = []
    with torch.cuda.stream(st):
        for i in range(pool):
            pos, cnt = eng.sample_positives(anchors[i], 1, sampling_seed=42)
            a2 = anchors[i].view(-1, 1)
            roots = torch.cat([a2, torch.where(cnt.view(-1, 1) > 0, pos.view(-1, 1), a2)], dim=1).reshape(-1)
            batches.append((roots.contiguous(), cnt.to(torch.int32).contiguous(), negs[i].contiguous()))
    st.synchronize()
    setup_s = time.time() - t0
    ref_model = copy.deepcopy(enc)
    plan = GatNablpTrainPlan(eng, enc, B, 1, n_neg, fanouts, temperature=0.07, remove_accidental_hits=True, lr=5e-3,
                             weight_decay=1e-6)
    prefetch = not args.no_train_prefetch
    nxt = lambda i, hi: (batches[i + 1][0], batches[i + 1][2]) if prefetch and i + 1 < hi else None
    losses = []
    with torch.cuda.stream(st):
        for i in range(W):
            losses.append(plan.step(*batches[i], next_roots=nxt(i, W)).clone())
    st.synchronize()
    # exact edge counts of the timed batches (both encodes), through the separate entry points, untimed
    counts = np.zeros(2, dtype=np.float64)
    n_count = min(K, 8)
    with torch.cuda.stream(st), torch.no_grad():
        for i in range(W, W + n_count):
            for r in (batches[i][0], batches[i][2]):
                tree = eng.sample_khop(r, fanouts)
                u = eng.union_build(tree)
                rowlen = (u.rowend - u.rowptr).to(torch.int64)
                a_ = torch.arange(rowlen.numel(), device=dev)
                agg = sum((rowlen * (a_ < u.meta[GIGL_META_LEVEL0 + (L - 1 - l)])).sum() for l in range(L))
                counts += np.array([float(sum(c.sum() for c in tree.cnt)), float(agg)])
    st.synchronize()
    sampled, agg = counts / n_count
    reps = []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < args.min_seconds or len(reps) < 3:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(st):
            for i in range(W, W + K):
                last = plan.step(*batches[i], next_roots=nxt(i, W + K))
        st.synchronize()
        reps.append(time.perf_counter() - t1)
    rep_np = np.array(reps)
    elapsed, steps_total = float(rep_np.sum()), K * len(reps)
    first, lastv = float(losses[0][0]), float(last[0])
    assert np.isfinite(lastv), "the training loss went non-finite"
    plan.close()
    # ---- the autograd-driven step (torch autograd over device-built batch graphs, torch.optim.Adam), a few steps
    autograd_ms = None
    try:
        model = LinkPredictionGNN(encoder=ref_model, decoder=LinkPredictionDecoder(DecoderType.inner_product)).to(dev)
        model.encoder.engine = eng
        model.decoder.engine = eng
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=1e-6)
        tasks = NodeAnchorBasedLinkPredictionTasks()
        tasks.add_task(Retrieval(temperature=0.07, remove_accidental_hits=True), weight=1.0)
        resident = ResidentGraph.from_engine(eng, np.arange(n, dtype=np.int64), fanouts)
        resident.train_as_graph_data, resident.defer_x = True, True
        a_host = anchors.cpu().numpy().astype(np.int64).reshape(-1)
        main_it = resident.nablp_batches(a_host, np.ones(a_host.size, dtype=np.int64), B, 1, loop=True)
        negs_host = negs.cpu().numpy().astype(np.int64)

        def autograd_step(i):
            with torch.cuda.stream(st):
                mb = next(main_it)
                g, ri = resident.train_graph(negs[i % pool])
                rb = HbmTrainBatch(graph=g, root_node_indices=ri, root_node_labels=None, root_ids=negs_host[i % pool])
                opt.zero_grad(set_to_none=True)
                ti = _infer_task_inputs_hbm(model, mb, rb, False, dev)
                loss, _ = tasks.calculate_losses(ti, None, should_eval=False, device=dev)
                loss.backward()
                opt.step()
        for i in range(3):
            autograd_step(i)
        st.synchronize()
        t1 = time.perf_counter()
        for i in range(3, 3 + 16):
            autograd_step(i)
        st.synchronize()
        autograd_ms = (time.perf_counter() - t1) / 16 * 1e3
    except Exception as exc:  # noqa: BLE001 — a comparison figure only
        print(f"gat-lp train: autograd comparison unavailable ({type(exc).__name__}: {str(exc)[:200]})", file=sys.stderr)
    ms_rep = rep_np / K * 1e3
    q_ = lambda a, p: float(np.percentile(a, p))
    line = {
        "metric": "sampled+aggregated edges/s", "value": float(sampled + agg) * steps_total / elapsed, "unit": "edges/s",
        "n_gpus": 1, "steps": steps_total, "warmup": W, "ms_per_step": elapsed / steps_total * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing": {"repetitions": len(reps), "steps_per_repetition": K, "timed_region_s": round(elapsed, 3),
                   "ms_per_step_median": q_(ms_rep, 50), "ms_per_step_p10": q_(ms_rep, 10), "ms_per_step_p90": q_(ms_rep, 90)},
        "config": {"workload": f"MAG240M-shaped RMAT x{scale:g} (N={n}, E={eng.n_edges} directed, D={d} fp16), link-prediction "
                               f"TRAINING step: {B} anchors + 1 positive each + {n_neg} random negatives, fanout={fanouts}, "
                               f"2-layer GAT heads={heads} hid={hid} out={out_dim}, inner-product scores + retrieval loss "
                               "(temperature 0.07, both masks), backward, Adam(lr 5e-3, wd 1e-6)",
                   "driver": "gigl_gat_nablp_train_plan_* via gigl_nablp_train_plan_step2: ONE library call per step (the next "
                             "batch's sample + union on a side stream when prefetch is on), replayed as hipGraphs; no torch "
                             "kernel inside a step",
                   "prefetch": prefetch, "sampled_edges_per_step": float(sampled), "aggregated_edges_per_step": float(agg),
                   "loss_first_step": first, "loss_last_step": lastv, "autograd_driven_ms_per_step": autograd_ms,
                   "setup_s": round(setup_s, 1)},
        "roofline": None, "cpu_baseline": None,
        "note": "secondary line; the per-kernel picture of a step is the rocprofv3 summary under profiles/ (the plan's launches "
                "run on a private ctx: no per-group HIP-event timers); --gat-train-autograd is round 4's autograd-driven line "
                "with its CPU baseline",
    }
    if rank == 0:
        emit(line)
    eng.close()


def run_gat_lp(args, rank, world, local_rank):
    """BASELINE.json configs[4] / SURVEY.md 8(d) C5 on one GPU's share of the MAG240M-shaped graph (--shard-scale of
    it as a self-contained graph): link-prediction step of the GAT encoder — anchors + one positive each (sampled
    out-neighbour, counter 3) and 512 random negatives go through sample -> union graph -> 2-layer GAT (heads 2, hid
    128, out 128: attention-weighted segmented reduce) -> root embeddings; inner-product scores against positives +
    random negatives and the fused retrieval loss (infer_task_inputs + Retrieval, python/gigl/src/common/
    modeling_task_specs/utils/infer.py, models/layers/task.py:140-205).  The two encodes are GAT one-call plans
    (gigl_gat_plan_create), G steps per call; the decoder and the loss run per step; a secondary line."""
    from gigl_amd._lib import GIGL_META_LEVEL0, STATS
    from gigl_amd.engine import HipEngine
    from gigl_amd.link_prediction import DecoderType, LinkPredictionDecoder, RetrievalLoss
    from gigl_amd.models import HipBatch
    from gigl_amd.models_attn import GAT

    w_ = gat_lp_world(args, local_rank)
    eng, dev, fanouts, L, B, n_neg, scale, n = (w_[k] for k in ("eng", "dev", "fanouts", "L", "B", "n_neg", "scale", "n"))
    d, hid, out_dim, heads, t0, model = (w_[k] for k in ("d", "hid", "out_dim", "heads", "t0", "model"))
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=dev)  # (the resident data was written on torch's default stream)
    eng.bind_stream(st)
    dec = LinkPredictionDecoder(DecoderType.inner_product)
    dec.engine = eng
    loss_fn = RetrievalLoss(temperature=0.07, remove_accidental_hits=True)
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    K, W = max(args.steps, 8), max(args.warmup, 2)
    # calls of G = 64 steps rotate over S streams (own ctx, plans and captured graph each, the resident graph shared) like
    # the headline's: one call's sampler / union run under another's attention reductions
    S_gat = max(1, int(args.streams)) if not (os.environ.get("GIGL_BENCH_GAT_STAGED") or os.environ.get("GIGL_BENCH_NO_GRAPH")
                                              or args.timed_only) else 1
    pool = 64 * S_gat
    anchors = torch.randint(0, n, (pool, B), generator=gp).to(torch.int32).to(dev)
    negs = torch.randint(0, n, (pool, n_neg), generator=gp).to(torch.int32).to(dev)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    lvl = [GIGL_META_LEVEL0 + (L - 1 - l) for l in range(L)]
    setup_s = time.time() - t0

    # The two encodes as one-call plans (GAT.make_plan: sample -> union -> layers -> one row per root in one library
    # call each), G consecutive steps per call like the headline's batches_per_call (a step's batches stay independent:
    # dedup, union and attention never cross a group); GIGL_BENCH_GAT_STAGED=1 keeps the per-stage entry points, one
    # step per call.
    plans, G = None, 1
    if not os.environ.get("GIGL_BENCH_GAT_STAGED"):
        G = max(1, int(os.environ.get("GIGL_BENCH_GAT_GROUPS", "64")))
        while pool % G:
            G -= 1
        plans = (model.make_plan(eng, 2 * B, fanouts, groups=G), model.make_plan(eng, n_neg, fanouts, groups=G))
        stats_acc = torch.zeros(_lib_stats_len(), dtype=torch.int64, device=dev)

    def encode(roots, count):
        tree = eng.sample_khop(roots, fanouts)
        u = eng.union_build(tree)
        emb = model(HipBatch(eng, tree, u))[u.root_local[: roots.numel()].long()]
        if count:
            rowlen = (u.rowend - u.rowptr).to(torch.int64)
            ar = torch.arange(rowlen.numel(), device=dev)
            agg = sum((rowlen * (ar < u.meta[j])).sum() for j in lvl)
            acc.add_(torch.stack([sum(c.sum() for c in tree.cnt).to(torch.int64), agg.to(torch.int64)]))
        return emb

    def steps_of(a, ng, count, eng=eng, plans=plans):
        """G steps: a [G, B] anchors, ng [G, n_neg] random negatives -> the G losses (eng / plans: the slot's)"""
        pos, cnt = eng.sample_positives(a.reshape(-1), 1)
        pos = pos.view(G, B)
        if plans is not None:
            roots = torch.cat([a, pos], dim=1).reshape(-1).contiguous()  # per step: anchors, then their positives
            nroots = ng.reshape(-1).contiguous()
            main = plans[0].run(roots).view(G, 2 * B, -1)
            rn = plans[1].run(nroots).view(G, n_neg, -1)
            if count:
                plans[0].stats(roots, stats_acc)
                plans[1].stats(nroots, stats_acc)
        else:
            main = encode(torch.cat([a[0], pos[0]]), count).unsqueeze(0)  # (INVALID positive: no out-edge)
            rn = encode(ng[0], count).unsqueeze(0)
        if plans is not None and not os.environ.get("GIGL_BENCH_GAT_TAIL_PER_BATCH"):
            # decoder + loss of the G batches: one GEMM launch (batch in grid.y) and one loss pass, bit-identical to the
            # per-batch entry points below (tests/test_gpu_entry_points.py::test_batched_decoder_and_loss)
            scores = eng.linear_batched(main[:, :B], torch.cat([main[:, B:], rn], dim=1))  # [B, G, B + n_neg]
            return eng.retrieval_loss_batched(scores, 0.07, None, a.long().contiguous(),
                                              torch.cat([pos, ng], dim=1).long())
        losses = []
        for g_ in range(G):
            scores = dec(main[g_, :B], torch.cat([main[g_, B:], rn[g_]]))
            losses.append(loss_fn.calculate_batch_retrieval_loss(
                scores, query_ids=a[g_].long(), candidate_ids=torch.cat([pos[g_], ng[g_]]).long()))
        return torch.stack(losses)

    def call(i0, count=False):
        with torch.cuda.stream(st), torch.no_grad():
            return steps_of(anchors[i0:i0 + G], negs[i0:i0 + G], count)

    for i in range(0, max(W, G), G):
        call(i % pool)
    for i0 in range(0, pool, G):
        call(i0, count=True)
    st.synchronize()
    per_step = acc.cpu().numpy().astype(np.float64) / pool
    if plans is not None:
        from gigl_amd._lib import STATS_AGGREGATED, STATS_SAMPLED
        sa = stats_acc.cpu().numpy().astype(np.float64)
        per_step = np.array([sa[STATS_SAMPLED], sa[STATS_AGGREGATED]]) / pool
    if args.timed_only:  # counter-collection child (collect_live_pmc): eager calls only, every kernel counted is the step's
        n_calls = (max(W, G) + G - 1) // G + pool // G
        for _ in range(8):
            for i0 in range(0, pool, G):
                call(i0)
                n_calls += 1
        st.synchronize()
        print(json.dumps({"timed_only": True, "workload": "gat-lp", "batches_per_call": G, "streams": 1,
                          "steps_executed": n_calls * G, "calls_executed": n_calls}))
        eng.close()
        return
    # The shapes are all capacities (counts stay on the device), so a call replays as a HIP graph over static input
    # rows; kept only when a replay reproduces the eager losses bit for bit, otherwise the eager driver stays.
    driver = (f"one-call GAT plans ({G} steps per call: anchors + positives, random negatives) + decoder + fused loss, "
              "one stream" if plans is not None else "per-stage entry points from Python, one stream")
    eager_call = call
    if not os.environ.get("GIGL_BENCH_NO_GRAPH"):
        try:
            a_buf, n_buf = anchors[:G].clone(), negs[:G].clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                with torch.no_grad():
                    loss_buf = steps_of(a_buf, n_buf, False)

            def graph_call(i0, count=False):
                with torch.cuda.stream(st):
                    a_buf.copy_(anchors[i0:i0 + G], non_blocking=True)
                    n_buf.copy_(negs[i0:i0 + G], non_blocking=True)
                    graph.replay()
                return loss_buf

            for i0 in (0, pool - G):
                with torch.cuda.stream(st):
                    want = eager_call(i0).clone()
                    got = graph_call(i0).clone()
                st.synchronize()
                if not torch.equal(want, got):
                    raise RuntimeError(f"replayed losses {got.tolist()[:2]} != eager {want.tolist()[:2]} at pool entry {i0}")
            call = graph_call
            driver = "one HIP graph per call (captured from " + driver.split(" + decoder")[0] + "), replayed over static inputs"
        except Exception as exc:  # noqa: BLE001 — the eager driver is the same path, only slower
            print(f"gat-lp: graph capture unavailable ({type(exc).__name__}: {str(exc)[:200]})", file=sys.stderr)
            call = eager_call
    # ---- the other streams' slots: own ctx (the resident graph and table shared), own plans, own captured graph
    slots = [(call, st)]
    extra_engs = []
    if call is not eager_call and plans is not None:
        for k in range(1, S_gat):
            e_k = HipEngine(local_rank)
            e_k.share_resident(eng)
            st_k = torch.cuda.Stream(device=dev)
            e_k.bind_stream(st_k)
            pl_k = (model.make_plan(e_k, 2 * B, fanouts, groups=G), model.make_plan(e_k, n_neg, fanouts, groups=G))
            a_k, n_k = anchors[:G].clone(), negs[:G].clone()
            with torch.cuda.stream(st_k), torch.no_grad():
                for _ in range(2):
                    dbg = steps_of(a_k, n_k, False, e_k, pl_k)
            st_k.synchronize()
            if os.environ.get("GIGL_BENCH_DEBUG"):
                print("slot", k, "eager", dbg[:3].tolist(), file=sys.stderr)
            g_k = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_k, stream=st_k):
                with torch.no_grad():
                    loss_k = steps_of(a_k, n_k, False, e_k, pl_k)

            def call_k(i0, count=False, a_k=a_k, n_k=n_k, g_k=g_k, st_k=st_k, loss_k=loss_k):
                with torch.cuda.stream(st_k):
                    a_k.copy_(anchors[i0:i0 + G], non_blocking=True)
                    n_k.copy_(negs[i0:i0 + G], non_blocking=True)
                    g_k.replay()
                return loss_k
            with torch.cuda.stream(st_k):
                got = call_k(0).clone()
            st_k.synchronize()
            with torch.cuda.stream(st):
                want = eager_call(0).clone()
            st.synchronize()
            if not torch.equal(want, got):
                raise RuntimeError(f"slot {k}: replayed losses differ from the eager ones: max |diff| "
                                   f"{float((want - got).abs().max())}, {want[:3].tolist()} vs {got[:3].tolist()}")
            slots.append((call_k, st_k))
            extra_engs.append((e_k, pl_k))
        if len(slots) > 1:
            driver += f"; {len(slots)} streams in flight (one call each)"
    rep_s, steps = [], 0
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < args.min_seconds or len(rep_s) < args.min_reps:
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for c, i0 in enumerate(range(0, pool, G)):
            slots[c % len(slots)][0](i0)
        torch.cuda.synchronize(dev)
        rep_s.append(time.perf_counter() - t1)
        steps += pool
    elapsed = float(sum(rep_s))
    ms_rep = np.array(rep_s) / pool * 1e3
    q_ = lambda a, p: float(np.percentile(a, p))
    # ---- roofline of the step's dominant kernel group: one untimed pass of the same calls with the library's HIP-event
    # timers on (eager launches: events cannot sit inside a replayed graph).  The attention reductions — the first
    # layer's one-pass kernel over the stored rows (gat_input_online_kernel) and the second layer's segmented reduce —
    # are timed as `gather_mean`; algorithmic bytes per SURVEY 8(d): layer 0 reads a stored row (D elements) per
    # aggregated edge and writes one fp32 D-wide operand row per head and destination; layer 1 reads an H*C fp32 row per
    # aggregated edge and per destination and writes one.
    roofline = None
    if plans is not None:
        names = ["expand", "union_insert", "union_relax", "union_nodes", "union_edge_sort", "union_csr", "gather_mean", "linear"]
        eng.profile_enable(names, capacity=(pool // G + 2) * 64)
        eng.profile_reset()
        for i0 in range(0, pool, G):
            eager_call(i0)
        st.synchronize()
        for p_ in plans:
            p_.flush_profile()
        prof = {k: eng.profile_read(k) for k in names}
        eng.profile_enable([], 0)
        sa = stats_acc.cpu().numpy().astype(np.float64) / pool  # per step (both encodes), counted on the device above
        agg0, agg1 = sa[STATS["agg_layer0"]], sa[STATS["agg_layer0"] + 1]
        rows0, rows1 = sa[STATS["rows_layer0"]], sa[STATS["rows_layer0"] + 1]
        esz_ = 2  # fp16 table
        alg = {"gather_mean": agg0 * (4 + d * esz_) + rows0 * (8 + heads * d * 4) +
                              agg1 * (4 + heads * hid * 4) + rows1 * (8 + 2 * heads * hid * 4)}
        by_kernel = {k: {"ms_per_step": round(v[0] / pool, 5), "launches": int(v[1])} for k, v in prof.items() if v[0] > 0}
        dominant = max(by_kernel, key=lambda k: by_kernel[k]["ms_per_step"])
        if "gather_mean" in by_kernel:
            gm = by_kernel["gather_mean"]
            gm.update(bound="hbm", achieved=round(alg["gather_mean"] / (gm["ms_per_step"] * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                      unit="GB/s")
            gm["frac"] = round(gm["achieved"] / HBM_PEAK_GBS, 4)
        head_k = "gather_mean" if "gather_mean" in by_kernel else dominant
        hk = by_kernel[head_k]
        launches = max(hk["launches"], 1)
        roofline = {"bound": "hbm", "kernel": "GAT attention reductions (gat_input_online_kernel + gat_gather_fast; timed as gather_mean)",
                    "achieved": hk.get("achieved"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hk.get("frac"),
                    "traffic": None, "dominant": dominant,
                    "alg_bytes_per_launch": round(alg["gather_mean"] * pool / launches),
                    "avg_launch_us": round(hk["ms_per_step"] * pool / launches * 1e3, 2), "launches": launches,
                    "timing": "HIP events on the plans' stream over one untimed eager pass of the timed calls (one stream: "
                              "a kernel's interval is its own)",
                    "share_of_step": round(hk["ms_per_step"] / (elapsed / steps * 1e3), 3), "by_kernel": by_kernel}
    # ---- HBM traffic by the counters: rocprofv3 PMC passes of a child run of this same command (eager calls)
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if roofline is not None and rank == 0 and world == 1 and not args.no_live_pmc and not under_profiler and \
            not os.environ.get("GIGL_BENCH_CHILD"):
        passthrough = ["--workload", "gat-lp", "--batch", str(B), "--fanouts", ",".join(str(f) for f in fanouts),
                       "--shard-scale", str(args.shard_scale)]
        torch.cuda.synchronize()
        doc_, note_ = collect_live_pmc(passthrough, timeout_s=600.0)
        if doc_ is not None and doc_.get("steps_executed"):
            n_exec, n_calls = int(doc_["steps_executed"]), int(doc_.get("calls_executed") or 1)
            step_ms_ = elapsed / steps * 1e3
            by_step, per_k = step_traffic_of(doc_, n_exec, min_calls=n_calls)
            roofline["step"] = {"bound": "hbm", "traffic_bytes_per_step": round(by_step),
                                "achieved": round(by_step / (step_ms_ * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(by_step / (step_ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a child run of this command "
                                          f"({n_exec} steps in {n_calls} eager calls), the library's kernels launched at least "
                                          "once per call",
                                "by_kernel_bytes_per_step": {k: round(v) for k, v in sorted(per_k.items(), key=lambda kv: -kv[1])[:12]}}
            gm_bytes, _ = step_traffic_of(doc_, n_exec, ["gat_input_online_kernel", "gat_gather_fast", "gat_gather_heavy",
                                                         "gat_alpha_fast"], min_calls=n_calls)
            roofline["traffic"] = round(gm_bytes * pool / launches)  # per launch, like alg_bytes_per_launch
            roofline["traffic_frac"] = round(gm_bytes / (hk["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roofline["live_pmc"] = "collected"
        else:
            roofline["live_pmc"] = note_ or "child reported no step count"
    line = {
        "metric": "sampled+aggregated edges/s", "value": float(per_step.sum()) * steps / elapsed, "unit": "edges/s",
        "n_gpus": 1, "steps": steps, "warmup": W, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing": {"repetitions": len(rep_s), "steps_per_repetition": pool, "timed_region_s": round(elapsed, 3),
                   "ms_per_step_median": q_(ms_rep, 50), "ms_per_step_p10": q_(ms_rep, 10), "ms_per_step_p90": q_(ms_rep, 90)},
        "config": {"workload": f"MAG240M-shaped RMAT x{scale:g} (N={n}, E={eng.n_edges} directed, D={d} fp16), link-prediction "
                               f"step: {B} anchors + 1 positive each + {n_neg} random negatives, fanout={fanouts}, 2-layer GAT "
                               f"heads={heads} hid={hid} out={out_dim}, inner-product scores + fused retrieval loss",
                   "sampled_edges_per_step": float(per_step[0]), "aggregated_edges_per_step": float(per_step[1]),
                   "steps_per_call": G, "driver": driver, "setup_s": round(setup_s, 1)},
        "roofline": roofline, "cpu_baseline": None,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        torch.cuda.synchronize(dev)
        eng.bind_stream(torch.cuda.current_stream(dev))  # (the baseline's device helpers run on torch's current stream)
        line["cpu_baseline"] = run_cpu_gat_lp_baseline(eng, model, anchors, negs, fanouts, heads, L)
    if world > 1:  # a replica per GPU: whole-job rate = sum over the ranks, step time = the slowest rank's
        import torch.distributed as dist
        v = torch.tensor([line["value"]], dtype=torch.float64, device=dev)
        t = torch.tensor([line["ms_per_step"]], dtype=torch.float64, device=dev)
        all_reduce(v, dist.ReduceOp.SUM)
        all_reduce(t, dist.ReduceOp.MAX)
        line.update(value=float(v.item()), ms_per_step=float(t.item()), n_gpus=world)
    if rank == 0:
        emit(line)
    for e_k, pl_k in extra_engs:
        for p_ in pl_k:
            p_.close()
        e_k.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_cpu_gat_lp_baseline(eng, model, anchors, negs, fanouts, heads, L, budget_s=15.0, train=False):
    """the CPU port of the GAT link-prediction step on one host core: oracle sampler + collate (C) of the anchors +
    positives batch and of the random-negative batch, 2-layer GAT forward over the WHOLE union graph in fp32 torch
    (oracle/gnn_ref.gat_conv: the reference's execution order), inner-product scores and the retrieval loss rows —
    full steps of the GPU line's shape, counted in its unit (sampled edges + the edges the trimmed schedule aggregates).
    The positives (one sampled out-neighbour per anchor) are taken from the device, untimed: they are an input here.
    train: the TRAINING step — the same forward with autograd, an in-batch softmax loss, backward and an Adam update."""
    import oracle
    from oracle import gnn_ref
    rowptr, col = eng.graph_to_host()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(1)
    opt = None
    if train:
        sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        opt = torch.optim.Adam(list(sd.values()), lr=5e-3, weight_decay=1e-6)

    def encode(roots):
        nbr, cnt = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
        u = oracle.union_build(roots, fanouts, nbr)
        return u, cnt, gnn_ref.union_edge_index(u["rowptr"], u["col"])

    def units(u, cnt):
        meta, rp = u["meta"], u["rowptr"].astype(np.int64)
        return int(sum(int(c.sum()) for c in cnt)) + sum(int(rp[int(meta[2 + (L - 1 - l)])]) for l in range(L))

    def fetch(u):
        ids = torch.from_numpy(u["nodes"].astype(np.int64)).to(torch.int32).to(eng.device)
        return eng.gather_rows(ids, torch.tensor([ids.numel()], dtype=torch.int32, device=eng.device), int(ids.numel())).cpu()

    def forward(x, ei):
        h = x
        for l in range(L):
            pfx = f"conv_layers.{l}."
            h = gnn_ref.gat_conv(h, ei, sd[pfx + "lin.weight"], sd[pfx + "att_src"].reshape(-1), sd[pfx + "att_dst"].reshape(-1),
                                 sd.get(pfx + "bias"), heads if l < L - 1 else 1)
            if l < L - 1:
                h = torch.relu(h)
        return h

    t_used, edges, steps = 0.0, 0, 0
    while t_used < budget_s and steps < anchors.shape[0]:
        a = anchors[steps]
        pos, _ = eng.sample_positives(a, 1)
        a_h, p_h = a.cpu().numpy().view(np.uint32), pos.reshape(-1).cpu().numpy().view(np.uint32)
        ng = negs[steps].cpu().numpy().view(np.uint32)
        t0 = time.perf_counter()
        um, cm, eim = encode(np.concatenate([a_h, p_h[p_h != 0xFFFFFFFF]]))  # (an anchor without an out-edge has no positive)
        un, cn, ein = encode(ng)
        t_used += time.perf_counter() - t0
        xm, xn = fetch(um), fetch(un)
        t0 = time.perf_counter()
        em = forward(xm, eim)[torch.from_numpy(um["root_local"].astype(np.int64)).clamp(min=0)]
        en = forward(xn, ein)[torch.from_numpy(un["root_local"].astype(np.int64)).clamp(min=0)]
        B = a_h.size
        scores = em[:B] @ torch.cat([em[B:], en]).T / 0.07
        loss = torch.logsumexp(scores, dim=1).sum()
        if train:
            opt.zero_grad()
            (loss - scores[:, : min(B, scores.shape[1])].diagonal().sum()).div(B).backward()
            opt.step()
        t_used += time.perf_counter() - t0
        edges += units(um, cm) + units(un, cn)
        steps += 1
    return {"value": edges / max(t_used, 1e-9), "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": f"{steps} full steps ({anchors.shape[1]} anchors + positives, {negs.shape[1]} random negatives) of the "
                      f"same graph / fanout, {t_used:.1f} s; sampler + collate = oracle/gigl_oracle.c, forward = "
                      "oracle/gnn_ref.gat_conv over the whole union graph (fp32 torch, 1 thread), scores + loss rows in torch"
                      + (", autograd backward + Adam" if train else "")}


def run_cpu_baseline(eng, model, my, fanouts, W, n, d):
    """-> (cpu_baseline on one core, the same on many host cores).
    The CPU port of the same step — oracle (C restatement of the reference sampler + collate) + fp32 torch CPU forward
    over the WHOLE union graph (the reference's execution order, L*|E_union| edge visits) — on FULL batches of the
    same B roots, fanout and graph as the GPU line.  The unit is the GPU line's: sampled edges + the edges the trimmed
    schedule aggregates (sum_l |E_l|) of those batches, whatever extra work the reference order does for them."""
    import oracle
    from oracle import gnn_ref

    rowptr, col = eng.graph_to_host()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    L = len(fanouts)
    B = int(my.shape[1])

    def units(u, cnt):
        """the metric's edge count of one batch: sampled + sum_l (in-edges of the rows layer l must compute)"""
        meta, rp = u["meta"], u["rowptr"].astype(np.int64)
        agg = sum(int(rp[int(meta[2 + (L - 1 - l)])]) for l in range(L))  # rows are level-ordered: a prefix per layer
        return int(sum(int(c.sum()) for c in cnt)) + agg

    def fetch(u):  # the union graph's feature rows as fp32 (the reference's records carry them); untimed
        ids = torch.from_numpy(u["nodes"].astype(np.int64)).to(torch.int32).to(eng.device)
        n_dev = torch.tensor([ids.numel()], dtype=torch.int32, device=eng.device)
        return eng.gather_rows(ids, n_dev, int(ids.numel())).cpu()

    budget_s, t_used, edges, ref_edges, batches = 20.0, 0.0, 0, 0, 0
    torch.set_num_threads(1)
    i = W
    while t_used < budget_s and batches < 64:
        roots = my[i % my.shape[0]].cpu().numpy().view(np.uint32)
        t0 = time.perf_counter()
        nbr, cnt = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
        u = oracle.union_build(roots, fanouts, nbr)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        t_used += time.perf_counter() - t0
        xs = fetch(u)
        t0 = time.perf_counter()
        out = gnn_ref.graphsage_forward(xs, ei, sd, L)
        _ = out[u["root_local"]]
        t_used += time.perf_counter() - t0
        edges += units(u, cnt)
        ref_edges += int(sum(int(c.sum()) for c in cnt)) + L * int(u["meta"][1])
        batches += 1
        i += 1
    one = {"value": edges / t_used, "unit": "edges/s", "cores": 1, "kind": "port",
           "sample": f"{batches} full batches of {B} roots of the same graph/fanout, {t_used:.1f} s; sampler+collate = "
                     "oracle/gigl_oracle.c (1 thread), forward = fp32 torch CPU (1 thread) over the whole union graph "
                     "(the reference's execution order); edges counted in the GPU line's unit (sampled + trimmed "
                     f"aggregated); in the reference's own count (sampled + L*|E_union|) it is {ref_edges / t_used:.0f}/s"}
    # ---- the same work on many host cores ("run at 1 thread and at all cores"): one batch per worker
    # thread at a time (the C oracle and the torch ops release the GIL), two timed stages with the feature fetch between
    from concurrent.futures import ThreadPoolExecutor
    cores = min(os.cpu_count() or 1, 64)  # worker threads actually used (more only add GIL contention)
    nb = cores  # a bounded sample: one full batch per worker
    todo = [my[(W + batches + k) % my.shape[0]].cpu().numpy().view(np.uint32) for k in range(nb)]

    def stage1(roots):
        nbr, cnt = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
        u = oracle.union_build(roots, fanouts, nbr)
        return u, gnn_ref.union_edge_index(u["rowptr"], u["col"]), units(u, cnt)

    def stage2(item):
        (u, ei, _), xs = item
        return gnn_ref.graphsage_forward(xs, ei, sd, L)[u["root_local"]].shape[0]

    with ThreadPoolExecutor(max_workers=cores) as pool:
        t0 = time.perf_counter()
        s1 = list(pool.map(stage1, todo))
        t_all = time.perf_counter() - t0
        xs_all = [fetch(u) for u, _, _ in s1]
        t0 = time.perf_counter()
        list(pool.map(stage2, zip(s1, xs_all)))
        t_all += time.perf_counter() - t0
    edges_all = sum(c for _, _, c in s1)
    allc = {"value": edges_all / t_all, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{nb} full batches of {B} roots spread over {cores} worker threads (one batch per thread, "
                      f"1 intra-op thread each), {t_all:.1f} s wall; same code and unit as cpu_baseline"}
    return one, allc


if __name__ == "__main__":
    main()
