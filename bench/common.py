"""bench/common.py — what every workload of bench.py shares: the JSON line, peaks, live PMC passes, graph generators,
the self-launcher, the products-family workload builder."""
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # the repo root
BENCH_PY = os.path.join(ROOT, "bench.py")    # the CLI every child process is started through
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
    # (fused layers: the last layer is sage_fused_out_kernel, both projections linear_fused2x_kernel — round 5: linear_fused2_kernel)
    "gather_mean": ["gather_mean_kernel", "sage_fused_out_kernel"],
    "linear": ["linear_split_kernel", "linear_lds_kernel", "linear_mfma_kernel", "linear_fused2_kernel", "linear_fused2w_kernel",
               "linear_fused2x_kernel"],
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
                              ("bench", [sys.executable, BENCH_PY, "--streams", "1", "--no-graph", "--steps",
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
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env,
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

