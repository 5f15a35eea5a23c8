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

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--group G] [--small]

Steps are independent batches.  One library call takes G consecutive batches through ONE set of ~28 launches
(gigl_sage_plan_set_groups: every batch keeps its own union graph, rows are bit-identical to G single-batch
calls — tests/test_gpu_groups.py), because at B=1024 a launch set costs ~0.07 ms of dispatch floor against
~0.08 ms of work per batch; calls are pipelined over S HIP streams (one library ctx + one host thread per
stream, all sharing the HBM-resident graph; defaults S = 3, G = 32 — a sweep of S in 2..4, G in 8..32 stays within
±5 % of the best).  Exactly K steps (K*B roots) are timed in total; a remainder of
K mod G steps runs batch by batch.
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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # same guide: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)

# library timer id -> the device functions it brackets (names as rocprofv3 prints them, see scripts/pmc_summary.py)
PMC_KERNELS = {
    "expand": ["expand_kernel"],
    "gather_mean": ["gather_mean_kernel<float, 32, 1>", "gather_mean_kernel<float, 64, 1>"],
    "linear": ["linear_lds_kernel<2>"],
    "union_insert": ["init_scratch_kernel", "insert_roots_kernel", "insert_slots_kernel"],
    "union_nodes": ["count_kernel", "assign_kernel"],
    "union_edge_sort": ["edge_dedup_count_kernel", "row_scan_kernel", "edge_fill_kernel"],
    "union_csr": ["row_sort_kernel", "row_sort_big_kernel"],
}


def pmc_traffic(kernel_id: str):
    """HBM bytes per launch of `kernel_id` from the newest committed rocprofv3 PMC summary (profiles/*_pmc.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes by scripts/gpu_pmc.sh on the same workload, corrected
    with the factors calibrated there).  bench.py cannot collect PMC counters on itself, so this is a measured
    constant of the committed build, refreshed whenever the profile is; None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files or kernel_id not in PMC_KERNELS:
        return None, None
    doc = json.load(open(files[-1]))
    tot_bytes = tot_calls = 0.0
    for name in PMC_KERNELS[kernel_id]:
        e = doc["kernels"].get(name)
        if e is None:
            return None, None
        calls = e.get("FETCH_SIZE_calls", 0)
        tot_bytes += e["hbm_bytes_per_launch"] * calls
        tot_calls = max(tot_calls, calls) if kernel_id.startswith("union") else tot_calls + calls
    return (tot_bytes / tot_calls if tot_calls else None), os.path.basename(files[-1])


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
}


def build_workload(eng, args):
    dev = eng.device
    name = "small" if getattr(args, "small", False) else getattr(args, "workload", "products")
    n, scale, pairs, d, dtype, directed, hid, out_dim, seed, label = WORKLOADS[name]
    src, dst = rmat_edges_gpu(scale, pairs, seed=seed, device=dev)
    # fold the 2^scale id space onto [0, n) and scatter ids so hubs are not the low ids
    perm_mul = 0x9E3779B1
    src = ((src * perm_mul) % n).to(torch.int32)
    dst = ((dst * perm_mul) % n).to(torch.int32)
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
    ap.add_argument("--steps", type=int, default=1600)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--fanouts", type=str, default="25,10")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--group", type=int, default=0,
                    help="batches per library call: G independent batches of B roots share one set of launches "
                         "(each keeps its own union graph; results are bit-identical to G single-batch calls); "
                         "0 = the largest power of two <= min(32, steps / streams)")
    ap.add_argument("--workload", type=str, default="products", choices=["products", "mag-shard", "mag240m-sharded"],
                    help="products = BASELINE configs[1] (default, the N=1 workload; N>1: a replica per GPU); mag-shard = "
                         "one GPU's 1/8 share of the MAG240M-shaped graph as a self-contained graph (D=768 fp16, SAGE "
                         "768->256->256); mag240m-sharded = BASELINE configs[2]: the MAG240M-shaped graph hash-"
                         "partitioned over the ranks (owner = id %% world), per-hop all_to_all frontier exchange and "
                         "feature pull over RCCL — needs >= 2 GPUs at full size (--shard-scale shrinks it)")
    ap.add_argument("--shard-group", type=int, default=8,
                    help="mag240m-sharded: batches of B roots exchanged per set of collectives (dedup stays per batch)")
    ap.add_argument("--shard-scale", type=float, default=1.0,
                    help="mag240m-sharded: fraction of MAG240M's nodes and edges to generate (1.0 needs 8 GPUs' HBM)")
    ap.add_argument("--small", action="store_true", help="200k-node graph (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="counter-collection runs (scripts/gpu_pmc.sh): only warm-up + the timed region, so every "
                         "library launch in the trace is a grouped launch; prints timing without edge counts")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--mode", type=str, default="parity", choices=["parity", "fast"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == max(args.gpus, 1) or world == 1, "launch with torchrun --nproc-per-node == --gpus"
    if args.workload == "mag240m-sharded":
        return run_sharded(args, rank, world, local_rank)

    from gigl_amd._lib import KERNEL_IDS, MODE_FAST, MODE_SPARK_HASH
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE

    torch.cuda.set_device(local_rank)
    eng0 = HipEngine(local_rank)
    dev = eng0.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    B, K, W, S = args.batch, args.steps, args.warmup, max(1, args.streams)
    if args.group > 0:
        G = args.group
    else:  # as few pipelines as keep >= 16 steps each, G <= 32 batches per call, calls spread evenly
        S = min(S, max(1, K // 16))
        calls_per_stream = -(-K // (S * 32))
        G = max(1, K // (S * calls_per_stream))
    L = len(fanouts)
    mode = MODE_SPARK_HASH if args.mode == "parity" else MODE_FAST

    t0 = time.time()
    n, d = build_workload(eng0, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    esz = 4 if wl_dtype == torch.float32 else 2  # bytes per feature element in the resident table
    torch.manual_seed(0)
    model = GraphSAGE(d, hid, out_dim, num_layers=L).to(dev)
    # roots: seeded permutation of node ids (seed 42); rank r takes batches r, r+world, ...
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    total_batches = (W + K) * world
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
            plans[s].use_graph(True)  # the call's ~28 launches replayed as one hipGraph launch
        outs.append(torch.empty((G * B, out_dim), dtype=torch.float32, device=dev))
    # single-batch plan: the tail of a step range that is not a multiple of G, and the untimed counting pass
    plan1 = model.make_plan(engines[0], B, fanouts) if G > 1 else plans[0]
    out1 = torch.empty((B, out_dim), dtype=torch.float32, device=dev)
    # the timed range's remainder (K mod G steps) goes through one more grouped call of its own size
    R = K % G
    plan_rem = model.make_plan(engines[0], B, fanouts, groups=R) if R > 1 else None
    if plan_rem is not None and not args.no_graph:
        plan_rem.use_graph(True)
    out_rem = torch.empty((max(R, 1) * B, out_dim), dtype=torch.float32, device=dev)

    def run_range(lo, hi, S=S):
        """steps (= batches of B roots) lo..hi-1: call c takes the G consecutive batches lo+c*G.. on pipeline
        c % S (one host thread per pipeline); a remainder of < G steps runs as one call of its own size (timed range) or
        batch by batch, on pipeline 0"""
        n_calls = (hi - lo) // G

        def worker(s):
            for c in range(s, n_calls, S):
                i = lo + c * G
                plans[s].run(my[i:i + G].view(-1), out=outs[s], mode=mode)
            if s == 0:
                i0 = lo + n_calls * G
                if plan_rem is not None and hi - i0 == R:
                    plan_rem.run(my[i0:hi].view(-1), out=out_rem, mode=mode)
                else:
                    for i in range(i0, hi):
                        plan1.run(my[i], out=out1, mode=mode)
        ths = [threading.Thread(target=worker, args=(s,)) for s in range(S)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    names = list(KERNEL_IDS)
    if args.timed_only:
        run_range(0, max(W // G, 1) * G)  # grouped calls only: every library launch in the trace is a grouped one
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_range(W, W + (K // G) * G)
        torch.cuda.synchronize()
        print(json.dumps({"timed_only": True, "steps": (K // G) * G, "ms_per_step": (time.perf_counter() - t1) / max((K // G) * G, 1) * 1e3}))
        for e in reversed(engines):
            e.close()
        return
    # ---- untimed: warm-up, then find the dominant kernel with all event timers on
    P = min(max(G, (min(W, 2 * S * G) // G) * G), W + K)  # probe steps: whole calls, about two per pipeline
    run_range(0, P)
    torch.cuda.synchronize()
    for e in engines:
        e.profile_enable(names, capacity=64 * 16)
    # the probe runs on ONE stream so that an event interval is the kernel's own duration (with S streams the
    # intervals include time shared with the other streams' kernels)
    run_range(0, P, S=1)
    run_range(0, P, S=1)  # (graph mode: the first call after a mask change re-captures)
    for p in plans:
        p.flush_profile()
    prof = {k: [sum(x) for x in zip(*[e.profile_read(k) for e in engines])] for k in names}
    # graph mode: the first call after the mask change runs untimed (it is the eager pass before the re-capture)
    probe_steps_counted = 2 * P - (0 if args.no_graph else min(G, P))
    dominant = max(prof, key=lambda k: prof[k][0])
    for e in engines:
        e.profile_enable([dominant], capacity=(K // S + 8) * 8)
    run_range(0, W)
    # every plan replays from its hipGraph in the timed region: one untimed call each re-captures after the timer
    # mask change (warm-up alone does not reach all plans when W < S*G)
    for s_i in range(S):
        plans[s_i].run(my[:G].view(-1), out=outs[s_i], mode=mode)
        plans[s_i].run(my[:G].view(-1), out=outs[s_i], mode=mode)
    if plan_rem is not None:
        plan_rem.run(my[:R].view(-1), out=out_rem, mode=mode)
        plan_rem.run(my[:R].view(-1), out=out_rem, mode=mode)
    torch.cuda.synchronize()
    for e in engines:
        e.profile_reset()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run_range(W, W + K)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    for p in plans + ([plan_rem] if plan_rem is not None else []):
        p.flush_profile()
    dom_ms, dom_launches = [sum(x) for x in zip(*[e.profile_read(dominant) for e in engines])]
    for e in engines:
        e.profile_enable([], 0)

    # ---- untimed: exact edge counts and algorithmic bytes of the same K batches (sampling is deterministic)
    g_rowptr, _ = (None, None)
    rp_host = np.empty(n + 1, dtype=np.int64)
    rp_dev, _cl = C.c_void_p(), C.c_void_p()
    eng0._lib.gigl_graph_device_ptrs(eng0._graph, C.byref(rp_dev), C.byref(_cl))
    eng0._lib.gigl_memcpy(eng0._ctx, C.c_void_p(rp_host.ctypes.data), 0, rp_dev, 1, rp_host.nbytes)
    deg_all = np.diff(rp_host)
    sampled = aggregated = ref_equiv = 0
    alg_bytes = {k: 0.0 for k in names}
    flops_l = []
    dims = [d] + [hid] * (L - 1)
    heavy_thr = 4096
    count_steps = range(W, W + K)  # every timed batch, exactly
    for i in count_steps:
        plan1.run(my[i], out=out1, mode=mode)
        hb = plan1.last_batch_to_host()
        meta = hb["meta"]
        if meta[8]:
            raise RuntimeError("union dedup overflow (meta[GIGL_META_OVERFLOW])")
        s_edges = int(sum(int(c.sum()) for c in hb["cnt"]))
        sampled += s_edges
        nn, ne = int(meta[0]), int(meta[1])
        rowlen = (hb["rowend"][:nn] - hb["rowptr"][:nn]).astype(np.int64)
        agg_l = [int(rowlen[: int(meta[2 + (L - 1 - l)])].sum()) for l in range(L)]
        aggregated += sum(agg_l)
        ref_equiv += L * ne
        # algorithmic bytes:
        #  gather layer l: E_l*(4 + D_l*s) + N_dst*(8 + D_l*s_out) (+ the fused self-row copy D_l*s read + write)
        for l in range(L):
            n_dst = int(meta[2 + (L - 1 - l)])
            s_in = esz if l == 0 else 4  # layer 0 gathers rows of the resident table, later layers fp32 activations
            alg_bytes["gather_mean"] += agg_l[l] * (4 + dims[l] * s_in) + n_dst * (8 + dims[l] * s_in + 2 * dims[l] * 4)
        #  union: 16 B per sampled edge + 4 B per unique node, attributed evenly to its phases
        for k in ("union_insert", "union_relax", "union_nodes", "union_edge_sort", "union_csr"):
            alg_bytes[k] += (16 * s_edges + 4 * nn) / 4.0
        #  sampler (parity mode): 16 + 4*deg + 8*min(deg, f) per frontier node
        roots_h = my[i].cpu().numpy().view(np.uint32)
        parents = [roots_h] + [hb["nbr"][k] for k in range(L - 1)]
        for k in range(L):
            p = parents[k]
            valid = p != 0xFFFFFFFF
            dg = deg_all[np.where(valid, p, 0)] * valid
            per = 16 * valid + 4 * dg + 8 * np.minimum(dg, fanouts[k])
            alg_bytes["expand"] += float(per.sum())
        #  dense projections are FLOP-bound; bytes = operands once
        for l in range(L):
            n_dst = int(meta[2 + (L - 1 - l)])
            dout = hid if l < L - 1 else out_dim
            alg_bytes["linear"] += n_dst * (2 * dims[l] + dout) * 4 + dout * 2 * dims[l] * 4
            flops_l.append(2.0 * n_dst * 2 * dims[l] * dout)
    scale = K / len(count_steps)
    sampled, aggregated, ref_equiv = sampled * scale, aggregated * scale, ref_equiv * scale
    alg_bytes = {k: v * scale for k, v in alg_bytes.items()}
    alg_flops_linear = scale * sum(flops_l)

    # ---- reduce over ranks
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cc = torch.tensor([sampled, aggregated, ref_equiv], dtype=torch.float64, device=dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        sampled_all, aggregated_all, ref_equiv_all = [float(v) for v in cc.tolist()]
    else:
        sampled_all, aggregated_all, ref_equiv_all = float(sampled), float(aggregated), float(ref_equiv)

    value = (sampled_all + aggregated_all) / elapsed
    avg_launch_ms = dom_ms / max(dom_launches, 1)
    bytes_per_launch = alg_bytes[dominant] / max(dom_launches, 1)
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    probe_steps = max(probe_steps_counted, 1)
    traffic, traffic_src = pmc_traffic(dominant)
    # every kernel group against its own bound, from the untimed all-timers-on probe (approximate: the probe steps
    # are other batches than the K counted ones; per-step averages)
    by_kernel = {}
    for k, v in prof.items():
        ms_step = v[0] / probe_steps
        if ms_step <= 0:
            continue
        if k == "linear":
            tf = alg_flops_linear / K / (ms_step * 1e-3) / 1e12
            by_kernel[k] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                            "frac": round(tf / MFMA_F32_PEAK_TF, 4), "ms_per_step": round(ms_step, 4)}
        else:
            gbs = alg_bytes[k] / K / (ms_step * 1e-3) / 1e9
            by_kernel[k] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 4), "ms_per_step": round(ms_step, 4)}
    note = None
    if dominant == "expand" and args.mode == "parity":
        note = ("algorithmic bytes of parity sampling count the whole adjacency row of every frontier node "
                "(16 + 4*deg + 8*min(deg,f), SURVEY.md 8(d)); the kernel never reads the row: it selects from the "
                "precomputed table of the hash sequence (12.8 B per tabulated position, read around the window's "
                "threshold) and fetches only the f selected ids, so measured HBM traffic (`traffic`) is a fraction of "
                "the algorithmic bytes and the kernel is instruction-bound (SQ counters: ~77 % VALU-busy), not "
                "HBM-bound: frac is the contract's figure, not a bandwidth utilisation; by_kernel.gather_mean is "
                "the HBM-bound kernel")
    if dominant == "linear":  # the dense projection is the one MFMA-bound kernel
        tf = alg_flops_linear / max(dom_launches, 1) / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        head = {"bound": "mfma", "kernel": dominant, "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF,
                "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 5)}
    else:
        head = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5)}
    roofline = {**head,
                "traffic": None if traffic is None else round(traffic), "traffic_source": traffic_src,
                "avg_launch_us": round(avg_launch_ms * 1e3, 2), "alg_bytes_per_launch": round(bytes_per_launch),
                "launches": int(dom_launches), "note": note,
                "timing": f"HIP events on the kernel's stream over the timed region ({S} streams: intervals include "
                          "overlap with the other streams' kernels); by_kernel: single-stream untimed probe",
                "by_kernel": by_kernel}

    cpu_baseline = cpu_baseline_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (a reported baseline: N=1 only)
        cpu_baseline, cpu_baseline_all = run_cpu_baseline(eng0, model, my, fanouts, W, n, d)

    if rank == 0:
        line = {
            "metric": "sampled+aggregated edges/s", "value": value, "unit": "edges/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_label +
                       f" N={n} E={eng0.n_edges} {'directed' if wl_directed else 'bidirectionalised'} D={d} "
                       f"{'fp32' if esz == 4 else 'fp16'} features, fanout={fanouts} B={B}/GPU GraphSAGE "
                       f"{d}->{hid}->{out_dim} (fp32 accumulate) inference step (sample+union+forward), sampler mode="
                       + args.mode,
                       "graph": "replica per GPU, roots sharded across ranks",
                       "streams": S, "batches_per_call": G,
                       "sampled_edges_per_step": sampled_all / (K * world),
                       "aggregated_edges_per_step": aggregated_all / (K * world),
                       "reference_equivalent_aggregated_per_step": ref_equiv_all / (K * world),
                       "sampled_edges_per_s": sampled_all / elapsed, "aggregated_edges_per_s": aggregated_all / elapsed,
                       "edge_counts_from": "all timed batches re-run untimed (sampling is deterministic)",
                       "setup_s": round(setup_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baseline_all_cores": cpu_baseline_all,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for e in reversed(engines):
        e.close()


def run_sharded(args, rank, world, local_rank):
    """BASELINE.json configs[2]: MAG240M-shaped graph (N=244,160,499, E=1,728,364,232 directed RMAT, D=768 fp16,
    SURVEY.md 8(d) C3) hash-partitioned over the ranks: rank r holds the CSC rows and feature rows of the nodes
    with id % world == r (gigl_amd/dist.py).  A step = one batch of B roots per rank: per hop one all_to_all of
    (node, K) requests to the owners, gigl_expand_frontier there, one all_to_all back; union graph locally; the
    UNIQUE node ids pulled from their owners (all_to_all of ids, all_to_all of rows); 2-layer GraphSAGE 768->256->256.
    Frontier buckets are filled and scattered on the device (gigl_frontier_bucket / gigl_frontier_scatter) and move
    through equal-split all_to_alls; the only host read of a step is the split sizes of the feature-row exchange, and
    it is taken while the NEXT batch's sampling exchange is already queued on a second stream (two batches in flight:
    remote fetch overlapped with local expansion).  The collectives are still issued from Python (torch.distributed)."""
    import torch.distributed as dist
    from gigl_amd._lib import GIGL_META_LEVEL0
    from gigl_amd.dist import HipDistKHopSampler, HipFeaturePuller
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE, HipBatch

    torch.cuda.set_device(local_rank)
    if world == 1 and not dist.is_initialized():  # single-rank group: same code path, the exchange is a self-copy
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    L = len(fanouts)
    B, K, W = args.batch, args.steps, args.warmup
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
    dist.all_reduce(maxdeg, op=dist.ReduceOp.MAX)
    dist.all_reduce(e_local, op=dist.ReduceOp.SUM)
    eng.load_csc(rowptr, col)
    del key, rowptr, col
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x_local = torch.empty((n_local, d), device=dev, dtype=torch.float16)
    step_rows = max(1, (1 << 28) // d)
    for i in range(0, n_local, step_rows):
        x_local[i:i + step_rows] = torch.randn((min(step_rows, n_local - i), d), generator=g, device=dev).to(torch.float16)
    torch.cuda.empty_cache()
    torch.manual_seed(0)
    model = GraphSAGE(d, hid, out_dim, num_layers=L).to(dev)
    # every hash window ends below (hops+1)*n + seed*hops + maxdeg: lets the owners use the range table throughout
    bound = (L + 1) * n + 42 * L + int(maxdeg.item())
    gp = torch.Generator(device="cpu")
    gp.manual_seed(42)
    total_batches = (W + K) * world
    perm = torch.randint(0, n, (total_batches * B,), generator=gp)
    my = perm.view(total_batches, B)[rank::world].to(torch.int32).to(dev).contiguous()
    # G consecutive batches travel together: one set of collectives and launches per G steps; the union graph keeps
    # the batches apart (gigl_union_build_groups: dedup within a batch only), so a step's edges are those of its batch
    G = max(1, min(args.shard_group, K))
    while K % G or W % G:
        G -= 1
    GB = G * B
    my = my.view(-1, GB)

    class Slot:  # everything one in-flight batch owns; two slots alternate so that the host read of batch i's split
        pass     # sizes waits while batch i+1's sampling is already queued on the other stream

    slots = []
    for si in range(2):
        sl = Slot()
        sl.stream = torch.cuda.Stream(device=dev)
        sl.eng = eng if si == 0 else HipEngine(local_rank)  # a ctx (stream, arena, hash table) per in-flight batch
        if si:
            sl.eng.share_resident(eng)
        sl.eng.bind_stream(sl.stream)
        sl.tree = sl.eng.alloc_tree(GB, fanouts)
        sl.tree.c_struct.hops, sl.tree.c_struct.b = L, GB
        for k, f in enumerate(fanouts):
            sl.tree.c_struct.fanouts[k] = f
        sl.u = sl.eng.alloc_union(GB, fanouts)
        sl.sampler = HipDistKHopSampler(sl.eng, world, max_window_end=bound if bound < (1 << 30) else -1, sampling_seed=42)
        sl.puller = HipFeaturePuller(sl.eng, world, x_local, int(sl.u.nodes.numel()))
        slots.append(sl)
    acc = torch.zeros(3, dtype=torch.int64, device=dev)  # sampled, aggregated, pulled (counted on the device)
    lvl = [GIGL_META_LEVEL0 + (L - 1 - l) for l in range(L)]
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    def phase1(sl, i):
        """sampling (2 all_to_all per hop) + union graph + feature requests (2 all_to_all): queued, never waited for"""
        with torch.cuda.stream(sl.stream):
            sl.tree.roots = my[i]
            _, sl.cnt = sl.sampler.sample_khop(sl.tree.roots, fanouts, tree=sl.tree)
            sl.eng.union_build(sl.tree, out=sl.u, group_roots=B)
            p = sl.puller
            req = p.request(sl.u.nodes, sl.u.meta[0])
            dist.all_to_all_single(p.got, req)
            dist.all_to_all_single(p.recv_counts[:world], p.counts[:world])
            sl.host = torch.cat([p.counts, p.recv_counts[:world], sl.u.meta[:1]]).to("cpu", non_blocking=True)
            sl.ready = torch.cuda.Event()
            sl.ready.record(sl.stream)

    def phase2(sl, count):
        """the step's one host read (split sizes), the row exchange, the forward"""
        sl.ready.synchronize()
        host = sl.host.tolist()
        sc, overflow, rc, n_rows = host[:world], host[world], host[world + 1: 2 * world + 1], host[-1]
        if overflow:
            raise RuntimeError("feature-pull bucket overflow")
        with torch.cuda.stream(sl.stream):
            p = sl.puller
            rows = p.serve(p.got, rc)
            back = rows.new_empty((int(sum(sc)), rows.shape[1]))
            dist.all_to_all_single(back, rows, output_split_sizes=sc, input_split_sizes=rc)
            out = model(HipBatch(engine=sl.eng, tree=sl.tree, union=sl.u, x=back, x_index=p.place_index(sc)))
            sl.rows = out[sl.u.root_local[:GB].to(torch.int64)]
            if count:
                u = sl.u
                rowlen = (u.rowend - u.rowptr).to(torch.int64)
                ar = torch.arange(rowlen.numel(), device=dev)
                agg = sum((rowlen * (ar < u.meta[j])).sum() for j in lvl)
                acc.add_(torch.stack([sum(c.sum() for c in sl.cnt).to(torch.int64), agg.to(torch.int64),
                                      u.meta[0].to(torch.int64)]))

    tim = {"phase1_enqueue": 0.0, "wait_split_sizes": 0.0, "phase2_enqueue": 0.0} if os.environ.get("GIGL_SHARD_TIMING") else None

    def run(lo, hi, count):
        phase1(slots[lo % 2], lo)
        for i in range(lo, hi):
            ta = time.perf_counter()
            if i + 1 < hi:
                phase1(slots[(i + 1) % 2], i + 1)
            tb = time.perf_counter()
            if tim is not None:
                slots[i % 2].ready.synchronize()
            tc = time.perf_counter()
            phase2(slots[i % 2], count)
            if tim is not None:
                tim["phase1_enqueue"] += tb - ta
                tim["wait_split_sizes"] += tc - tb
                tim["phase2_enqueue"] += time.perf_counter() - tc

    run(0, W // G, False)
    run(W // G, (W + K) // G, True)  # untimed: the edge counts of the timed batches (sampling is deterministic)
    for k_ in (tim or {}):
        tim[k_] = 0.0
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run(W // G, (W + K) // G, False)
    if int(max(int(sl.sampler.overflow.item()) for sl in slots)):
        raise RuntimeError("frontier bucket overflow: rerun with a larger slack")
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t1
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    cc = acc.to(torch.float64)
    dist.all_reduce(cc, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    sampled_all, aggregated_all, pulled_all = [float(v) for v in cc.tolist()]
    if rank == 0 and tim is not None:
        print("host time per step (ms):", {k: round(v / K * 1e3, 4) for k, v in tim.items()}, file=sys.stderr)
    if rank == 0:
        line = {
            "metric": "sampled+aggregated edges/s", "value": (sampled_all + aggregated_all) / elapsed,
            "unit": "edges/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"MAG240M-shaped RMAT x{args.shard_scale:g}: N={n} E={int(e_local.item())} directed, "
                                   f"D={d} fp16 features, hash-partitioned over {world} rank(s) (owner = id % world), "
                                   f"fanout={fanouts} B={B}/GPU GraphSAGE {d}->{hid}->{out_dim}, sampler mode=parity, "
                                   f"{G} batches per exchange",
                       "graph": "CSC rows + feature rows of the owned nodes per rank; per-hop all_to_all frontier "
                                "exchange, feature pull of the unique union-graph nodes",
                       "sampled_edges_per_step": sampled_all / (K * world),
                       "aggregated_edges_per_step": aggregated_all / (K * world),
                       "pulled_feature_rows_per_step": pulled_all / (K * world),
                       "pulled_feature_bytes_per_s": pulled_all * d * 2 / elapsed, "setup_s": round(setup_s, 1)},
            "roofline": None, "cpu_baseline": None,
        }
        print(json.dumps(line))
    dist.barrier()
    dist.destroy_process_group()
    for sl in reversed(slots):
        sl.eng.close()


def run_cpu_baseline(eng, model, my, fanouts, W, n, d):
    """-> (cpu_baseline on one core, the same on all host cores).
    oracle (C port of the reference sampler + collate) + fp32 torch CPU forward over the WHOLE union graph
    (reference semantics), single thread for the sampler, on a bounded sample of the same workload."""
    import oracle
    from oracle import gnn_ref

    rowptr, col = eng.graph_to_host()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    L = len(fanouts)
    budget_s, t_used, edges, batches = 20.0, 0.0, 0, 0
    sub = 128  # roots per CPU batch (bounded sample: the full B=1024 batch costs minutes on one core)
    torch.set_num_threads(1)
    i = W
    while t_used < budget_s and batches < 64:
        roots = my[i % my.shape[0]].cpu().numpy().view(np.uint32)[:sub]
        t0 = time.perf_counter()
        nbr, cnt = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
        u = oracle.union_build(roots, fanouts, nbr)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        t_used += time.perf_counter() - t0
        # the union graph's feature rows as fp32 (the reference's records carry them): fetched from the resident
        # table, outside the timed region
        ids = torch.from_numpy(u["nodes"].astype(np.int64)).to(torch.int32).to(eng.device)
        n_dev = torch.tensor([ids.numel()], dtype=torch.int32, device=eng.device)
        xs = eng.gather_rows(ids, n_dev, int(ids.numel())).cpu()
        t0 = time.perf_counter()
        out = gnn_ref.graphsage_forward(xs, ei, sd, L)
        _ = out[u["root_local"]]
        t_used += time.perf_counter() - t0
        edges += int(sum(int(c.sum()) for c in cnt)) + L * int(u["meta"][1])
        batches += 1
        i += 1
    one = {"value": edges / t_used, "unit": "edges/s", "cores": 1, "kind": "port",
           "sample": f"{batches} batches of {sub} roots of the same graph/fanout, {t_used:.1f} s; sampler+collate = "
                     "oracle/gigl_oracle.c (1 thread), forward = fp32 torch CPU (1 thread) over the whole union graph "
                     "(L*|E_union| aggregated edges, the reference's execution order)"}
    # ---- the same work on many host cores ("run at 1 thread and at all cores"): one batch per worker
    # thread at a time (the C oracle and the torch ops release the GIL), two timed stages with the feature fetch between
    from concurrent.futures import ThreadPoolExecutor
    cores = min(os.cpu_count() or 1, 64)  # worker threads actually used (more only add GIL contention)
    nb = 4 * cores  # a bounded sample: a few batches per worker
    todo = [my[(W + batches + k) % my.shape[0]].cpu().numpy().view(np.uint32)[:sub] for k in range(nb)]

    def stage1(roots):
        nbr, cnt = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
        u = oracle.union_build(roots, fanouts, nbr)
        return u, gnn_ref.union_edge_index(u["rowptr"], u["col"]), int(sum(int(c.sum()) for c in cnt))

    def stage2(item):
        (u, ei, _), xs = item
        return gnn_ref.graphsage_forward(xs, ei, sd, L)[u["root_local"]].shape[0]

    with ThreadPoolExecutor(max_workers=cores) as pool:
        t0 = time.perf_counter()
        s1 = list(pool.map(stage1, todo))
        t_all = time.perf_counter() - t0
        xs_all = []
        for u, _, _ in s1:
            ids = torch.from_numpy(u["nodes"].astype(np.int64)).to(torch.int32).to(eng.device)
            n_dev = torch.tensor([ids.numel()], dtype=torch.int32, device=eng.device)
            xs_all.append(eng.gather_rows(ids, n_dev, int(ids.numel())).cpu())
        t0 = time.perf_counter()
        list(pool.map(stage2, zip(s1, xs_all)))
        t_all += time.perf_counter() - t0
    edges_all = sum(c + L * int(u["meta"][1]) for u, _, c in s1)
    allc = {"value": edges_all / t_all, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{nb} batches of {sub} roots spread over {cores} worker threads (one batch per thread at a time, "
                      f"1 intra-op thread each), {t_all:.1f} s wall; same code as cpu_baseline"}
    return one, allc


if __name__ == "__main__":
    main()
