"""bench/products.py — the fused one-call plan (gigl_sage_plan_run) over a graph resident on one GPU: the N = 1 headline."""
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

from .common import *  # noqa: F401,F403
from .common import _LIVE_PMC  # noqa: F401
from .cpu_baseline import run_cpu_baseline
from .sharded import run_emulated_world, run_sharded  # noqa: F401


def run_products(args, rank, world, local_rank):
    """the headline: BASELINE configs[1] (products-shaped) and the other single-GPU shapes of the same plan (mag-shard,
    rmat-shard, cora) — S streams x G batches per library call, replayed as hipGraphs"""
    import torch.distributed as dist
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
    # --steps that IS a whole number of rounds is honoured: a timed repetition is exactly K steps (the line's `steps`;
    # `steps_honoured`: true); anything else is rounded up to whole rounds, at least --min-rounds, and the line says so
    K_rep = K if (K >= rnd and K % rnd == 0) else max(-(-K // rnd), args.min_rounds) * rnd
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
            gp_ = getattr(args, "graph_priority", "off")
            if gp_ in ("high", "same"):
                # the latency-bound graph part of a call ahead of the other calls' bandwidth-bound layers
                plans[s].set_graph_stream(torch.cuda.Stream(device=dev, priority=-1 if gp_ == "high" else 0))
            elif gp_ in ("shared", "shared-high"):
                # ONE stream for every plan's graph part: a pipeline — the integer halves one after the other on it, the float
                # halves on the plans' own streams (always one latency-bound kernel beside S bandwidth-bound ones)
                if s == 0:
                    shared_graph_stream = torch.cuda.Stream(device=dev, priority=-1 if gp_ == "shared-high" else 0)
                plans[s].set_graph_stream(shared_graph_stream)
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
    # the dominant group = the one that costs the most WHERE THE NUMBER IS TAKEN: the largest HIP-event time under the timed
    # regime (S streams in flight; the single-stream ranking is kept as `dominant_alone`).  A group is every kernel of one
    # stage of the step: `gather_mean` = the first layer's segmented reduce + the fused last-layer reduction
    # (gather_mean_kernel + sage_fused_out_kernel), `linear` = the projection(s) (linear_fused2x_kernel: the single kernel with
    # the most time in the rocprofv3 summary; its own fractions are groups.linear)
    dominant_alone = dominant
    dominant = max(prof_ovl, key=lambda k: prof_ovl[k][0])

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
    # (round 6, linear_fused2x_kernel: ONE plane of whole p rows — fused_planes() == 1)
    p_planes = plans[0].fused_planes() if fused_layers and hasattr(plans[0], "fused_planes") else 2

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
                ab["gather_mean"] += agg_l * (4 + p_planes * 48 * 4) + rows_l * (8 + p_planes * 48 * 4 + out_dim * 4)
                continue
            if fused_layers and l == 0:
                two_src = True
                ab["gather_mean"] += agg_l * (4 + dims[l] * s_in) + rows_l * (8 + dims[l] * 4)
                # operand rows in, two planes of 96 floats out; + the second product's flops (256 -> 96, three products)
                # (one plane: a tile past the roots writes the W_l columns only — 64 of the 96 floats; the roots are the rows of
                # the LAST layer)
                rows_root = st[STATS["rows_layer0"] + 1] if (p_planes == 1 and not os.environ.get("GIGL_F2_ALL_WR")) else rows_l
                ab["linear"] += rows_l * 2 * dims[l] * 4 + (rows_root * 96 + (rows_l - rows_root) * 64) * 4 * p_planes + \
                    dout * 2 * dims[l] * 4 + 96 * dout * 4
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
                "dominant": dominant, "dominant_alone": dominant_alone,
                "dominant_from": "largest HIP-event time per kernel group under the TIMED regime (untimed probe with all "
                                 "timers on, the S streams in flight); dominant_alone = the single-stream ranking; a group = "
                                 "all kernels of one stage (gather_mean: gather_mean_kernel + sage_fused_out_kernel; linear: "
                                 "linear_fused2x_kernel, the largest SINGLE kernel of the rocprofv3 summary — its fractions "
                                 "are groups.linear); `groups` lists every group alone and overlapped",
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
            "steps": K_rep, "warmup": Wp, "ms_per_step": elapsed / steps_total * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "steps_requested": K, "warmup_requested": W, "steps_honoured": K_rep == K, "warmup_honoured": Wp == W,
            "timing": {"repetitions": reps, "steps_per_repetition": K_rep, "steps_total": steps_total,
                       "timed_region_s": round(elapsed, 3),
                       "ms_per_step_median": q(ms_rep, 50), "ms_per_step_p10": q(ms_rep, 10),
                       "ms_per_step_p90": q(ms_rep, 90), "value_median": q(rate_rep, 50),
                       "value_p10": q(rate_rep, 10), "value_p90": q(rate_rep, 90),
                       "protocol": "`steps` = the steps of ONE timed repetition: --steps itself when it is a whole number "
                                   "of rounds (streams x batches_per_call steps; steps_honoured), else rounded up to whole "
                                   "rounds (>= --min-rounds); the repetition is repeated until >= --min-seconds, each one "
                                   "bracketed by barrier + synchronize; ms_per_step / value = over all repetitions' "
                                   "max-over-ranks times"},
            "config": {"workload": wl_label +
                       f" N={n} E={eng0.n_edges} {'directed' if wl_directed else 'bidirectionalised'} D={d} "
                       f"{'fp32' if esz == 4 else 'fp16'} features, fanout={fanouts} B={B}/GPU GraphSAGE "
                       f"{d}->{hid}->{out_dim} (fp32 accumulate) inference step (sample+union+forward), sampler mode="
                       + args.mode,
                       "graph": "replica per GPU, roots sharded across ranks",
                       "streams": S, "batches_per_call": G, "graph_priority": getattr(args, "graph_priority", "off"), "fused_layers": bool(fused_layers),
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
        cmd = [sys.executable, BENCH_PY, "--gpus", str(world), "--workload", "mag240m-sharded",
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
        cmd = [sys.executable, BENCH_PY, "--workload", "mag240m-sharded", "--emulate-world", "8",
               "--shard-scale", os.environ.get("GIGL_BENCH_EMULATE_SCALE", "0.08"), "--fanouts", "25,10", "--batch", "1024",
               "--shard-group", "32", "--steps", "256", "--no-cpu-baseline", "--no-live-pmc"]
        try:
            cp = subprocess.run(cmd, env=dict(os.environ, GIGL_BENCH_CHILD="1"), capture_output=True, text=True, timeout=limit)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode == 0 and lines:
                sub = json.loads(lines[-1])
                line["sharded_emulated"] = {k: sub.get(k) for k in ("emulated_world", "value", "value_is", "ms_per_step", "route",
                                                                    "config", "emulated", "world1_reference")}
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
