"""bench/train.py — training steps: node classification (--train) and link prediction (--train --train-task lp)."""
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
from .cpu_baseline import run_cpu_train_baseline


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
