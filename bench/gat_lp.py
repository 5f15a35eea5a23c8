"""bench/gat_lp.py — BASELINE configs[4]: GAT link prediction (--workload gat-lp), inference and training steps."""
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
from .cpu_baseline import run_cpu_gat_lp_baseline


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
