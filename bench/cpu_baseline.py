"""bench/cpu_baseline.py — the `cpu_baseline` legs: the CPU restatement (oracle/) timed on the host cores.  The ONLY
bench module that imports `oracle`; nothing here is called inside a timed GPU region."""
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
