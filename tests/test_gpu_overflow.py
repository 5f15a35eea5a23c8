"""Batches that outgrow a plan's workspace are HANDLED, not reported.

A one-call plan provisions b*(1 + f0 + ...) inner rows — every batch in which no root is another root's sampled neighbour.
On a graph of a few thousand nodes with a small first and a large second fan-out that bound fails all the time (the
children of a root-valued slot move up a level); the reference's collate has no such bound
(/root/reference/python/gigl/src/training/v1/lib/data_loaders/rooted_node_neighborhood_data_loader.py:78-158 builds whatever
the records hold).  Every entry point therefore redoes such a batch and hands out the rows / applies the update the
reference would: the inferencer (pipelined calls settled before their rows reach the writer; staged launches for the redo;
on a hash-partitioned graph every rank redoes the call through worst-case buckets), the node-classification plan (the device
halts the queue, the plan grows, the steps are redone from the failed batch on), both link-prediction plans (grown on the
first NaN loss).  Bars as everywhere: 1e-5 for rows against the TFRecord route / the fp32 CPU forward, 1e-4 for loss
histories against the CPU restatement of the training loop."""
import os

import numpy as np
import pytest
import torch
import yaml

import oracle
from oracle import gnn_ref
from test_gpu_hbm_route import _infer_worker, _rows, _variant, _write_small_job
from test_gpu_train_plan import _lp_batches, _lp_loss_torch

pytestmark = pytest.mark.gpu

N, E, FAN, B = 4096, 60_000, (2, 50), 128


def _graph(d):
    from helpers import rmat_edges
    src, dst = rmat_edges(15, E, 7)
    src, dst = (src.astype(np.int64) * 0x9E3779B1) % N, (dst.astype(np.int64) * 0x9E3779B1) % N
    keep = src != dst
    src, dst = src[keep], dst[keep]
    rowptr, col = oracle.build_csc(N, src.astype(np.uint32), dst.astype(np.uint32), is_directed=False)
    x = (np.random.default_rng(3).standard_normal((N, d)) / 4).astype(np.float32)
    return rowptr, col, x


def _overflows(rowptr, col, roots):
    nbr, _ = oracle.sample_khop(rowptr, col, roots, list(FAN), canonical=True)
    u = oracle.union_build(roots, list(FAN), nbr)
    return int(u["meta"][3]) > roots.size * (1 + FAN[0]), u


# ---------------------------------------------------------------------------------------------- inference
@pytest.fixture(scope="module")
def overflow_job(tmp_path_factory):
    from gigl_amd.models import GraphSAGE
    base = str(tmp_path_factory.mktemp("gigl_overflow"))
    n, src, dst, x = _write_small_job(base, n=N, e=E, d=16, hid=32, out_dim=8, fan=FAN, batch=B)
    torch.manual_seed(5)
    model = GraphSAGE(16, 32, 8, num_layers=2)
    os.makedirs(os.path.join(base, "out/model"), exist_ok=True)
    torch.save(model.state_dict(), os.path.join(base, "out/model/model.pt"))
    return base, n, src, dst, x


def test_inferencer_redoes_the_calls_that_outgrow_the_plan(overflow_job, monkeypatch):
    """4,096 nodes, fan-out [2, 50], 128 roots per batch: more than half of the batches do not fit the one-call plan.  The
    pipelined in-HBM route (three lanes, calls settled before their rows reach the writer) writes the TFRecord route's rows
    (1e-5) — and the fp32 CPU forward's over oracle-collated batches, for a batch that fits and one that does not."""
    from gigl_amd.config import GbmlConfigPbWrapper
    from gigl_amd.inferencer import Inferencer
    from gigl_amd.subgraph_sampler import SubgraphSampler
    from gigl_amd import config
    base, n, src, dst, x = overflow_job
    monkeypatch.setattr(config, "RECORDS_PER_PART_FILE", 3000)  # (two part files: the reader's permutation matters)
    SubgraphSampler().run("job", "configs/job.yaml", None, uri_base=base)
    a, b = Inferencer(), Inferencer()
    out_t = a.run("job", _variant(base, "configs/job.yaml", "tf"), None, uri_base=base, route="tfrecord")
    out_h = b.run("job", _variant(base, "configs/job.yaml", "hbm"), None, uri_base=base, route="hbm")
    assert a.rows_written == b.rows_written == n
    assert b.hbm_overflow_redone > 0, "no call overflowed: the test graph no longer exercises the redo"
    rt, rh = _rows(out_t["embeddings"]), _rows(out_h["embeddings"])
    ids = [r["node_id"] for r in rh]
    assert ids == [r["node_id"] for r in rt] and sorted(ids) == list(range(n))
    eh, et = np.array([r["emb"] for r in rh], np.float32), np.array([r["emb"] for r in rt], np.float32)
    assert np.isfinite(eh).all()
    np.testing.assert_allclose(eh, et, rtol=1e-5, atol=1e-5)
    cfg = GbmlConfigPbWrapper.from_uri("configs/job.yaml", uri_base=base)
    rowptr, col = oracle.build_csc(n, src.astype(np.uint32), dst.astype(np.uint32), is_directed=False)
    sd = torch.load(cfg.trained_model_uri, map_location="cpu")
    seen = set()
    for lo in range(0, n, B):
        roots = np.array(ids[lo:lo + B], dtype=np.uint32)
        over, u = _overflows(rowptr, col, roots)
        if over in seen:
            continue
        seen.add(over)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        o = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, sd, 2)
        np.testing.assert_allclose(eh[lo:lo + B], o[u["root_local"]].numpy(), rtol=1e-5, atol=1e-5)
    assert seen == {False, True}


def test_one_encode_call_is_checked_and_redone_in_place():
    """ResidentGraph.encode outside a pipeline (no batch.defer_overflow_check): the flag is read in the call, whatever the
    graph's size — b = 1, fan-out [64, 64], a root that is its own sampled neighbour (tests/test_gpu_plan.py)"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.hbm import ResidentGraph
    from gigl_amd.models import GraphSAGE
    n, d = 400, 8
    found = None
    for r in range(1, 200):
        nbrs = np.unique(np.concatenate([[r], np.arange(200, 320)])).astype(np.uint32)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        rowptr[r + 1:] = nbrs.size
        nbr_o, _ = oracle.sample_khop(rowptr, nbrs, np.array([r], np.uint32), [64, 64], canonical=True)
        if r in nbr_o[0] and int(oracle.union_build(np.array([r], np.uint32), [64, 64], nbr_o)["meta"][3]) > 65:
            found = (r, rowptr, nbrs)
            break
    assert found is not None
    r, rowptr, col = found
    feats = np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)
    eng = HipEngine(0)
    try:
        eng.load_csc(rowptr, col)
        eng.load_features(feats)
        res = ResidentGraph.from_engine(eng, np.arange(n), [64, 64])
        torch.manual_seed(0)
        model = GraphSAGE(d, 8, 4, num_layers=2).to(eng.device)
        (hb,) = list(res.root_batches(np.array([r]), 1, 1))
        out = res.encode(model, hb)
        nbr_o, _ = oracle.sample_khop(rowptr, col, np.array([r], np.uint32), [64, 64], canonical=True)
        u = oracle.union_build(np.array([r], np.uint32), [64, 64], nbr_o)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        want = gnn_ref.graphsage_forward(torch.from_numpy(feats[u["nodes"].astype(np.int64)]),
                                         gnn_ref.union_edge_index(u["rowptr"], u["col"]), sd, 2)
        want = want[torch.from_numpy(u["root_local"].astype(np.int64))]
        np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
        assert res.overflow_redone == 1
        (ok,) = list(res.root_batches(np.array([r + 200 if r + 200 < n else 201]), 1, 1))
        assert torch.isfinite(res.encode(model, ok)).all() and res.overflow_redone == 1
        # a pipelined call nobody settles is reported at the end of the pass
        (hb2,) = list(res.root_batches(np.array([r]), 1, 1))
        hb2.defer_overflow_check = True
        assert torch.isnan(res.encode(model, hb2)).all()
        with pytest.raises(RuntimeError, match="not settled"):
            res.raise_on_overflow()
        res.raise_on_overflow()
        # ... and one that is settled says so, and its redo gives the rows
        (hb3,) = list(res.root_batches(np.array([r]), 1, 1))
        hb3.defer_overflow_check = True
        res.encode(model, hb3)
        assert res.call_overflowed(hb3)
        hb3.force_staged, hb3.defer_overflow_check = True, False
        np.testing.assert_allclose(res.encode(model, hb3).cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
        res.raise_on_overflow()
        res.close()
    finally:
        eng.close()


def test_two_rank_inferencer_redoes_overflowed_calls_on_every_rank(overflow_job, monkeypatch):
    """the same job on a graph hash-partitioned over two ranks (two processes on the test GPU, gloo + the callback
    transport): a call that overflows on EITHER rank is redone by both through the staged sharded plan with worst-case
    buckets; the union of the ranks' rows == the single-process rows (1e-5)"""
    import torch.multiprocessing as mp
    from gigl_amd import config
    from gigl_amd.inferencer import Inferencer
    base, n, *_ = overflow_job
    monkeypatch.setattr(config, "RECORDS_PER_PART_FILE", 3000)  # (what _infer_worker sets: the same planned root order)
    single = Inferencer().run("job", _variant(base, "configs/job.yaml", "w1"), None, uri_base=base, route="hbm")
    want = {r["node_id"]: np.array(r["emb"], np.float32) for r in _rows(single["embeddings"])}
    cfg2 = _variant(base, "configs/job.yaml", "w2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29840 + os.getpid() % 40
    procs = [ctx.Process(target=_infer_worker, args=(r, 2, port, base, cfg2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    got = {}
    for rank, status, out, n_rows in res:
        assert status == "ok", f"rank {rank}: {out}"
        for r in _rows(out["embeddings"]):
            assert r["node_id"] not in got
            got[r["node_id"]] = np.array(r["emb"], np.float32)
    assert sorted(got) == sorted(want)
    ids = sorted(want)
    g = np.stack([got[i] for i in ids])
    assert np.isfinite(g).all()
    np.testing.assert_allclose(g, np.stack([want[i] for i in ids]), rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------- training plans
def test_node_classification_plan_grows_and_redoes_the_queue_from_the_failed_batch():
    """SageTrainPlan.run_steps: six steps issued without a host read; the first batch fits, later ones do not.  The device
    halts the queue at the first failed batch (nothing behind it is applied), the plan grows (wide workspaces, Adam state
    adopted) and the steps are redone from there: loss history == the CPU restatement (oracle sample -> collate -> fp32
    autograd -> Adam) to 1e-4, trained weights too"""
    from gigl_amd.engine import HipEngine, SageTrainPlan
    from gigl_amd.models import GraphSAGE
    rowptr, col, x = _graph(24)
    steps, fan = 6, list(FAN)
    perm = np.random.default_rng(0).permutation(N)
    roots_np = perm[: steps * B].astype(np.uint32)
    labels_np = np.random.default_rng(1).integers(0, 7, roots_np.size)
    over = [_overflows(rowptr, col, roots_np[i * B:(i + 1) * B])[0] for i in range(steps)]
    assert not over[0] and any(over), over
    torch.manual_seed(2)
    model = GraphSAGE(24, 32, 7, num_layers=2)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=0.01, weight_decay=5e-4)
    want = []
    for i in range(steps):
        _, u = _overflows(rowptr, col, roots_np[i * B:(i + 1) * B])
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        out = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, params, 2)
        loss = torch.nn.functional.cross_entropy(out[torch.from_numpy(u["root_local"].astype(np.int64))],
                                                 torch.from_numpy(labels_np[i * B:(i + 1) * B]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        want.append(float(loss.detach()))
    eng = HipEngine(0)
    try:
        eng.load_csc(rowptr, col)
        eng.load_features(x)
        model = model.to(eng.device)
        st = torch.cuda.Stream(device=eng.device)
        torch.cuda.synchronize()
        eng.bind_stream(st)
        plan = SageTrainPlan(eng, model, B, fan, lr=0.01, weight_decay=5e-4)
        r_dev = torch.from_numpy(roots_np.view(np.int32)).to(eng.device)
        l_dev = torch.from_numpy(labels_np).to(eng.device)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            got = plan.run_steps(steps, lambda i: dict(roots=r_dev[i * B:(i + 1) * B], labels=l_dev[i * B:(i + 1) * B],
                                                       next_roots=r_dev[(i + 1) * B:(i + 2) * B]))
        eng.synchronize()
        got = got.cpu().numpy()
        assert plan.wide and plan.overflow_redone == 1
        plan.store(model)
        plan.close()
        eng.bind_stream(torch.cuda.current_stream(eng.device))
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
        for k, v in model.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), params[k].detach().numpy(), rtol=1e-3, atol=1e-4, err_msg=k)
    finally:
        eng.close()


@pytest.mark.parametrize("encoder", ["sage", "gat"])
def test_link_prediction_plans_grow_on_the_first_batch_that_does_not_fit(encoder):
    """NablpTrainPlan / GatNablpTrainPlan.step_checked on the same graph (anchors with their positives are each other's
    neighbours by construction): loss history == the CPU restatement of the step (oracle sample -> collate -> fp32 forward
    of both batches -> normalise -> scores -> retrieval loss -> autograd -> Adam) to 1e-4, no NaN, no raise"""
    from gigl_amd.engine import GatNablpTrainPlan, HipEngine, NablpTrainPlan
    from gigl_amd.models import GraphSAGE
    from gigl_amd.models_attn import GAT
    d = 64
    rowptr, col, x = _graph(d)
    b, P, n_rn, fan, steps, temp, heads = 48, 1, 32, list(FAN), 4, 0.07, 2
    eng = HipEngine(0)
    try:
        eng.load_csc(rowptr, col)
        eng.load_features(x)
        dst = np.repeat(np.arange(N, dtype=np.uint32), np.diff(rowptr).astype(np.int64))
        eng.build_from_coo(N, dst, col.astype(np.uint32), is_directed=True, out_graph=True)
        batches = _lp_batches(eng, N, b, P, n_rn, steps, seed=11)
        over = [roots.numel() and int(oracle.union_build(
            roots.cpu().numpy().view(np.uint32), fan,
            oracle.sample_khop(rowptr, col, roots.cpu().numpy().view(np.uint32), fan, canonical=True)[0])["meta"][3])
            > roots.numel() * (1 + fan[0]) for roots, _, _ in batches]
        assert any(over), over
        torch.manual_seed(6)
        if encoder == "sage":
            model = GraphSAGE(d, 32, 16, num_layers=2, should_l2_normalize_embedding_layer_output=True)
        else:
            model = GAT(d, 16, 32, num_layers=2, heads=heads, should_l2_normalize_embedding_layer_output=True)
        params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        opt = torch.optim.Adam(list(params.values()), lr=5e-3, weight_decay=1e-6)
        want = []
        for roots, cnt, rn in batches:
            embs = []
            for r in (roots, rn):
                r_h = r.cpu().numpy().view(np.uint32)
                nbr, _ = oracle.sample_khop(rowptr, col, r_h, fan, canonical=True)
                u = oracle.union_build(r_h, fan, nbr)
                ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
                h = torch.from_numpy(x[u["nodes"].astype(np.int64)])
                if encoder == "sage":
                    h = gnn_ref.graphsage_forward(h, ei, params, 2)
                else:
                    for l, hd in enumerate((heads, 1)):
                        p = f"conv_layers.{l}."
                        h = gnn_ref.gat_conv(h, ei, params[p + "lin.weight"], params[p + "att_src"], params[p + "att_dst"],
                                             params[p + "bias"], hd)
                        if l == 0:
                            h = torch.relu(h)
                h = torch.nn.functional.normalize(h, p=2, dim=1)
                embs.append(h[torch.from_numpy(u["root_local"].astype(np.int64))])
            loss = _lp_loss_torch(embs[0], embs[1], roots.cpu(), cnt.cpu(), rn.cpu(), cnt.numel(), P, temp)
            opt.zero_grad()
            loss.backward()
            opt.step()
            want.append(float(loss))
        lib = (GraphSAGE(d, 32, 16, num_layers=2, should_l2_normalize_embedding_layer_output=True) if encoder == "sage" else
               GAT(d, 16, 32, num_layers=2, heads=heads, should_l2_normalize_embedding_layer_output=True)).to(eng.device)
        lib.load_state_dict(model.state_dict())
        st = torch.cuda.Stream(device=eng.device)
        torch.cuda.synchronize()
        eng.bind_stream(st)
        cls = NablpTrainPlan if encoder == "sage" else GatNablpTrainPlan
        plan = cls(eng, lib, b, P, n_rn, fan, temperature=temp, lr=5e-3, weight_decay=1e-6)
        with torch.cuda.stream(st):
            got = [plan.step_checked(*bt, next_roots=(batches[i + 1][0], batches[i + 1][2]) if i + 1 < steps else None)
                   for i, bt in enumerate(batches)]
        eng.synchronize()
        assert plan.wide and plan.overflow_redone == 1
        plan.store(lib)
        plan.close()
        eng.bind_stream(torch.cuda.current_stream(eng.device))
        print(f"{encoder} link-prediction plan after growing: losses", got, "vs", want)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
        for k, v in lib.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), params[k].detach().numpy(), rtol=5e-3, atol=6e-3, err_msg=k)
    finally:
        eng.close()
