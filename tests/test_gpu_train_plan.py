"""The library's training step (gigl_sage_train_plan_*, csrc/pipeline.hip) against the autograd path it replaces and
against the CPU restatement of the reference's loop
(/root/reference/python/gigl/src/common/modeling_task_specs/node_classification_modeling_task_spec.py:134-173:
zero_grad -> model(inputs) -> F.cross_entropy(out[root_node_indices], labels) -> backward -> Adam(lr 0.01, wd 5e-4)):
the same batches from the same initial weights give the same loss history (1e-6 relative per step against the autograd
path on the device, 1e-4 against fp32 CPU autograd over oracle-collated batches) and the same trained weights."""
import numpy as np
import pytest
import torch

import oracle
from helpers import assert_adam_state, rmat_edges, torch_adam_moments
from oracle import gnn_ref


@pytest.fixture(scope="module")
def setup():
    from gigl_amd.engine import HipEngine
    s, d = rmat_edges(13, 150000, seed=8)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = (np.random.default_rng(0).standard_normal((n, 100)) / 4).astype(np.float32)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    yield eng, rowptr, col, x, n
    eng.close()


def _autograd_losses(eng, model, roots_all, labels_all, b, fan, steps, lr, wd):
    """the step GraphedTrainStep captures, eagerly: sample + union in HBM, forward with autograd, CE, backward, Adam"""
    import torch.nn.functional as F
    from gigl_amd.models import HipBatch
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd, foreach=False)  # (amsgrad-free, one tensor at a time)
    losses = []
    for i in range(steps):
        roots = roots_all[i * b:(i + 1) * b]
        tree = eng.sample_khop(roots, fan)
        u = eng.union_build(tree)
        out = model(HipBatch(eng, tree, u, train=True))
        loss = F.cross_entropy(out[u.root_local[: roots.numel()].long()], labels_all[i * b:(i + 1) * b])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses, torch_adam_moments(opt, dict(model.named_parameters()))


@pytest.mark.gpu
@pytest.mark.parametrize("dims,fan,b,prefetch", [((100, 64, 7), [10, 5], 256, False), ((100, 256, 47), [25, 10], 128, True),
                                                  ((100, 64, 7), [10, 5], 256, True)])
def test_library_training_step_equals_the_autograd_step(setup, dims, fan, b, prefetch):
    from gigl_amd.engine import SageTrainPlan
    from gigl_amd.models import GraphSAGE
    eng, rowptr, col, x, n = setup
    steps = 12
    rng = np.random.default_rng(3)
    roots_all = torch.from_numpy(rng.permutation(n)[: steps * b].astype(np.uint32).view(np.int32)).to(eng.device)
    labels_all = torch.from_numpy(rng.integers(0, dims[2], steps * b)).to(eng.device)
    torch.manual_seed(1)
    ref = GraphSAGE(dims[0], dims[1], dims[2], num_layers=2).to(eng.device)
    lib = GraphSAGE(dims[0], dims[1], dims[2], num_layers=2).to(eng.device)
    lib.load_state_dict(ref.state_dict())
    ref.train()
    want, ref_moments = _autograd_losses(eng, ref, roots_all, labels_all, b, fan, steps, 0.01, 5e-4)
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = SageTrainPlan(eng, lib, b, fan, lr=0.01, weight_decay=5e-4)
    got = []
    with torch.cuda.stream(st):  # (the loss is a device scalar written on the plan's stream)
        for i in range(steps):  # (the parts run eagerly once, are captured on their second run, then replayed)
            # every other step hands the plan the next batch's roots: its graph part then overlaps this step's layers
            nxt = roots_all[(i + 1) * b:(i + 2) * b] if (prefetch and i + 1 < steps and i % 3 != 2) else None
            got.append(plan.step(roots_all[i * b:(i + 1) * b], labels_all[i * b:(i + 1) * b], next_roots=nxt).clone())
    eng.synchronize()
    got = [float(v) for v in got]
    plan.store(lib)
    moments = plan.moments()
    plan.close()
    eng.bind_stream(torch.cuda.current_stream(eng.device))
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    # trained state: Adam's moments everywhere, the parameters where the gradients (not their rounding) decide the direction
    # (tol_p: 1.1e-5 seen once in ~15 runs at 1e-5 — the autograd side sums its gradients with float atomics)
    assert_adam_state("node-classification plan vs autograd", lib.state_dict(), moments, ref.state_dict(), ref_moments,
                      tol_m=2e-5, tol_v=2e-5, tol_p=2e-5)


@pytest.mark.gpu
def test_library_training_step_against_the_cpu_restatement(setup):
    """oracle sample -> collate -> fp32 CPU autograd -> Adam, five steps; a short last batch is padded and masked"""
    from gigl_amd.engine import SageTrainPlan
    from gigl_amd.models import GraphSAGE
    eng, rowptr, col, x, n = setup
    b, fan, steps = 64, [10, 5], 5
    rng = np.random.default_rng(9)
    roots_np = rng.permutation(n)[: steps * b - 20].astype(np.uint32)  # (the last batch has 44 real roots)
    labels_np = rng.integers(0, 7, roots_np.size)
    torch.manual_seed(2)
    model = GraphSAGE(100, 32, 7, num_layers=2)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=0.01, weight_decay=5e-4, foreach=False)
    want = []
    for lo in range(0, roots_np.size, b):
        roots = roots_np[lo:lo + b]
        nbr, _ = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
        u = oracle.union_build(roots, fan, nbr)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        out = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, params, 2)
        loss = torch.nn.functional.cross_entropy(out[torch.from_numpy(u["root_local"].astype(np.int64))],
                                                 torch.from_numpy(labels_np[lo:lo + b]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        want.append(float(loss.detach()))
    model = model.to(eng.device)
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = SageTrainPlan(eng, model, b, fan, lr=0.01, weight_decay=5e-4)
    r_dev = torch.from_numpy(roots_np.view(np.int32)).to(eng.device)
    l_dev = torch.from_numpy(labels_np).to(eng.device)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        got = [plan.step(r_dev[lo:lo + b], l_dev[lo:lo + b]).clone() for lo in range(0, roots_np.size, b)]
    eng.synchronize()
    got = [float(v) for v in got]
    plan.store(model)
    moments = plan.moments()
    plan.close()
    eng.bind_stream(torch.cuda.current_stream(eng.device))
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    assert_adam_state("node-classification plan vs the CPU restatement", model.state_dict(), moments, params,
                      torch_adam_moments(opt, params), tol_m=1e-4, tol_v=1e-4, tol_p=1e-4)


def _lp_batches(eng, n, b, P, n_rn, steps, seed):
    """main roots (anchor-major: anchor + its P positive slots, a missing positive repeats the anchor), positives per
    anchor and random-negative roots of `steps` batches — drawn as the link-prediction trainer's in-HBM route draws them
    (engine.sample_positives over the out-graph)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(steps):
        anchors = torch.from_numpy(rng.permutation(n)[:b].astype(np.uint32).view(np.int32)).to(eng.device)
        pos, cnt = eng.sample_positives(anchors, P, sampling_seed=42)
        ar = torch.arange(P, device=eng.device).view(1, P)
        a2 = anchors.view(-1, 1)
        grouped = torch.where(ar < cnt.view(-1, 1), pos.view(-1, P), a2.expand(-1, P))
        roots = torch.cat([a2, grouped], dim=1).reshape(-1).contiguous()
        rn = torch.from_numpy(rng.permutation(n)[:n_rn].astype(np.uint32).view(np.int32)).to(eng.device)
        out.append((roots, cnt.to(torch.int32).contiguous(), rn))
    return out


def _lp_loss_torch(emb_main, emb_rn, roots, cnt, rn, b, P, temperature):
    """infer_task_inputs + Retrieval on embeddings, in torch (utils/infer.py; loss.py:209-331 restated row by row in
    oracle/gnn_ref.retrieval_loss_rows): repeated queries x cat(positives, random negatives), both masks, CE / rows"""
    T = 1 + P
    ids = (roots.to(torch.int64) & 0xFFFFFFFF).view(b, T)
    k = cnt.to(torch.int64)
    slot = torch.arange(P, device=roots.device).view(1, P)
    ok = (slot < k.view(-1, 1)).reshape(-1)
    q_rows = (torch.arange(b, device=roots.device) * T).repeat_interleave(P)[ok]
    p_rows = (torch.arange(b, device=roots.device).view(-1, 1) * T + 1 + slot).reshape(-1)[ok]
    rq, pos = emb_main[q_rows], emb_main[p_rows]
    cand = torch.cat([pos, emb_rn])
    scores = rq @ cand.T / temperature
    qid = ids[:, 0].repeat_interleave(P)[ok]
    cid = torch.cat([ids.reshape(-1)[p_rows], rn.to(torch.int64) & 0xFFFFFFFF])
    Q, Cn = scores.shape
    eye = torch.zeros((Q, Cn), dtype=torch.bool, device=scores.device)
    eye[torch.arange(Q), torch.arange(Q)] = True
    same_q = torch.zeros_like(eye)
    same_q[:, :Q] = qid.view(-1, 1) == qid.view(1, -1)
    hit = cid.view(1, -1) == cid[:Q].view(-1, 1)
    masked = scores.masked_fill((same_q | hit) & ~eye, torch.finfo(torch.float32).min)
    return torch.nn.functional.cross_entropy(masked, torch.arange(Q, device=scores.device), reduction="sum") / max(Q, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,fan,b,P,n_rn,norm,prefetch", [((100, 32, 16), [10, 5], 128, 1, 64, True, False),
                                                             ((100, 64, 32), [10, 5], 96, 2, 50, True, True),
                                                             ((100, 32, 16), [10, 5], 128, 1, 64, True, True),
                                                             ((100, 256, 128), [25, 10], 64, 1, 40, False, False)])
def test_library_link_prediction_step_equals_the_autograd_step(setup, dims, fan, b, P, n_rn, norm, prefetch):
    """gigl_nablp_train_plan_* (one library call per link-prediction training step: two encodes, inner-product scores,
    retrieval loss, backward of both encodes, Adam) against the autograd step over the same in-HBM batches — the loop body
    of node_anchor_based_link_prediction_modeling_task_spec.py:334-451 with the reference's defaults (GraphSAGE encoder,
    L2-normalised embeddings, temperature 0.07, accidental-hit removal, Adam lr 5e-3 wd 1e-6): same loss history, same
    trained weights; anchors with fewer than P positives and hipGraph replay included.  prefetch: most steps announce the
    next batch's roots (its sampling + union then run on the plan's side stream beside the step's layers), some do not,
    and one announces a batch that is then NOT the one trained on (the plan must sample again)"""
    from gigl_amd.engine import NablpTrainPlan
    from gigl_amd.models import GraphSAGE, HipBatch
    eng, rowptr, col, x, n = setup
    if getattr(eng, "_graph_out", None) is None:
        dst = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr).astype(np.int64))
        eng.build_from_coo(n, dst, col.astype(np.uint32), is_directed=True, out_graph=True)  # (out-edges = reversed in-edges)
    steps, temp = 8, 0.07
    batches = _lp_batches(eng, n, b, P, n_rn, steps, seed=5)
    if P > 1:
        assert any(int((c < P).sum()) > 0 for _, c, _ in batches)  # (some anchors have fewer than P positives)
    torch.manual_seed(4)
    kw = dict(num_layers=2, should_l2_normalize_embedding_layer_output=norm)
    ref = GraphSAGE(dims[0], dims[1], dims[2], **kw).to(eng.device)
    lib = GraphSAGE(dims[0], dims[1], dims[2], **kw).to(eng.device)
    lib.load_state_dict(ref.state_dict())
    ref.train()
    opt = torch.optim.Adam(ref.parameters(), lr=5e-3, weight_decay=1e-6, foreach=False)
    want = []
    for roots, cnt, rn in batches:
        embs = []
        for r in (roots, rn):
            tree = eng.sample_khop(r, fan)
            u = eng.union_build(tree)
            embs.append(ref(HipBatch(eng, tree, u, train=True))[u.root_local[: r.numel()].long()])
        loss = _lp_loss_torch(embs[0], embs[1], roots, cnt, rn, b, P, temp)
        opt.zero_grad()
        loss.backward()
        opt.step()
        want.append(float(loss))
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = NablpTrainPlan(eng, lib, b, P, n_rn, fan, temperature=temp, remove_accidental_hits=True, lr=5e-3, weight_decay=1e-6)
    got = []
    with torch.cuda.stream(st):
        for i, (roots, cnt, rn) in enumerate(batches):  # (eager once, captured on the second step, replayed from then on)
            nxt = None
            if prefetch and i + 1 < steps and i % 4 != 2:
                j = i + 1 if i != 4 else 0  # (step 4 announces the wrong batch)
                nxt = (batches[j][0], batches[j][2])
            got.append(plan.step(roots, cnt, rn, next_roots=nxt).clone())
    eng.synchronize()
    rows = [float(v[1]) for v in got]
    got = [float(v[0]) for v in got]
    plan.store(lib)
    moments = plan.moments()
    plan.close()
    eng.bind_stream(torch.cuda.current_stream(eng.device))
    assert rows == [float(int(c.clamp(max=P).sum())) for _, c, _ in batches]
    # (without the normalisation the logits are unbounded and the loss climbs at this learning rate: a step's rounding is
    # amplified by the next ones)
    np.testing.assert_allclose(got, want, rtol=2e-5 if norm else 3e-4, atol=2e-6)
    assert_adam_state("link-prediction plan vs autograd", lib.state_dict(), moments, ref.state_dict(),
                      torch_adam_moments(opt, dict(ref.named_parameters())), tol_m=1e-4 if norm else 3e-2,
                      tol_v=1e-4 if norm else 1e-2, tol_p=1e-4 if norm else 5e-3)  # (not normalised: the climbing loss
    # amplifies every step's rounding — the loss history itself is only held to 3e-4 there)


@pytest.mark.gpu
def test_library_link_prediction_step_against_the_cpu_restatement(setup):
    """the same step on the CPU: oracle sample -> collate -> fp32 forward of both batches (gnn_ref, every layer over the
    whole union graph) -> normalise -> scores -> row-wise retrieval loss -> torch autograd -> Adam; a short batch (fewer
    anchors and negatives than the plan's capacity) is padded and masked"""
    from gigl_amd.engine import NablpTrainPlan
    from gigl_amd.models import GraphSAGE
    eng, rowptr, col, x, n = setup
    if getattr(eng, "_graph_out", None) is None:
        dst = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr).astype(np.int64))
        eng.build_from_coo(n, dst, col.astype(np.uint32), is_directed=True, out_graph=True)
    b, P, n_rn, fan, steps, temp = 48, 1, 32, [10, 5], 4, 0.07
    batches = _lp_batches(eng, n, b, P, n_rn, steps, seed=11)
    batches[-1] = (batches[-1][0][: 2 * 30].contiguous(), batches[-1][1][:30].contiguous(), batches[-1][2][:20].contiguous())
    torch.manual_seed(6)
    model = GraphSAGE(100, 32, 16, num_layers=2, should_l2_normalize_embedding_layer_output=True)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=5e-3, weight_decay=1e-6, foreach=False)
    want = []
    for roots, cnt, rn in batches:
        embs = []
        for r in (roots, rn):
            r_h = r.cpu().numpy().view(np.uint32)
            nbr, _ = oracle.sample_khop(rowptr, col, r_h, fan, canonical=True)
            u = oracle.union_build(r_h, fan, nbr)
            ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
            out = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, params, 2)
            out = torch.nn.functional.normalize(out, p=2, dim=1)
            embs.append(out[torch.from_numpy(u["root_local"].astype(np.int64))])
        na = cnt.numel()
        loss = _lp_loss_torch(embs[0], embs[1], roots.cpu(), cnt.cpu(), rn.cpu(), na, P, temp)
        opt.zero_grad()
        loss.backward()
        opt.step()
        want.append(float(loss))
    lib = GraphSAGE(100, 32, 16, num_layers=2, should_l2_normalize_embedding_layer_output=True).to(eng.device)
    lib.load_state_dict(model.state_dict())
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = NablpTrainPlan(eng, lib, b, P, n_rn, fan, temperature=temp, lr=5e-3, weight_decay=1e-6)
    with torch.cuda.stream(st):
        got = [plan.step(*bt).clone() for bt in batches]
    eng.synchronize()
    got = [float(v[0]) for v in got]
    plan.store(lib)
    moments = plan.moments()
    plan.close()
    eng.bind_stream(torch.cuda.current_stream(eng.device))
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    assert_adam_state("link-prediction plan vs the CPU restatement", lib.state_dict(), moments, params,
                      torch_adam_moments(opt, params), tol_m=1e-3, tol_v=1e-3, tol_p=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("d,aggr", [(256, "mean"), (128, "mean"), (64, "sum"), (20, "mean"), (300, "sum")])
def test_transposed_backward_gather_equals_the_scatter_and_a_dense_restatement(d, aggr):
    """gigl_gather_mean_backward_transposed (every source row written once by a gather over the transposed rows) against
    the atomics scatter it replaces in the training plans and against a dense fp64 restatement: random ragged rows, sources
    shared by many rows, sources nobody reads, empty rows; rows past *n_src stay untouched"""
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    st = torch.cuda.Stream()
    eng.bind_stream(st)
    rng = np.random.default_rng(d)
    n_rows, n_src, cap_src = 700, 9000, 9500
    deg = rng.integers(0, 40, n_rows)
    deg[:5] = 0
    rowptr = np.zeros(n_rows + 1, np.int32)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n_src, int(rowptr[-1])).astype(np.int32)
    col[: 200] = rng.integers(0, 7, 200)  # a few sources read by very many rows
    dout = rng.standard_normal((n_rows, 2 * d)).astype(np.float32)
    want = np.zeros((n_src, d), np.float64)
    want[:n_rows] += dout[:, d:]
    for i in range(n_rows):
        if deg[i]:
            np.add.at(want, col[rowptr[i]:rowptr[i + 1]], dout[i, :d].astype(np.float64) / (deg[i] if aggr == "mean" else 1.0))
    dev = eng.device
    t = lambda a: torch.from_numpy(a).to(dev)
    with torch.cuda.stream(st):
        rp, cl, do = t(rowptr), t(col), t(dout)
        nr = torch.tensor([n_rows], dtype=torch.int32, device=dev)
        ns = torch.tensor([n_src], dtype=torch.int32, device=dev)
        got = torch.full((cap_src, d), 7.0, dtype=torch.float32, device=dev)
        eng.gather_mean_backward_transposed(do, d, rp, None, cl, nr, n_rows, ns, got, aggr=aggr)
        ref = torch.zeros((cap_src, d), dtype=torch.float32, device=dev)
        eng.gather_mean_backward(do, d, rp, None, cl, nr, n_rows, ref, aggr=aggr)
    st.synchronize()
    g = got.cpu().numpy()
    np.testing.assert_allclose(g[:n_src], want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(g[:n_src], ref.cpu().numpy()[:n_src], rtol=2e-5, atol=2e-5)
    assert (g[n_src:] == 7.0).all()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("heads,hid,out,norm,dtype", [(2, 16, 32, True, torch.float32), (1, 32, 16, True, torch.float16),
                                                       (4, 8, 16, False, torch.float32)])
def test_library_gat_link_prediction_step_equals_the_autograd_step(setup, heads, hid, out, norm, dtype):
    """gigl_gat_nablp_train_plan_* — the link-prediction training step with configs[4]'s encoder (two GATConv layers, the
    first from the input side) as one library call per step — against the autograd step over the same batches handed to
    models_attn.GAT as device-built batch graphs (hbm.ResidentGraph.train_graph -> GAT._forward_graph_input_side): the
    first step's parameter gradients (all eight tensors), then the loss history and the trained parameters over several
    steps with prefetch, hipGraph replay included"""
    from gigl_amd.engine import GatNablpTrainPlan, HipEngine
    from gigl_amd.hbm import ResidentGraph
    from gigl_amd.models_attn import GAT
    _, rowptr, col, x, n = setup
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(torch.from_numpy(x).to(dtype))
    dst = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr).astype(np.int64))
    eng.build_from_coo(n, dst, col.astype(np.uint32), is_directed=True, out_graph=True)
    fan, b, P, n_rn, steps, temp = [10, 5], 96, 1, 40, 6, 0.07
    batches = _lp_batches(eng, n, b, P, n_rn, steps, seed=9)
    torch.manual_seed(6)
    kw = dict(num_layers=2, heads=heads, should_l2_normalize_embedding_layer_output=norm)
    ref = GAT(100, hid, out, **kw).to(eng.device)
    lib = GAT(100, hid, out, **kw).to(eng.device)
    lib.load_state_dict(ref.state_dict())
    ref.train()
    ref.engine = eng
    res = ResidentGraph.from_engine(eng, np.arange(n, dtype=np.int64), fan)
    res.train_as_graph_data, res.defer_x = True, True
    opt = torch.optim.Adam(ref.parameters(), lr=5e-3, weight_decay=1e-6, foreach=False)
    want, first_grads = [], None
    for roots, cnt, rn in batches:
        embs = []
        for r in (roots, rn):
            g, ri = res.train_graph(r)
            embs.append(ref(g)[ri])
        loss = _lp_loss_torch(embs[0], embs[1], roots, cnt, rn, b, P, temp)
        opt.zero_grad()
        loss.backward()
        if first_grads is None:
            first_grads = [[c.lin.weight.grad.clone(), c.att_src.grad.reshape(-1).clone(), c.att_dst.grad.reshape(-1).clone(),
                            c.bias.grad.clone()] for c in ref.conv_layers]
        opt.step()
        want.append(float(loss))
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = GatNablpTrainPlan(eng, lib, b, P, n_rn, fan, temperature=temp, remove_accidental_hits=True, lr=5e-3, weight_decay=1e-6)
    got = []
    with torch.cuda.stream(st):
        for i, (roots, cnt, rn) in enumerate(batches):
            nxt = (batches[i + 1][0], batches[i + 1][2]) if i + 1 < steps and i != 2 else None
            got.append(plan.step(roots, cnt, rn, next_roots=nxt).clone())
            if i == 0:
                grads = [plan.grads(l) for l in range(2)]
                again = [plan.grads(l) for l in range(2)]  # (the partial sums are added to the buffers once, not per request)
            if i == steps - 1:
                last = [plan.grads(l) for l in range(2)]   # (after replayed steps: the hook still finds this step's sums)
    eng.synchronize()
    for l in range(2):
        for a, a2, z in zip(grads[l], again[l], last[l]):
            assert torch.equal(a, a2) and torch.isfinite(z).all() and float(z.abs().max()) > 0
    errs = {}
    for l in range(2):
        for name, a, w_ in zip(("w", "att_src", "att_dst", "bias"), grads[l], first_grads[l]):
            errs[f"layer {l} d {name}"] = float((a - w_).abs().max()) / (float(w_.abs().max()) + 1e-12)
    got = [float(v[0]) for v in got]
    print("GAT plan: first-step gradient errors (max |err| / max |grad|):", {k: f"{v:.2e}" for k, v in errs.items()},
          "| loss", got[0], "vs", want[0])
    assert max(errs.values()) < 2e-4 and abs(got[0] - want[0]) < 1e-4 * abs(want[0]), (errs, got[0], want[0])
    plan.store(lib)
    moments = plan.moments()
    plan.close()
    eng.bind_stream(torch.cuda.current_stream(eng.device))
    np.testing.assert_allclose(got, want, rtol=1e-4 if norm else 1e-3, atol=1e-5)
    flat = lambda sd: {k: v.reshape(-1) for k, v in sd.items()}
    assert_adam_state("GAT link-prediction plan vs autograd", flat(lib.state_dict()), moments, flat(ref.state_dict()),
                      {k: (m.reshape(-1), v.reshape(-1)) for k, (m, v) in
                       torch_adam_moments(opt, dict(ref.named_parameters())).items()},
                      tol_m=1e-3, tol_v=1e-3, tol_p=1e-4)
    eng.close()


@pytest.mark.gpu
def test_library_gat_link_prediction_step_against_the_cpu_restatement(setup):
    """the GAT link-prediction step on the CPU: oracle sample -> collate -> fp32 GAT forward of both batches
    (oracle/gnn_ref.gat_conv, every layer over the WHOLE union graph, projection first — the reference's execution order)
    -> normalise -> scores -> retrieval loss -> torch autograd -> Adam; a short last batch is padded and masked.  The
    library runs the first layer from the input side for the nodes of level <= 1 only: same function of the parameters."""
    from gigl_amd.engine import GatNablpTrainPlan, HipEngine
    from gigl_amd.models_attn import GAT
    _, rowptr, col, x, n = setup
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    dst = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr).astype(np.int64))
    eng.build_from_coo(n, dst, col.astype(np.uint32), is_directed=True, out_graph=True)
    b, P, n_rn, fan, steps, temp, heads = 48, 1, 32, [10, 5], 4, 0.07, 2
    batches = _lp_batches(eng, n, b, P, n_rn, steps, seed=13)
    batches[-1] = (batches[-1][0][: 2 * 30].contiguous(), batches[-1][1][:30].contiguous(), batches[-1][2][:20].contiguous())
    torch.manual_seed(8)
    model = GAT(100, 16, 32, num_layers=2, heads=heads, should_l2_normalize_embedding_layer_output=True)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=5e-3, weight_decay=1e-6, foreach=False)
    want = []
    for roots, cnt, rn in batches:
        embs = []
        for r in (roots, rn):
            r_h = r.cpu().numpy().view(np.uint32)
            nbr, _ = oracle.sample_khop(rowptr, col, r_h, fan, canonical=True)
            u = oracle.union_build(r_h, fan, nbr)
            ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
            h = torch.from_numpy(x[u["nodes"].astype(np.int64)])
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
    lib = GAT(100, 16, 32, num_layers=2, heads=heads, should_l2_normalize_embedding_layer_output=True).to(eng.device)
    lib.load_state_dict(model.state_dict())
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    plan = GatNablpTrainPlan(eng, lib, b, P, n_rn, fan, temperature=temp, lr=5e-3, weight_decay=1e-6)
    with torch.cuda.stream(st):
        got = [plan.step(*bt).clone() for bt in batches]
    eng.synchronize()
    got = [float(v[0]) for v in got]
    plan.store(lib)
    moments = plan.moments()
    plan.close()
    print("GAT link-prediction plan vs the CPU restatement: losses", got, "vs", want)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    flat = lambda sd: {k: v.reshape(-1) for k, v in sd.items()}
    assert_adam_state("GAT link-prediction plan vs the CPU restatement", flat(lib.state_dict()), moments, flat(params),
                      {k: (m.reshape(-1), v.reshape(-1)) for k, (m, v) in torch_adam_moments(opt, params).items()},
                      tol_m=1e-3, tol_v=1e-3, tol_p=1e-4)
    eng.close()
