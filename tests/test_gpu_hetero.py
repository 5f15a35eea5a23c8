"""HGT / SimpleHGN on the device (csrc/hetero.hip + gigl_linear) vs the edge-list restatements in oracle/gnn_ref.py of
HGTConv (python/gigl/src/common/models/pyg/nn/conv/hgt_conv.py) and SimpleHGNConv (.../simplehgn_conv.py), 1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gnn_ref

pytestmark = pytest.mark.gpu

NT = {"user": (300, 12), "item": (200, 20), "tag": (50, 8)}
ET = [("user", "follows", "user"), ("user", "buys", "item"), ("item", "bought_by", "user"), ("tag", "labels", "item")]


def make_data(seed=0, with_edge_attr=False):
    from gigl_amd.models_hetero import HeteroGraphData
    g = torch.Generator().manual_seed(seed)
    x = {t: torch.randn(n, d, generator=g) for t, (n, d) in NT.items()}
    ei, ea = {}, {}
    for (s, r, d), e in zip(ET, (2500, 1800, 1700, 300)):
        src = torch.randint(0, NT[s][0], (e,), generator=g)
        # skewed destinations: hubs and nodes without in-edges both occur
        dst = (torch.rand(e, generator=g) ** 2 * NT[d][0]).long().clamp(max=NT[d][0] - 1)
        ei[(s, r, d)] = torch.stack([src, dst])
        if with_edge_attr:
            ea[(s, r, d)] = torch.randn(e, 5, generator=g)
    return HeteroGraphData(x, ei, ea)


def test_hgt_matches_the_restatement():
    from gigl_amd.models_hetero import HGT
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    heads, hid, out_dim = 4, 64, 32
    model = HGT({t: d for t, (_, d) in NT.items()}, {e: 0 for e in ET}, hid_dim=hid, out_dim=out_dim, num_layers=2,
                num_heads=heads)
    with torch.no_grad():  # non-trivial gates / priors
        for conv in model.convs:
            for p in conv.skip.values():
                p.uniform_(-1, 1)
            for p in conv.p_rel.values():
                p.uniform_(0.5, 1.5)
    data = make_data()
    got = model.to(dev)(data.to(dev), ["user", "item", "tag"])
    # inference (no autograd) runs over COMPOSED weights (HGTConv._forward_composed: k_rel(K(x)), v_rel(V(x)) and the
    # gate's scale of out_lin multiplied out once per parameter state); composed_inference = False is the staged order
    with torch.no_grad():
        got_c = model(data.to(dev), ["user", "item", "tag"])
        for conv in model.convs:
            conv.composed_inference = False
        got_s = model(data.to(dev), ["user", "item", "tag"])
        for conv in model.convs:
            conv.composed_inference = True
        # a parameter update invalidates the composed weights
        model.convs[0].k_rel.weight.mul_(1.5)
        moved = model(data.to(dev), ["user", "item", "tag"])
        model.convs[0].k_rel.weight.div_(1.5)
        back = model(data.to(dev), ["user", "item", "tag"])
    for t in NT:
        assert torch.equal(got_s[t], got[t].detach())
        assert not torch.allclose(moved[t], got_c[t], atol=1e-4) or t == "tag"
        np.testing.assert_allclose(back[t].cpu().numpy(), got_c[t].cpu().numpy(), rtol=1e-5, atol=1e-5)
    # the same forward from the parameters, on the CPU
    model = model.cpu()
    h = {t: torch.relu(F.linear(x, model.lin_dict[t].weight, model.lin_dict[t].bias)) for t, x in data.x_dict.items()}
    for conv in model.convs:
        p = dict(kqv={t: (conv.kqv_lin.lins[t].weight, conv.kqv_lin.lins[t].bias) for t in NT},
                 out={t: (conv.out_lin.lins[t].weight, conv.out_lin.lins[t].bias) for t in NT},
                 k_rel=conv.k_rel.weight, v_rel=conv.v_rel.weight, skip={t: conv.skip[t] for t in NT},
                 p_rel={e: conv.p_rel["__".join(e)] for e in ET}, edge_types=ET)
        with torch.no_grad():
            h = gnn_ref.hgt_conv(h, data.edge_index_dict, p, heads)
    for t in NT:
        want = F.linear(h[t], model.lin.weight, model.lin.bias).detach().numpy()
        np.testing.assert_allclose(got[t].detach().cpu().numpy(), want, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got_c[t].cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # a node type nobody points at still comes back (the in-repo modification of PyG's HGTConv)
    assert got["tag"].shape == (50, out_dim)


@pytest.mark.parametrize("with_edge_attr", [False, True])
def test_simplehgn_matches_the_restatement(with_edge_attr):
    from gigl_amd.models_hetero import SimpleHGN
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    heads, hid, out_dim = 2, 32, 24
    model = SimpleHGN({t: d for t, (_, d) in NT.items()}, {e: (5 if with_edge_attr else 0) for e in ET}, node_hid_dim=hid,
                      edge_hid_dim=8, edge_type_dim=16, node_out_dim=out_dim, num_layers=2, num_heads=heads)
    data = make_data(3, with_edge_attr)
    got = model.to(dev)(data.to(dev), ["user", "item"])
    model = model.cpu()
    off, n, xs = {}, 0, []
    for t, x in data.x_dict.items():
        lin = model.node_type_lin_dict[t]
        xs.append(F.linear(x, lin.weight, lin.bias))
        off[t] = n
        n += x.shape[0]
    h = torch.cat(xs)
    ei = torch.cat([e + torch.tensor([[off[k[0]]], [off[k[2]]]]) for k, e in data.edge_index_dict.items()], dim=1)
    ety = torch.cat([torch.full((e.shape[1],), i) for i, e in enumerate(data.edge_index_dict.values())])
    ef = None
    if with_edge_attr:
        ef = torch.cat([F.linear(data.edge_attr_dict[k], model.edge_type_lin_dict[f"{k[0]}-{k[1]}-{k[2]}"].weight,
                                 model.edge_type_lin_dict[f"{k[0]}-{k[1]}-{k[2]}"].bias) for k in data.edge_index_dict])
    with torch.no_grad():
        for i, conv in enumerate(model.convs):
            p = dict(W_nfeat=conv.W_nfeat, a_l=conv.a_l, a_r=conv.a_r, a_etype=conv.a_etype,
                     edge_type_emb=conv.edge_type_emb, W_etype=(conv.W_etype.weight, conv.W_etype.bias),
                     residual=(conv.residual.weight, conv.residual.bias))
            if with_edge_attr:
                p.update(W_efeat=conv.W_efeat, a_efeat=conv.a_efeat)
            h = gnn_ref.simplehgn_conv(ei, h, ety, p, heads, hid, edge_feat=ef)
            if i != len(model.convs) - 1:
                h = F.elu(h)
        emb = F.linear(h, model.lin.weight, model.lin.bias)
    for t in ("user", "item"):
        want = emb[off[t]: off[t] + NT[t][0]].numpy()
        np.testing.assert_allclose(got[t].detach().cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    with pytest.raises(ValueError):
        model.to(dev)(data.to(dev), ["nobody"])


def test_attention_kernels_reject_unsupported_shapes():
    from gigl_amd._lib import GiglError
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    z = torch.zeros((4, 6 * 5), device=eng.device)
    rp = torch.zeros(5, dtype=torch.int32, device=eng.device)
    col = torch.zeros(1, dtype=torch.int32, device=eng.device)
    with pytest.raises(GiglError):
        eng.hgt_aggregate(z, z, z, 6, 5, rp, col, None, None, 4, z.clone())  # dim % 4 != 0
    eng.close()


@pytest.mark.parametrize("route", ["staged", "one-call plan"])
def test_dag_sampler_to_hgt_end_to_end(route):
    """typed samples stay in HBM: SamplingOp-DAG sampler -> typed batch graph -> HGT; the batch graph equals the union
    of the per-root restatement's samples (oracle/dag_sampler.py) and the root embeddings equal the CPU forward over it"""
    from gigl_amd.graphdb_sampler import (INCOMING, OUTGOING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG)
    from gigl_amd.models_hetero import HGT
    from oracle import dag_sampler
    A2P, P2A = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
    node_types, cet = {"author": 0, "paper": 1}, {A2P: 0, P2A: 1}
    rng = np.random.default_rng(0)
    n = {"author": 800, "paper": 1200}
    a = (rng.zipf(1.7, 6000) % n["author"]).astype(np.uint32)
    p = rng.integers(0, n["paper"], 6000).astype(np.uint32)
    edges = {A2P: (a, p), P2A: (p, a)}
    feats = {"author": rng.standard_normal((800, 6)).astype(np.float32), "paper": rng.standard_normal((1200, 10)).astype(np.float32)}
    s = HipGraphDBSampler(node_types, n, edges, cet, feats)
    ops = [SamplingOp("op0", A2P, 4, [], INCOMING), SamplingOp("op1", A2P, 3, ["op0"], OUTGOING),
           SamplingOp("op2", P2A, 2, ["op1"], OUTGOING)]
    dag = SamplingOpDAG.from_ops(ops)
    roots = rng.integers(0, n["paper"], 64)
    # (staged: one library call per op + torch.unique chains; one-call plan: gigl_typed_plan_*, csrc/typed_plan.hip)
    build = s.batch_graph if route == "staged" else s.batch_graph_plan
    data, root_index, uniq = build(roots, "paper", dag)
    s.engine.synchronize()
    # == union of the per-root samples
    nbrs = dag_sampler.neighbour_lists(edges)
    want_e, want_n = set(), set()
    for r in roots:
        e_, n_ = dag_sampler.sample_for_root(int(r), ops, nbrs, node_types, cet, "paper")
        want_e |= e_
        want_n |= n_
    got_n = {(int(v), node_types[t]) for t, u in uniq.items() for v in u.cpu().tolist()}
    assert got_n == want_n
    got_e = set()
    for (st_, rel, dt_), ei in data.edge_index_dict.items():
        c = cet[A2P] if rel == "writes" else cet[P2A]
        for sl, dl in ei.t().cpu().tolist():
            got_e.add((int(uniq[st_][sl]), int(uniq[dt_][dl]), c))
    assert got_e == want_e
    for t in uniq:
        np.testing.assert_array_equal(data.x_dict[t].cpu().numpy(), feats[t][uniq[t].cpu().numpy()])
    assert torch.equal(uniq["paper"][root_index].cpu(), torch.from_numpy(roots))
    # encoder over it
    torch.manual_seed(5)
    ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
    model = HGT({"author": 6, "paper": 10}, {e: 0 for e in ets}, hid_dim=32, out_dim=16, num_layers=2, num_heads=2)
    model.engine = s.engine
    with torch.cuda.stream(s.engine._stream):
        got = model.to(s.engine.device)(data, ["paper"])["paper"][root_index]
    s.engine.synchronize()
    model = model.cpu()
    xd = {t: x.cpu() for t, x in data.x_dict.items()}
    eid = {k: v.cpu() for k, v in data.edge_index_dict.items()}
    h = {t: torch.relu(F.linear(x, model.lin_dict[t].weight, model.lin_dict[t].bias)) for t, x in xd.items()}
    with torch.no_grad():
        for conv in model.convs:
            pr = dict(kqv={t: (conv.kqv_lin.lins[t].weight, conv.kqv_lin.lins[t].bias) for t in xd},
                      out={t: (conv.out_lin.lins[t].weight, conv.out_lin.lins[t].bias) for t in xd},
                      k_rel=conv.k_rel.weight, v_rel=conv.v_rel.weight, skip={t: conv.skip[t] for t in xd},
                      p_rel={e: conv.p_rel["__".join(e)] for e in ets}, edge_types=ets)
            h = gnn_ref.hgt_conv(h, eid, pr, 2)
        want = F.linear(h["paper"], model.lin.weight, model.lin.bias)[root_index.cpu()]
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-5)
    s.close()


def test_typed_records_to_hgt_through_the_trainer_side_loader(tmp_path):
    """the two halves as the reference wires them: the typed sampler writes TFRecords of RootedNodeNeighborhood
    (encoded on the device), the trainer-side loader collates them natively into a typed batch graph and HGT embeds the
    roots — the same embeddings as over the batch graph that never left HBM"""
    from gigl_amd import wire
    from gigl_amd.batches import HeteroRootedNodeNeighborhoodBatch
    from gigl_amd.graphdb_sampler import (INCOMING, OUTGOING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG)
    from gigl_amd.models_hetero import HGT
    A2P, P2A = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
    node_types, cet = {"author": 0, "paper": 1}, {A2P: 0, P2A: 1}
    rng = np.random.default_rng(3)
    n = {"author": 500, "paper": 900}
    a = (rng.zipf(1.7, 4000) % n["author"]).astype(np.uint32)
    p = rng.integers(0, n["paper"], 4000).astype(np.uint32)
    feats = {"author": rng.standard_normal((500, 6)).astype(np.float32), "paper": rng.standard_normal((900, 10)).astype(np.float32)}
    s = HipGraphDBSampler(node_types, n, {A2P: (a, p), P2A: (p, a)}, cet, feats)
    dag = SamplingOpDAG.from_ops([SamplingOp("op0", A2P, 4, [], INCOMING), SamplingOp("op1", A2P, 3, ["op0"], OUTGOING),
                                  SamplingOp("op2", P2A, 2, ["op1"], OUTGOING)])
    roots = rng.integers(0, n["paper"], 80)
    part = str(tmp_path / "part-00000.tfrecord")
    assert s.write_tfrecords(part, roots, "paper", dag) == 80
    recs = list(wire.read_tfrecords(part))  # (verifies both CRCs of every frame)
    batch = HeteroRootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(
        recs, {0: "author", 1: "paper"}, {0: ("author", "writes", "paper"), 1: ("paper", "written_by", "author")})
    assert [g for _, g in batch.root_nodes] == roots.tolist()
    data, root_index, uniq = s.batch_graph(roots, "paper", dag)
    torch.manual_seed(5)
    ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
    model = HGT({"author": 6, "paper": 10}, {e: 0 for e in ets}, hid_dim=32, out_dim=16, num_layers=2, num_heads=2)
    model.engine = s.engine
    model = model.to(s.engine.device)
    with torch.cuda.stream(s.engine._stream):
        want = model(data, ["paper"])["paper"][root_index]
        got = model(batch.graph.to(s.engine.device), ["paper"])["paper"][
            batch.condensed_node_type_to_root_node_indices_map[1].to(s.engine.device)]
    s.engine.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    s.close()


def _grad_dict(model):
    return {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}


def test_hgt_training_gradients_match_torch_autograd():
    """HGT under autograd: gigl_hgt_aggregate_backward + the projections' backward GEMMs give the parameter gradients
    torch autograd gives through the CPU restatement (oracle/gnn_ref.hgt_conv) for the same loss"""
    import copy
    from gigl_amd.models_hetero import HGT
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    heads, hid, out_dim = 2, 32, 16
    model = HGT({t: d for t, (_, d) in NT.items()}, {e: 0 for e in ET}, hid_dim=hid, out_dim=out_dim, num_layers=2,
                num_heads=heads)
    with torch.no_grad():
        for conv in model.convs:
            for p in conv.skip.values():
                p.uniform_(-1, 1)
            for p in conv.p_rel.values():
                p.uniform_(0.5, 1.5)
    ref = copy.deepcopy(model)
    data = make_data()
    target = {t: torch.randn(n, out_dim) for t, (n, _) in NT.items() if t != "tag"}
    model = model.to(dev)
    out = model(data.to(dev), ["user", "item"])
    loss = sum(((out[t] - target[t].to(dev)) ** 2).mean() for t in target)
    loss.backward()
    got = _grad_dict(model)
    # the same loss through the CPU restatement
    h = {t: torch.relu(F.linear(x, ref.lin_dict[t].weight, ref.lin_dict[t].bias)) for t, x in data.x_dict.items()}
    for conv in ref.convs:
        p = dict(kqv={t: (conv.kqv_lin.lins[t].weight, conv.kqv_lin.lins[t].bias) for t in NT},
                 out={t: (conv.out_lin.lins[t].weight, conv.out_lin.lins[t].bias) for t in NT},
                 k_rel=conv.k_rel.weight, v_rel=conv.v_rel.weight, skip={t: conv.skip[t] for t in NT},
                 p_rel={e: conv.p_rel["__".join(e)] for e in ET}, edge_types=ET)
        h = gnn_ref.hgt_conv(h, data.edge_index_dict, p, heads)
    ref_loss = sum(((F.linear(h[t], ref.lin.weight, ref.lin.bias) - target[t]) ** 2).mean() for t in target)
    ref_loss.backward()
    want = _grad_dict(ref)
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))
    assert set(got) == set(want) and len(got) > 20
    for name in want:
        np.testing.assert_allclose(got[name].numpy(), want[name].numpy(), rtol=2e-4, atol=2e-5, err_msg=name)


@pytest.mark.parametrize("with_edge_attr", [False, True])
def test_simplehgn_conv_training_gradients_match_torch_autograd(with_edge_attr):
    """SimpleHGNConv under autograd (source-grouped softmax, weighted aggregate, projections) vs torch autograd through
    oracle/gnn_ref.simplehgn_conv"""
    import copy
    from gigl_amd.models_hetero import SimpleHGNConv
    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    n, ne, H, D, T, Ein = 300, 2500, 2, 16, 3, 8
    conv = SimpleHGNConv(24, D, T, edge_in_channels=Ein if with_edge_attr else None, num_heads=H, edge_type_dim=8)
    ref = copy.deepcopy(conv)
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, n, (2, ne), generator=g)
    et = torch.randint(0, T, (ne,), generator=g)
    x = torch.randn(n, 24, generator=g)
    ef = torch.randn(ne, Ein, generator=g) if with_edge_attr else None
    tgt = torch.randn(n, H * D, generator=g)
    conv = conv.to(dev)
    xd = x.to(dev).requires_grad_(True)
    out = conv(ei.to(dev), xd, et.to(dev), ef.to(dev) if ef is not None else None)
    loss = ((out - tgt.to(dev)) ** 2).mean()
    loss.backward()
    got = _grad_dict(conv)
    p = dict(W_nfeat=ref.W_nfeat, a_l=ref.a_l, a_r=ref.a_r, a_etype=ref.a_etype, edge_type_emb=ref.edge_type_emb,
             W_etype=(ref.W_etype.weight, ref.W_etype.bias), residual=(ref.residual.weight, ref.residual.bias))
    if with_edge_attr:
        p.update(W_efeat=ref.W_efeat, a_efeat=ref.a_efeat)
    xr = x.clone().requires_grad_(True)
    ro = gnn_ref.simplehgn_conv(ei, xr, et, p, H, D, 0.2, edge_feat=ef)
    rl = ((ro - tgt) ** 2).mean()
    rl.backward()
    want = _grad_dict(ref)
    assert abs(float(loss) - float(rl)) < 1e-5 * max(1.0, abs(float(rl)))
    assert set(got) == set(want)
    for name in want:
        np.testing.assert_allclose(got[name].numpy(), want[name].numpy(), rtol=2e-4, atol=2e-5, err_msg=name)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-4, atol=2e-5)


def test_hgt_last_layer_on_a_row_subset_equals_the_full_forward():
    """HGT(..., row_subset={type: ids}): the last layer computes the listed rows only (an inference pass needs the roots'
    rows): bit-identical to the same rows of the full forward"""
    from gigl_amd.graphdb_sampler import (INCOMING, OUTGOING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG)
    from gigl_amd.models_hetero import HGT
    A2P, P2A = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
    rng = np.random.default_rng(3)
    n = {"author": 900, "paper": 1500}
    a = (rng.zipf(1.7, 7000) % n["author"]).astype(np.uint32)
    p = rng.integers(0, n["paper"], 7000).astype(np.uint32)
    feats = {"author": rng.standard_normal((900, 6)).astype(np.float32), "paper": rng.standard_normal((1500, 10)).astype(np.float32)}
    s = HipGraphDBSampler({"author": 0, "paper": 1}, n, {A2P: (a, p), P2A: (p, a)}, {A2P: 0, P2A: 1}, feats)
    dag = SamplingOpDAG.from_ops([SamplingOp("op0", A2P, 4, [], INCOMING), SamplingOp("op1", A2P, 3, ["op0"], OUTGOING),
                                  SamplingOp("op2", P2A, 2, ["op1"], OUTGOING)])
    roots = rng.choice(n["paper"], size=96, replace=False)
    data, root_index, _ = s.batch_graph_plan(roots, "paper", dag)
    torch.manual_seed(2)
    ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
    for layers in (1, 2, 3):
        model = HGT({"author": 6, "paper": 10}, {e: 0 for e in ets}, hid_dim=32, out_dim=16, num_layers=layers,
                    num_heads=2).to(s.engine.device).eval()
        model.engine = s.engine
        with torch.no_grad():
            full = model(data, ["paper"])["paper"][root_index]
            sub = model(data, ["paper"], row_subset={"paper": root_index})["paper"]
        s.engine.synchronize()
        assert sub.shape == full.shape and torch.equal(sub, full), layers
        # the plan's own merged CSR by destination (gigl_typed_plan_merged_csr) and its rows of the roots: the arrays
        # the layers would have built with torch ops, and the same embeddings bit for bit
        # (the model's edge-type numbering is the REVERSE of the plan's slot order here)
        model_r = HGT({"author": 6, "paper": 10}, {e: 0 for e in ets[::-1]}, hid_dim=32, out_dim=16, num_layers=layers,
                      num_heads=2).to(s.engine.device).eval()
        model_r.engine = s.engine
        for mdl in (model, model_r):
            ids = mdl.convs[0].edge_types_map
            data_p, ri_p, _ = s.batch_graph_plan(roots, "paper", dag, edge_type_ids=ids)
            mc = data_p.merged_csr
            assert torch.equal(ri_p, root_index)
            srcs, dsts, etys, n_src, dst_off, n_dst = [], [], [], 0, {}, 0
            for t, x in data_p.x_dict.items():
                dst_off[t] = n_dst
                n_dst += x.shape[0]
            for et, ei in data_p.edge_index_dict.items():
                srcs.append(ei[0] + n_src)
                dsts.append(ei[1] + dst_off[et[2]])
                etys.append(torch.full((ei.shape[1],), ids[tuple(et)], dtype=torch.int32, device=ei.device))
                n_src += data_p.x_dict[et[0]].shape[0]
            src, dst, ety = torch.cat(srcs), torch.cat(dsts), torch.cat(etys)
            order = torch.sort(dst, stable=True).indices
            rowptr = torch.zeros(n_dst + 1, dtype=torch.int64, device=dst.device)
            rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n_dst), 0)
            assert torch.equal(mc["csr"][0].long(), rowptr)
            assert torch.equal(mc["csr"][1].long(), src[order]) and torch.equal(mc["csr"][2], ety[order])
            rows = root_index + dst_off["paper"]
            lens = rowptr[rows + 1] - rowptr[rows]
            assert torch.equal(mc["root_csr"][0][1:].long(), torch.cumsum(lens, 0)) and int(mc["root_csr"][0][0]) == 0
            want_col = torch.cat([src[order][rowptr[r]:rowptr[r + 1]] for r in rows.tolist()])
            assert torch.equal(mc["root_csr"][1][: want_col.numel()].long(), want_col)
            with torch.no_grad():
                plain = mdl(data, ["paper"], row_subset={"paper": root_index})["paper"]
                planned = mdl(data_p, ["paper"], row_subset={"paper": ri_p})["paper"]
                planned_full = mdl(data_p, ["paper"])["paper"][ri_p]
            s.engine.synchronize()
            assert torch.equal(planned, plain) and torch.equal(planned_full, plain)
    s.close()


@pytest.mark.parametrize("layers,l2", [(2, False), (1, True), (3, False)])
def test_one_call_typed_inference_step_equals_the_staged_forward(layers, l2):
    """gigl_hgt_infer_* (csrc/hgt_plan.hip: DAG sampler -> typed batch graph at capacity prefixes -> HGT over composed
    weights -> the roots' rows, one library call, replayed as a hipGraph) against batch_graph_plan + HGT.forward(row_subset)
    on the same roots: eager first run, captured second run, replays, a smaller batch, weights changed in place"""
    from gigl_amd.graphdb_sampler import (INCOMING, OUTGOING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG)
    from gigl_amd.models_hetero import HGT, HgtInferPlan
    A2P, P2A = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
    node_types, cet = {"author": 0, "paper": 1}, {A2P: 0, P2A: 1}
    rng = np.random.default_rng(1)
    n = {"author": 3000, "paper": 5000}
    a = (rng.zipf(1.6, 40000) % n["author"]).astype(np.uint32)
    p = rng.integers(0, n["paper"], 40000).astype(np.uint32)
    edges = {A2P: (a, p), P2A: (p, a)}
    feats = {"author": rng.standard_normal((3000, 8)).astype(np.float32), "paper": rng.standard_normal((5000, 12)).astype(np.float32)}
    s = HipGraphDBSampler(node_types, n, edges, cet, feats)
    ops = [SamplingOp("h1", A2P, 5, [], INCOMING), SamplingOp("h2", P2A, 3, ["h1"], INCOMING),
           SamplingOp("h3", A2P, 2, ["h2"], OUTGOING)]
    dag = SamplingOpDAG.from_ops(ops)
    torch.manual_seed(7)
    ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
    model = HGT({"author": 8, "paper": 12}, {e: 0 for e in ets}, hid_dim=32, out_dim=24, num_layers=layers, num_heads=2,
                should_l2_normalize_embedding_layer_output=l2).to(s.engine.device).eval()
    with torch.no_grad():  # (skip gates and relation priors away from their initial 1: both enter the composed weights)
        for conv in model.convs:
            for t in conv.skip:
                conv.skip[t].fill_(0.3)
            for k in conv.p_rel:
                conv.p_rel[k].copy_(torch.rand_like(conv.p_rel[k]) + 0.5)
    model.engine = s.engine
    B = 256
    plan = HgtInferPlan(model, s, "paper", dag, B)

    def staged(roots):
        graph, ri, _ = s.batch_graph_plan(roots, "paper", dag, b_max=B, edge_type_ids=model.convs[0].edge_types_map)
        with torch.no_grad(), torch.cuda.stream(s.engine._stream):
            out = model(graph, ["paper"], row_subset={"paper": ri})["paper"]
        s.engine.synchronize()
        return out.cpu().numpy()

    def one_call(roots, nxt=None):
        r = roots if torch.is_tensor(roots) else torch.from_numpy(roots.astype(np.uint32).view(np.int32)).to(s.engine.device)
        torch.cuda.synchronize()
        out = plan.run(r, nxt)
        s.engine.synchronize()
        torch.cuda.synchronize()
        return out.cpu().numpy()

    pools = [rng.integers(0, n["paper"], B) for _ in range(4)]
    for roots in pools:  # run 1 eager, run 2 captured, then replays
        np.testing.assert_allclose(one_call(roots), staged(roots), rtol=2e-5, atol=2e-5)
    short = rng.integers(0, n["paper"], 100)  # a smaller batch: another capture
    np.testing.assert_allclose(one_call(short), staged(short), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(one_call(short), staged(short), rtol=2e-5, atol=2e-5)
    dup = np.concatenate([pools[0][:50], pools[0][:50]])  # repeated roots
    np.testing.assert_allclose(one_call(dup), staged(dup), rtol=2e-5, atol=2e-5)
    with torch.no_grad():  # a parameter update: the composed weights are rebuilt, the step re-captured
        model.lin.weight.mul_(1.5)
        model.convs[0].kqv_lin.lins["author"].weight.add_(0.05)
    for roots in pools[:3]:
        np.testing.assert_allclose(one_call(roots), staged(roots), rtol=2e-5, atol=2e-5)
    # announced batches: batch i + 1's graph part is built on the plan's own stream under batch i's layers
    devs = [torch.from_numpy(r.astype(np.uint32).view(np.int32)).to(s.engine.device) for r in pools]
    want = [staged(r) for r in pools]
    for rep in range(2):
        for i in range(4):
            got = one_call(devs[i], devs[i + 1] if i + 1 < 4 else None)
            np.testing.assert_allclose(got, want[i], rtol=2e-5, atol=2e-5)
    # an announcement that is not honoured (other roots arrive): the workspace is rebuilt
    got = one_call(devs[2], devs[3])
    np.testing.assert_allclose(one_call(devs[0]), want[0], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got, want[2], rtol=2e-5, atol=2e-5)
    plan.use_graph(False)
    np.testing.assert_allclose(one_call(pools[3]), staged(pools[3]), rtol=2e-5, atol=2e-5)
    plan.close()
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(64, 64), (128, 128), (64, 200)])
def test_grouped_projection_launch_equals_the_single_launches(k, n):
    """gigl_linear_grouped (the typed layers' per-type / per-slot projections behind one grid) against gigl_linear on every
    product: bit-identical rows (same kernel, same summation order), rows past a group's device count untouched, groups of
    very different row counts, one without a bias, one empty"""
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    st = torch.cuda.Stream()
    eng.bind_stream(st)
    dev = eng.device
    g = torch.Generator(device="cpu").manual_seed(k + n)
    caps, ms = [700, 130, 5000, 64, 256], [651, 130, 4097, 0, 1]
    groups, want = [], []
    with torch.cuda.stream(st):
        for i, (cap, m) in enumerate(zip(caps, ms)):
            a = torch.randn((cap, k), generator=g).to(dev)
            w = (torch.randn((n, k), generator=g) / 8).to(dev)
            bias = None if i == 1 else torch.randn(n, generator=g).to(dev)
            m_dev = torch.tensor([m], dtype=torch.int32, device=dev)
            y = torch.full((cap, n), 7.0, device=dev)
            groups.append((a, w, bias, m_dev, y))
            want.append(eng.linear(a, w, bias, m_dev, cap, act=1, out=torch.full((cap, n), 7.0, device=dev)))
        eng.linear_grouped(groups, k, n, act=1)
    st.synchronize()
    for (a, w, bias, m_dev, y), ref, m in zip(groups, want, ms):
        assert torch.equal(y[:m], ref[:m])
        assert bool((y[m:] == 7.0).all())
    eng.close()
