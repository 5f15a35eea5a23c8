"""SAGEConv aggr = mean | sum | max and the BasicHomogeneousGNN options (batchnorm, dropout, activation order,
linear head, return_emb) on the HIP kernels vs a plain fp32 torch restatement of PyG 2.5.3's documented formulas
(python/gigl/src/common/models/pyg/homogeneous.py:107-153,171-202; PyG is not vendored: "parity unpinned").
Bar: 1e-5 forward, 1e-4 relative on gradients."""
import numpy as np
import pytest
import torch

from gigl_amd.models import GraphSAGE
from gigl_amd.nn import GraphData
from helpers import rmat_edges

pytestmark = pytest.mark.gpu


def _ref_conv(x, ei, wl, bl, wr, aggr):
    n = x.shape[0]
    src, dst = ei[0], ei[1]
    if aggr == "max":
        agg = torch.full((n, x.shape[1]), float("-inf"), dtype=x.dtype)
        agg = agg.scatter_reduce(0, dst[:, None].expand(-1, x.shape[1]), x[src], reduce="amax", include_self=True)
        agg = torch.where(torch.isinf(agg), torch.zeros_like(agg), agg)
    else:
        agg = torch.zeros((n, x.shape[1]), dtype=x.dtype).index_add_(0, dst, x[src])
        if aggr == "mean":
            deg = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=x.dtype))
            agg = agg / deg.clamp(min=1)[:, None]
    out = agg @ wl.T + x @ wr.T
    return out + bl if bl is not None else out


def _ref_forward(model, x, ei):
    h = x
    xs = []
    L = model.num_layers
    for l, conv in enumerate(model.conv_layers):
        h = _ref_conv(h, ei, conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight, model.aggr)
        if l == L - 1 and not model.activation_after_last_conv and model.jk_layer is None:
            break
        if model.activation_before_norm:
            h = torch.relu(h)
        if model.batchnorm:
            h = model.batchnorm_layers[l](h)
        if not model.activation_before_norm:
            h = torch.relu(h)
        h = model.dropout(h)
        xs.append(h)
    if model.jk_layer is not None:
        h = model.jk_layer(xs)
    if model.should_l2_normalize_embedding_layer_output:
        h = torch.nn.functional.normalize(h, p=2, dim=1)
    if model.return_emb:
        return h
    return model.linear(h) if model.linear_layer else h


def _graph(n=400, e=3000, d=12, seed=0):
    rng = np.random.default_rng(seed)
    src, dst = rmat_edges(9, e, seed=seed + 1)
    src, dst = src % n, dst % n
    pairs = np.unique(np.stack([src, dst]), axis=1)
    x = rng.standard_normal((n, d)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(pairs.astype(np.int64))


@pytest.mark.parametrize("aggr", ["mean", "sum", "max"])
@pytest.mark.parametrize("d", [12, 7])  # vectorised and generic gather kernels
def test_aggr_forward_backward(aggr, d):
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    x, ei = _graph(d=d)
    torch.manual_seed(0)
    model = GraphSAGE(d, 16, 5, num_layers=2, aggr=aggr)
    ref_out = _ref_forward(model, x.clone().requires_grad_(False), ei)
    ref_loss = (ref_out ** 2).sum()
    ref_grads = torch.autograd.grad(ref_loss, list(model.parameters()))
    m = model.to(eng.device)
    m.engine = eng
    out = m(GraphData(x=x, edge_index=ei).to(eng.device))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.detach().numpy(), rtol=1e-5, atol=1e-5)
    loss = (out ** 2).sum()
    grads = torch.autograd.grad(loss, list(m.parameters()))
    for g, r in zip(grads, ref_grads):
        scale = float(r.abs().max()) + 1e-6
        assert float((g.cpu() - r).abs().max()) / scale < 1e-4
    eng.close()


def test_max_gradient_is_shared_among_ties():
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    x = torch.tensor([[1.0, 2.0], [1.0, 0.0], [0.5, 2.0], [0.0, 0.0]])
    ei = torch.tensor([[0, 1, 2], [3, 3, 3]])  # node 3 aggregates 0, 1, 2: ties in both features
    g = GraphData(x=x, edge_index=ei).to(eng.device)
    h = g.x.clone().requires_grad_(True)
    from gigl_amd.nn import sage_conv
    wl = torch.eye(2, device=eng.device)
    wr = torch.zeros(2, 2, device=eng.device)
    y = sage_conv(h, wl, None, wr, eng, g, False, "max")
    assert y[3].tolist() == [1.0, 2.0]
    y[3].sum().backward()
    assert h.grad.cpu().tolist() == [[0.5, 0.5], [0.5, 0.0], [0.0, 0.5], [0.0, 0.0]]
    eng.close()


@pytest.mark.parametrize("opts", [dict(batchnorm=True), dict(batchnorm=True, activation_before_norm=True),
                                  dict(linear_layer=True), dict(linear_layer=True, return_emb=True),
                                  dict(activation_after_last_conv=True, should_l2_normalize_embedding_layer_output=True),
                                  dict(dropout=0.5), dict(jk_mode="cat"), dict(jk_mode="max", batchnorm=True),
                                  dict(jk_mode="lstm", linear_layer=True)])
def test_model_options_eval_mode(opts):
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    x, ei = _graph(d=12, seed=3)
    torch.manual_seed(1)
    model = GraphSAGE(12, 16, 5, num_layers=2, **opts)
    if opts.get("batchnorm"):  # non-trivial running statistics
        with torch.no_grad():
            model.batchnorm_layers[0].running_mean.uniform_(-0.5, 0.5)
            model.batchnorm_layers[0].running_var.uniform_(0.5, 2.0)
    model.eval()
    ref = _ref_forward(model, x, ei).detach().numpy()
    m = model.to(eng.device)
    m.engine = eng
    out = m(GraphData(x=x, edge_index=ei).to(eng.device)).detach().cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)
    sd = model.state_dict()
    assert "conv_layers.0.lin_l.weight" in sd and ("linear.weight" in sd) == bool(opts.get("linear_layer"))
    assert ("batchnorm_layers.0.running_mean" in sd) == bool(opts.get("batchnorm"))
    assert ("jk_layer.output_linear.weight" in sd) == bool(opts.get("jk_mode"))
    eng.close()


def test_union_inference_with_options_matches_whole_graph():
    """trimmed inference over the union graph (HipBatch) == the whole-graph forward at the roots, with max
    aggregation + batchnorm + linear head"""
    import oracle
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import HipBatch
    from oracle import gnn_ref
    eng = HipEngine(0)
    rng = np.random.default_rng(0)
    n, d = 3000, 12
    src, dst = rmat_edges(12, 20000, seed=5)
    src, dst = (src % n).astype(np.uint32), (dst % n).astype(np.uint32)
    feats = rng.standard_normal((n, d)).astype(np.float32)
    eng.build_from_coo(n, src, dst, is_directed=False)
    eng.load_features(feats)
    roots = rng.integers(0, n, 64).astype(np.uint32)
    tree = eng.sample_khop(roots, [6, 4])
    u = eng.union_build(tree)
    torch.manual_seed(2)
    model = GraphSAGE(d, 16, 5, num_layers=2, aggr="max", batchnorm=True, linear_layer=True, jk_mode="cat").eval()
    with torch.no_grad():
        model.batchnorm_layers[0].running_mean.uniform_(-0.5, 0.5)
    nodes, rp, col = u.to_csr()
    ei = gnn_ref.union_edge_index(rp, col)
    ei = ei if isinstance(ei, torch.Tensor) else torch.from_numpy(ei)
    ref = _ref_forward(model, torch.from_numpy(feats[nodes]), ei.to(torch.int64)).detach().numpy()
    m = model.to(eng.device)
    out = m(HipBatch(engine=eng, tree=tree, union=u))
    rl = u.root_local[: len(roots)].cpu().numpy()
    np.testing.assert_allclose(out.cpu().numpy()[rl], ref[rl], rtol=1e-5, atol=1e-5)
    eng.close()
