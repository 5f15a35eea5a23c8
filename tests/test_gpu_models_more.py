"""GIN / Transformer encoders (gigl_amd/models_more.py) against the fp32 CPU restatements of PyG 2.5.3's GINConv /
TransformerConv (oracle/gnn_ref.py): root embeddings over a sampled batch (forward, 1e-5) and every parameter's
gradient over a coalesced batch graph (torch autograd through the restatement, 1e-4)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges
from oracle import gnn_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from gigl_amd.engine import HipEngine
    s, d = rmat_edges(12, 60000, seed=9)
    n = 1 << 12
    rowptr, col = oracle.build_csc(n, s, d, is_directed=True)
    x = (np.random.default_rng(0).standard_normal((n, 40)) / 4).astype(np.float32)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    yield eng, rowptr, col, x, n
    eng.close()


def _union(eng, rowptr, col, roots, fan):
    from gigl_amd.models import HipBatch
    tree = eng.sample_khop(roots, fan)
    u = eng.union_build(tree)
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
    o = oracle.union_build(roots, fan, nbr_o)
    return HipBatch(eng, tree, u), u, o


def _bn(sd, prefix):
    return (sd[prefix + "weight"], sd[prefix + "bias"], sd[prefix + "running_mean"], sd[prefix + "running_var"], 1e-5)


def _gin_ref(x, ei, sd, L, eps, batchnorm, act_first):
    h = x
    for l in range(L):
        p = f"conv_layers.{l}.nn."
        bn = _bn(sd, p + "norms.0.module.") if batchnorm else None
        h = gnn_ref.gin_conv(h, ei, sd[p + "lins.0.weight"], sd[p + "lins.0.bias"], sd[p + "lins.1.weight"],
                             sd[p + "lins.1.bias"], eps=float(sd[f"conv_layers.{l}.eps"]), act_first=act_first, bn=bn)
        if l < L - 1:
            if act_first:
                h = torch.relu(h)
            if batchnorm:
                g, b, mu, var, e = _bn(sd, f"batchnorm_layers.{l}.")
                h = (h - mu) / torch.sqrt(var + e) * g + b
            if not act_first:
                h = torch.relu(h)
    return h


@pytest.mark.parametrize("hid,out,fan,eps,batchnorm,act_first", [(32, 16, [8, 5], 0.0, False, False),
                                                                 (64, 24, [6, 4, 3], 0.25, False, False),
                                                                 (48, 20, [8, 5], 0.1, True, False),
                                                                 (48, 20, [8, 5], 0.0, True, True)])
def test_gin_roots_match_the_whole_graph_forward(setup, hid, out, fan, eps, batchnorm, act_first):
    from gigl_amd.models_more import GIN
    eng, rowptr, col, x, n = setup
    torch.manual_seed(hid)
    L = len(fan)
    model = GIN(40, hid, out, num_layers=L, eps=eps, batchnorm=batchnorm, activation_before_norm=act_first).to(eng.device)
    if batchnorm:
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(0, 0.1)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.normal_(1, 0.1)
                    m.bias.normal_(0, 0.1)
    model.eval()
    roots = np.random.default_rng(4).integers(0, n, size=130).astype(np.uint32)
    batch, u, o = _union(eng, rowptr, col, roots, fan)
    got = model(batch)[u.root_local[:130].long()].cpu().numpy()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
    ref = _gin_ref(torch.from_numpy(x[o["nodes"]]), ei, sd, L, eps, batchnorm, act_first)
    np.testing.assert_allclose(got, ref[o["root_local"]].numpy(), rtol=2e-5, atol=2e-5)


def _graph(rng, n, e, d):
    ei = torch.from_numpy(np.unique(rng.integers(0, n, (2, e)), axis=1))
    ei = ei[:, np.lexsort((ei[1].numpy(), ei[0].numpy()))]
    x = torch.from_numpy((rng.standard_normal((n, d)) / 2).astype(np.float32))
    return ei, x


@pytest.mark.parametrize("train_eps", [False, True])
def test_gin_training_gradients_match_torch_autograd(train_eps):
    from gigl_amd.engine import HipEngine
    from gigl_amd.models_more import GIN
    from gigl_amd.nn import GraphData
    rng = np.random.default_rng(3)
    n, d, hid, out = 300, 12, 32, 16
    ei, x = _graph(rng, n, 2200, d)
    eng = HipEngine(0)
    try:
        torch.manual_seed(2)
        model = GIN(d, hid, out, num_layers=2, eps=0.3, train_eps=train_eps).to(eng.device).train()
        model.engine = eng
        g = GraphData(x=x.clone(), edge_index=ei).to(eng.device)
        g.x.requires_grad_(True)
        wsum = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32))
        y = model(g)
        (y * wsum.to(eng.device)).sum().backward()
        ref = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        h = xr
        for l in range(2):
            p = f"conv_layers.{l}.nn."
            src, dst = ei[0], ei[1]
            agg = torch.zeros_like(h).index_add(0, dst, h[src])
            z = (agg + (1.0 + ref[f"conv_layers.{l}.eps"]) * h) @ ref[p + "lins.0.weight"].T + ref[p + "lins.0.bias"]
            h = torch.relu(z) @ ref[p + "lins.1.weight"].T + ref[p + "lins.1.bias"]
            if l == 0:
                h = torch.relu(h)
        np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=1e-5)
        (h * wsum).sum().backward()
        names = [k for k, _ in model.named_parameters()]
        assert ("conv_layers.0.eps" in names) == train_eps
        for name, prm in model.named_parameters():
            want = ref[name].grad
            assert prm.grad is not None and want is not None, name
            scale = float(want.abs().max()) + 1e-6
            np.testing.assert_allclose(prm.grad.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)
        np.testing.assert_allclose(g.x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4,
                                   atol=1e-4 * float(xr.grad.abs().max()))
    finally:
        eng.close()


def _transformer_ref(x, ei, sd, L, heads, hid, out, beta):
    h = x
    for l in range(L):
        pre = f"conv_layers.{l}."
        p = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        h = gnn_ref.transformer_conv(h, ei, p, heads if l < L - 1 else 1, hid if l < L - 1 else out, beta=beta)
        if l < L - 1:
            h = torch.relu(h)
    return h


@pytest.mark.parametrize("heads,hid,out,beta", [(2, 16, 32, False), (4, 32, 64, True), (1, 64, 32, False)])
def test_transformer_roots_match_the_whole_graph_forward(setup, heads, hid, out, beta):
    from gigl_amd.models_more import Transformer
    eng, rowptr, col, x, n = setup
    torch.manual_seed(heads)
    model = Transformer(40, hid, out, num_layers=2, heads=heads, beta=beta).to(eng.device).eval()
    roots = np.random.default_rng(6).integers(0, n, size=110).astype(np.uint32)
    batch, u, o = _union(eng, rowptr, col, roots, [7, 4])
    got = model(batch)[u.root_local[:110].long()].cpu().numpy()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
    ref = _transformer_ref(torch.from_numpy(x[o["nodes"]]), ei, sd, 2, heads, hid, out, beta)
    np.testing.assert_allclose(got, ref[o["root_local"]].numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("heads,hid,out,beta", [(2, 16, 32, False), (4, 32, 16, True)])
def test_transformer_training_gradients_match_torch_autograd(heads, hid, out, beta):
    from gigl_amd.engine import HipEngine
    from gigl_amd.models_more import Transformer
    from gigl_amd.nn import GraphData
    rng = np.random.default_rng(heads)
    n, d = 280, 12
    ei, x = _graph(rng, n, 2000, d)
    eng = HipEngine(0)
    try:
        torch.manual_seed(5)
        model = Transformer(d, hid, out, num_layers=2, heads=heads, beta=beta).to(eng.device).train()
        model.engine = eng
        g = GraphData(x=x.clone(), edge_index=ei).to(eng.device)
        g.x.requires_grad_(True)
        wsum = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32))
        y = model(g)
        (y * wsum.to(eng.device)).sum().backward()
        ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        h = _transformer_ref(xr, ei, ref, 2, heads, hid, out, beta)
        np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=1e-5)
        (h * wsum).sum().backward()
        # (the key bias shifts every logit of a row alike: its gradient is zero up to rounding — absolute floor)
        floor = 1e-4 * max(float(ref[k].grad.abs().max()) for k, _ in model.named_parameters())
        for name, prm in model.named_parameters():
            want = ref[name].grad
            assert prm.grad is not None and want is not None, name
            scale = float(want.abs().max()) + 1e-6
            np.testing.assert_allclose(prm.grad.cpu().numpy(), want.numpy(), rtol=1e-4,
                                       atol=max(1e-4 * scale, floor if "lin_key.bias" in name else 0.0), err_msg=name)
        np.testing.assert_allclose(g.x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4,
                                   atol=1e-4 * float(xr.grad.abs().max()))
    finally:
        eng.close()


@pytest.mark.parametrize("proj,diag", [(None, 0.0), (10, 0.5)])
def test_dcn_v2_feature_interaction_before_the_convs(setup, proj, diag):
    """feature_interaction_layer=DCNv2 (feature_interaction.py:104-155) runs on the node features before the first conv,
    both over a sampled batch (resident feature table) and under autograd over a batch graph"""
    from gigl_amd.models import GraphSAGE
    from gigl_amd.models_more import DCNv2
    from gigl_amd.nn import GraphData
    eng, rowptr, col, x, n = setup
    torch.manual_seed(7)
    model = GraphSAGE(40, 32, 16, num_layers=2,
                      feature_interaction_layer=DCNv2(40, num_layers=2, projection_dim=proj, diag_scale=diag)).to(eng.device)
    model.engine = eng
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}

    def dcn(t, sd):
        x0, xl = t, t
        for i in range(2):
            p = f"feats_interaction._layers.{i}."
            if proj is None:
                prod = xl @ sd[p + "_lin.weight"].T + sd[p + "_lin.bias"]
            else:
                prod = (xl @ sd[p + "_lin_u.weight"].T + sd[p + "_lin_u.bias"]) @ sd[p + "_lin_v.weight"].T + sd[p + "_lin_v.bias"]
            xl = x0 * (prod + diag * xl) + xl
        return xl

    roots = np.random.default_rng(8).integers(0, n, size=90).astype(np.uint32)
    batch, u, o = _union(eng, rowptr, col, roots, [6, 4])
    model.eval()
    got = model(batch)[u.root_local[:90].long()].cpu().numpy()
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
    ref = gnn_ref.graphsage_forward(dcn(torch.from_numpy(x[o["nodes"]]), sd), ei, sd, 2)
    np.testing.assert_allclose(got, ref[o["root_local"]].numpy(), rtol=2e-5, atol=2e-5)
    # training over a coalesced graph: gradients reach the interaction layer's weights
    rng = np.random.default_rng(9)
    ei2, x2 = _graph(rng, 200, 1500, 40)
    g = GraphData(x=x2.clone(), edge_index=ei2).to(eng.device)
    model.train()
    y = model(g)
    wsum = torch.from_numpy(rng.standard_normal((200, 16)).astype(np.float32))
    (y * wsum.to(eng.device)).sum().backward()
    refp = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    h = gnn_ref.graphsage_forward(dcn(x2, refp), ei2, refp, 2)
    np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=2e-5, atol=2e-5)
    (h * wsum).sum().backward()
    for name, prm in model.named_parameters():
        want = refp[name].grad
        scale = float(want.abs().max()) + 1e-6
        np.testing.assert_allclose(prm.grad.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)


def _gatv2_ref(x, ei, sd, L, heads, hid, out, share):
    h = x
    for l in range(L):
        pre = f"conv_layers.{l}."
        p = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        h = gnn_ref.gatv2_conv(h, ei, p, heads if l < L - 1 else 1, hid if l < L - 1 else out, share_weights=share)
        if l < L - 1:
            h = torch.relu(h)
    return h


@pytest.mark.parametrize("heads,hid,out,share,fan", [(2, 16, 32, False, [7, 4]), (4, 64, 64, True, [6, 4]),
                                                     (1, 256, 128, False, [5, 3]), (4, 8, 16, False, [5, 3, 2])])
def test_gatv2_roots_match_the_whole_graph_forward(setup, heads, hid, out, share, fan):
    from gigl_amd.models_more import GATv2
    eng, rowptr, col, x, n = setup
    torch.manual_seed(heads)
    L = len(fan)
    model = GATv2(40, hid, out, num_layers=L, heads=heads, share_weights=share).to(eng.device).eval()
    with torch.no_grad():
        for c in model.conv_layers:
            c.bias.normal_(0, 0.1)
    roots = np.random.default_rng(6).integers(0, n, size=110).astype(np.uint32)
    batch, u, o = _union(eng, rowptr, col, roots, fan)
    got = model(batch)[u.root_local[:110].long()].cpu().numpy()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
    ref = _gatv2_ref(torch.from_numpy(x[o["nodes"]]), ei, sd, L, heads, hid, out, share)
    np.testing.assert_allclose(got, ref[o["root_local"]].numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("heads,hid,out,share", [(2, 16, 32, False), (4, 32, 16, True), (1, 128, 64, False)])
def test_gatv2_training_gradients_match_torch_autograd(heads, hid, out, share):
    from gigl_amd.engine import HipEngine
    from gigl_amd.models_more import GATv2
    from gigl_amd.nn import GraphData
    rng = np.random.default_rng(heads + 20)
    n, d = 280, 12
    ei, x = _graph(rng, n, 2000, d)
    assert bool((ei[0] == ei[1]).any())  # a few self loops: removed, then one per node added back
    eng = HipEngine(0)
    try:
        torch.manual_seed(5)
        model = GATv2(d, hid, out, num_layers=2, heads=heads, share_weights=share).to(eng.device).train()
        model.engine = eng
        with torch.no_grad():
            for c in model.conv_layers:
                c.bias.normal_(0, 0.1)
        g = GraphData(x=x.clone(), edge_index=ei).to(eng.device)
        g.x.requires_grad_(True)
        wsum = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32))
        y = model(g)
        (y * wsum.to(eng.device)).sum().backward()
        ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        if share:
            for l in range(2):  # one tensor behind both names, as in the module
                ref[f"conv_layers.{l}.lin_r.weight"] = ref[f"conv_layers.{l}.lin_l.weight"]
                ref[f"conv_layers.{l}.lin_r.bias"] = ref[f"conv_layers.{l}.lin_l.bias"]
        xr = x.clone().requires_grad_(True)
        h = _gatv2_ref(xr, ei, ref, 2, heads, hid, out, share)
        np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=1e-5)
        (h * wsum).sum().backward()
        for name, prm in model.named_parameters():
            want = ref[name].grad
            assert prm.grad is not None and want is not None, name
            scale = float(want.abs().max()) + 1e-6
            np.testing.assert_allclose(prm.grad.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)
        np.testing.assert_allclose(g.x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4,
                                   atol=1e-4 * float(xr.grad.abs().max()))
    finally:
        eng.close()


def test_gine_gradients_and_sampled_batch():
    """GINE (messages relu(x_j + lin(e_ji)), homogeneous.py:252-297): training gradients over a batch graph with edge
    features == torch autograd through oracle/gnn_ref.gine_conv; root embeddings of a sampled batch (edge rows from the
    resident edge-feature table) == the whole-union-graph forward"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import HipBatch
    from gigl_amd.models_more import GINE
    from gigl_amd.nn import GraphData
    rng = np.random.default_rng(12)
    n, d, de, hid, out = 260, 12, 5, 32, 16
    ei, x = _graph(rng, n, 1800, d)
    ea = torch.from_numpy(rng.standard_normal((ei.shape[1], de)).astype(np.float32))
    eng = HipEngine(0)
    try:
        torch.manual_seed(4)
        model = GINE(d, hid, out, num_layers=2, edge_dim=de, eps=0.2, train_eps=True).to(eng.device).train()
        model.engine = eng
        g = GraphData(x=x.clone(), edge_index=ei, edge_attr=ea).to(eng.device)
        g.x.requires_grad_(True)
        wsum = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32))
        y = model(g)
        (y * wsum.to(eng.device)).sum().backward()
        ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        h = xr
        for l in range(2):
            p = f"conv_layers.{l}."
            h = gnn_ref.gine_conv(h, ei, ea, ref[p + "lin.weight"], ref[p + "lin.bias"], ref[p + "nn.lins.0.weight"],
                                  ref[p + "nn.lins.0.bias"], ref[p + "nn.lins.1.weight"], ref[p + "nn.lins.1.bias"],
                                  eps=ref[p + "eps"])
            if l == 0:
                h = torch.relu(h)
        np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=1e-5)
        (h * wsum).sum().backward()
        for name, prm in model.named_parameters():
            want = ref[name].grad
            assert prm.grad is not None and want is not None, name
            scale = float(want.abs().max()) + 1e-6
            np.testing.assert_allclose(prm.grad.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)
        np.testing.assert_allclose(g.x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4,
                                   atol=1e-4 * float(xr.grad.abs().max()))
        # sampled batch: the same graph resident on the device with its edge-feature table
        src, dst = ei[0].numpy().astype(np.uint32), ei[1].numpy().astype(np.uint32)
        rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
        eng.load_csc(rowptr, col)
        eng.load_features(x.numpy())
        eng.load_edge_features(src, dst, ea.numpy(), True)
        roots = rng.integers(0, n, size=60).astype(np.uint32)
        tree = eng.sample_khop(roots, [6, 4])
        u = eng.union_build(tree)
        model.eval()
        got = model(HipBatch(eng, tree, u))[u.root_local[:60].long()].cpu().numpy()
        nbr_o, _ = oracle.sample_khop(rowptr, col, roots, [6, 4], canonical=True)
        o = oracle.union_build(roots, [6, 4], nbr_o)
        uei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
        rows = {(int(s_), int(d_)): i for i, (s_, d_) in enumerate(zip(src.tolist(), dst.tolist()))}
        gl = o["nodes"]
        uea = ea[[rows[(int(gl[a]), int(gl[b]))] for a, b in zip(uei[0].tolist(), uei[1].tolist())]]
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        h = torch.from_numpy(x.numpy()[gl])
        for l in range(2):
            p = f"conv_layers.{l}."
            h = gnn_ref.gine_conv(h, uei, uea, sd[p + "lin.weight"], sd[p + "lin.bias"], sd[p + "nn.lins.0.weight"],
                                  sd[p + "nn.lins.0.bias"], sd[p + "nn.lins.1.weight"], sd[p + "nn.lins.1.bias"],
                                  eps=float(sd[p + "eps"]))
            if l == 0:
                h = torch.relu(h)
        np.testing.assert_allclose(got, h[o["root_local"]].numpy(), rtol=2e-5, atol=2e-5)
    finally:
        eng.close()


@pytest.mark.parametrize("heads,hid,out,share", [(2, 16, 32, False), (4, 32, 16, True)])
def test_gatv2_with_edge_features_gradients(heads, hid, out, share):
    """GATv2Conv(edge_dim): lin_edge(e) inside the logit's leaky_relu, self loops filled with the row's mean attribute;
    forward and every gradient (lin_edge included) against torch autograd through the restatement"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models_more import GATv2
    from gigl_amd.nn import GraphData
    rng = np.random.default_rng(heads + 40)
    n, d, de = 240, 12, 6
    ei, x = _graph(rng, n, 1700, d)
    ea = torch.from_numpy(rng.standard_normal((ei.shape[1], de)).astype(np.float32))
    eng = HipEngine(0)
    try:
        torch.manual_seed(6)
        model = GATv2(d, hid, out, num_layers=2, heads=heads, share_weights=share, edge_dim=de).to(eng.device).train()
        model.engine = eng
        g = GraphData(x=x.clone(), edge_index=ei, edge_attr=ea).to(eng.device)
        g.x.requires_grad_(True)
        wsum = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32))
        y = model(g)
        (y * wsum.to(eng.device)).sum().backward()
        ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        if share:
            for l in range(2):
                ref[f"conv_layers.{l}.lin_r.weight"] = ref[f"conv_layers.{l}.lin_l.weight"]
                ref[f"conv_layers.{l}.lin_r.bias"] = ref[f"conv_layers.{l}.lin_l.bias"]
        xr = x.clone().requires_grad_(True)
        h = xr
        for l in range(2):
            pre = f"conv_layers.{l}."
            p = {k[len(pre):]: v for k, v in ref.items() if k.startswith(pre)}
            h = gnn_ref.gatv2_conv(h, ei, p, heads if l == 0 else 1, hid if l == 0 else out, share_weights=share,
                                   edge_attr=ea)
            if l == 0:
                h = torch.relu(h)
        np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=1e-5)
        (h * wsum).sum().backward()
        assert "conv_layers.0.lin_edge.weight" in dict(model.named_parameters())
        for name, prm in model.named_parameters():
            want = ref[name].grad
            assert prm.grad is not None and want is not None, name
            scale = float(want.abs().max()) + 1e-6
            np.testing.assert_allclose(prm.grad.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4 * scale, err_msg=name)
        np.testing.assert_allclose(g.x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4,
                                   atol=1e-4 * float(xr.grad.abs().max()))
    finally:
        eng.close()


@pytest.mark.parametrize("heads,hid,out,beta", [(2, 16, 32, False), (4, 32, 16, True)])
def test_transformer_with_edge_features_gradients(heads, hid, out, beta):
    """TransformerConv(edge_dim): lin_edge(e) joins the keys and the values of every edge
    (gigl_transformer_aggregate_edge); forward and every gradient against torch autograd through the restatement"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models_more import Transformer
    from gigl_amd.nn import GraphData
    rng = np.random.default_rng(heads + 60)
    n, d, de = 230, 12, 5
    ei, x = _graph(rng, n, 1600, d)
    ea = torch.from_numpy(rng.standard_normal((ei.shape[1], de)).astype(np.float32))
    eng = HipEngine(0)
    try:
        torch.manual_seed(8)
        model = Transformer(d, hid, out, num_layers=2, heads=heads, beta=beta, edge_dim=de).to(eng.device).train()
        model.engine = eng
        g = GraphData(x=x.clone(), edge_index=ei, edge_attr=ea).to(eng.device)
        g.x.requires_grad_(True)
        wsum = torch.from_numpy(rng.standard_normal((n, out)).astype(np.float32))
        y = model(g)
        (y * wsum.to(eng.device)).sum().backward()
        ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        h = xr
        for l in range(2):
            pre = f"conv_layers.{l}."
            p = {k[len(pre):]: v for k, v in ref.items() if k.startswith(pre)}
            h = gnn_ref.transformer_conv(h, ei, p, heads if l == 0 else 1, hid if l == 0 else out, beta=beta, edge_attr=ea)
            if l == 0:
                h = torch.relu(h)
        np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=1e-5)
        (h * wsum).sum().backward()
        floor = 1e-4 * max(float(ref[k].grad.abs().max()) for k, _ in model.named_parameters())
        assert "conv_layers.0.lin_edge.weight" in dict(model.named_parameters())
        for name, prm in model.named_parameters():
            want = ref[name].grad
            assert prm.grad is not None and want is not None, name
            scale = float(want.abs().max()) + 1e-6
            np.testing.assert_allclose(prm.grad.cpu().numpy(), want.numpy(), rtol=1e-4,
                                       atol=max(1e-4 * scale, floor if "lin_key.bias" in name else 0.0), err_msg=name)
        np.testing.assert_allclose(g.x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4,
                                   atol=1e-4 * float(xr.grad.abs().max()))
    finally:
        eng.close()
