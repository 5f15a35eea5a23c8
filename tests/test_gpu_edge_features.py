"""Edge features on the HIP path (S1 `_edge_features`, S6 hydrateEdges, T4 GATConv(edge_dim) / T6 EdgeAttrGATConv):
edge-id lookups and the `col`-ordered feature table against numpy, union-graph edge attributes against a dictionary
lookup, and GAT / EdgeAttrGAT root embeddings against the fp32 restatement in oracle/gnn_ref.py (1e-5)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges
from oracle import gnn_ref

pytestmark = pytest.mark.gpu
DE = 6


def _csc_positions(rowptr, col, src, dst):
    """numpy restatement: position of (src -> dst) in col, -1 when absent"""
    out = np.full(len(src), -1, dtype=np.int64)
    for i, (s, d) in enumerate(zip(src, dst)):
        row = col[rowptr[d]:rowptr[d + 1]]
        k = np.searchsorted(row, s)
        if k < len(row) and row[k] == s:
            out[i] = rowptr[d] + k
    return out


@pytest.fixture(scope="module")
def setup():
    from gigl_amd.engine import HipEngine
    s, d = rmat_edges(11, 30000, seed=5)
    n = 1 << 11
    s = np.concatenate([s, np.arange(0, 100, dtype=np.uint32)])  # some self loops
    d = np.concatenate([d, np.arange(0, 100, dtype=np.uint32)])
    rowptr, col = oracle.build_csc(n, s, d, is_directed=True)
    x = (np.random.default_rng(0).standard_normal((n, 20)) / 4).astype(np.float32)
    efeat = (np.random.default_rng(1).standard_normal((len(col), DE)) / 2).astype(np.float32)  # row p = edge at col[p]
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    # hand the features over in a shuffled COO order: the engine must put them back in `col` order
    dst_of = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr))
    perm = np.random.default_rng(2).permutation(len(col))
    eng.load_edge_features(col[perm], dst_of[perm], efeat[perm], is_directed=True)
    yield eng, rowptr, col, x, efeat, n
    eng.close()


def test_edge_ids_and_table_order(setup):
    eng, rowptr, col, x, efeat, n = setup
    assert eng.edge_feat_dim == DE
    np.testing.assert_array_equal(eng._efeat.cpu().numpy(), efeat)
    rng = np.random.default_rng(3)
    dst_of = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr))
    pick = rng.integers(0, len(col), size=4000)
    src = np.concatenate([col[pick], rng.integers(0, n, size=4000).astype(np.uint32), [0xFFFFFFFF]]).astype(np.uint32)
    dst = np.concatenate([dst_of[pick], rng.integers(0, n, size=4000).astype(np.uint32), [3]]).astype(np.uint32)
    got = eng.edge_ids(torch.from_numpy(src.view(np.int32)), torch.from_numpy(dst.view(np.int32))).cpu().numpy()
    want = _csc_positions(rowptr, col, src[:-1], dst[:-1])
    np.testing.assert_array_equal(got[:-1], want)
    assert got[-1] == -1 and (want[:4000] == pick).all() and (want[4000:] == -1).any()


def test_undirected_table_first_row_wins():
    """bidirectionalised ingest: (a,b) and (b,a) share one row; several input rows for the same edge -> the first one"""
    from gigl_amd.engine import HipEngine
    src = np.array([0, 1, 2, 2, 3, 1], dtype=np.uint32)
    dst = np.array([1, 2, 1, 3, 2, 0], dtype=np.uint32)  # 1-2 three times (rows 1, 2), 2-3 twice (rows 3, 4), 0-1 twice
    feats = np.arange(6, dtype=np.float32)[:, None] * np.ones((1, 2), np.float32)
    eng = HipEngine(0)
    eng.build_from_coo(4, src, dst, is_directed=False)
    eng.load_edge_features(src, dst, feats, is_directed=False)
    rowptr, col = eng.graph_to_host()
    table = eng._efeat.cpu().numpy()
    first = {(0, 1): 0.0, (1, 2): 1.0, (2, 3): 3.0}
    for v in range(4):
        for p in range(rowptr[v], rowptr[v + 1]):
            a, b = sorted((int(col[p]), v))
            assert table[p, 0] == first[(a, b)] and table[p, 1] == first[(a, b)]
    with pytest.raises(ValueError, match="cover"):
        eng.load_edge_features(src[:2], dst[:2], feats[:2], is_directed=False)
    eng.close()


def _union(eng, rowptr, col, roots, fan):
    from gigl_amd.models import HipBatch
    tree = eng.sample_khop(roots, fan)
    u = eng.union_build(tree)
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
    o = oracle.union_build(roots, fan, nbr_o)
    return HipBatch(eng, tree, u), u, o


def _ref_edge_attr(o, ei, rowptr, col, efeat):
    nodes = np.asarray(o["nodes"])
    pos = _csc_positions(rowptr, col, nodes[ei[0].numpy()], nodes[ei[1].numpy()])
    assert (pos >= 0).all()
    return torch.from_numpy(efeat[pos])


def test_union_edge_attr(setup):
    eng, rowptr, col, x, efeat, n = setup
    roots = np.random.default_rng(4).integers(0, n, size=200).astype(np.uint32)
    batch, u, o = _union(eng, rowptr, col, roots, [6, 4])
    attr = eng.union_edge_attr(u).cpu().numpy()
    eid = eng.union_edge_ids(u).cpu().numpy()
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
    want = _ref_edge_attr(o, ei, rowptr, col, efeat).numpy()
    nn = int(u.meta[0])
    rp, re_ = u.rowptr[:nn].cpu().numpy(), u.rowend[:nn].cpu().numpy()
    used = np.concatenate([np.arange(a, b) for a, b in zip(rp, re_)])
    np.testing.assert_array_equal(attr[used], want)  # same (dst-major, ascending src) edge order as the oracle's CSR
    unused = np.setdiff1d(np.arange(len(eid)), used)
    assert (eid[unused] == -1).all() and (attr[unused] == 0).all()


@pytest.mark.parametrize("conv,share,heads,hid,out,fan", [
    ("gat", True, 1, 16, 8, [6, 4]), ("gat", True, 3, 8, 12, [5, 3]), ("gat", True, 4, 64, 128, [6, 4]),
    ("gat", True, 2, 256, 32, [5, 3]),
    ("edge_attr_gat", True, 2, 8, 8, [6, 4]), ("edge_attr_gat", False, 2, 16, 5, [4, 3, 2]),
    ("edge_attr_gat", False, 1, 80, 70, [5, 3]),
    # single-pass kernels with the online z accumulation: 4 x 64 -> 256; 2 x 256 -> 512 (two chunk rows); 1 x 512
    ("edge_attr_gat", True, 4, 64, 256, [6, 4]), ("edge_attr_gat", False, 2, 256, 512, [5, 3]),
    ("edge_attr_gat", False, 1, 512, 256, [5, 3])])
def test_gat_with_edge_features(setup, conv, share, heads, hid, out, fan):
    from gigl_amd.models_attn import GAT
    eng, rowptr, col, x, efeat, n = setup
    torch.manual_seed(heads * 7 + hid)
    L = len(fan)
    model = GAT(20, hid, out, num_layers=L, heads=heads, edge_dim=DE, conv=conv,
                share_edge_att_message_weight=share).to(eng.device)
    with torch.no_grad():
        for c in model.conv_layers:
            c.bias.normal_(0, 0.1)
    roots = np.random.default_rng(6).integers(0, n, size=100).astype(np.uint32)
    roots[:5] = np.arange(5)  # nodes with a self loop in the graph
    batch, u, o = _union(eng, rowptr, col, roots, fan)
    got = model(batch)[u.root_local[:100].long()].cpu().numpy()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
    ea = _ref_edge_attr(o, ei, rowptr, col, efeat)
    assert bool((ei[0] == ei[1]).any())  # the self-loop removal / mean-fill path is exercised
    h = torch.from_numpy(x[o["nodes"]])
    for l in range(L):
        p = f"conv_layers.{l}."
        hd = heads if l < L - 1 else 1
        w_msg = None
        if conv == "edge_attr_gat":
            w_msg = sd[p + "lin_edge.weight"] if share else sd[p + "lin_edge_message.weight"]
        h = gnn_ref.gat_conv(h, ei, sd[p + "lin.weight"], sd[p + "att_src"], sd[p + "att_dst"], sd[p + "bias"], hd,
                             edge_attr=ea, w_edge=sd[p + "lin_edge.weight"], att_edge=sd[p + "att_edge"],
                             w_edge_msg=w_msg)
        if l < L - 1:
            h = torch.relu(h)
    np.testing.assert_allclose(got, h[o["root_local"]].numpy(), rtol=1e-5, atol=1e-5)


def test_tfrecord_path_carries_edge_features_end_to_end(setup):
    """sampler -> device-encoded records (Edge.feature_values) -> native collate (edge_attr) -> GAT over the coalesced
    batch graph == fp32 restatement on the collated arrays == the in-HBM union path, per root"""
    from gigl_amd.batches import RootedNodeNeighborhoodBatch
    from gigl_amd.models import HipBatch
    from gigl_amd.models_attn import GAT
    from gigl_amd import wire
    eng, rowptr, col, x, efeat, n = setup
    torch.manual_seed(3)
    model = GAT(20, 12, 10, num_layers=2, heads=2, edge_dim=DE, conv="edge_attr_gat",
                share_edge_att_message_weight=False).to(eng.device).eval()
    model.engine = eng
    roots = np.random.default_rng(8).integers(0, n, size=64).astype(np.uint32)
    tree = eng.sample_khop(roots, [5, 3])
    buf, off = eng.encode_records(tree)
    recs = list(wire.iter_tfrecords(buf.cpu().numpy().tobytes()))
    assert len(recs) == 64
    batch = RootedNodeNeighborhoodBatch.process_raw_pyg_samples_and_collate_fn(recs)
    g = batch.graph
    assert g.edge_attr is not None and g.edge_attr.shape == (g.num_edges, DE)
    # every collated edge carries the table row of its global (src, dst) pair
    l2g = batch.condensed_node_type_to_subgraph_id_to_global_node_id[0]
    gsrc = np.array([l2g[int(v)] for v in g.edge_index[0]], dtype=np.uint32)
    gdst = np.array([l2g[int(v)] for v in g.edge_index[1]], dtype=np.uint32)
    np.testing.assert_array_equal(g.edge_attr.numpy(), efeat[_csc_positions(rowptr, col, gsrc, gdst)])
    idx = batch.condensed_node_type_to_root_node_indices_map[0]
    got = model(g.to(eng.device))[idx.to(eng.device)].cpu().numpy()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    h = g.x
    for l in range(2):
        p = f"conv_layers.{l}."
        h = gnn_ref.gat_conv(h, g.edge_index, sd[p + "lin.weight"], sd[p + "att_src"], sd[p + "att_dst"], sd[p + "bias"],
                             2 if l == 0 else 1, edge_attr=g.edge_attr, w_edge=sd[p + "lin_edge.weight"],
                             att_edge=sd[p + "att_edge"], w_edge_msg=sd[p + "lin_edge_message.weight"])
        if l == 0:
            h = torch.relu(h)
    np.testing.assert_allclose(got, h[idx].numpy(), rtol=1e-5, atol=1e-5)
    u = eng.union_build(tree)
    via_union = model(HipBatch(eng, tree, u))[u.root_local[:64].long()].cpu().numpy()
    np.testing.assert_allclose(via_union, got, rtol=1e-5, atol=1e-5)
