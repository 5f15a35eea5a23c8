"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit-exact for all
integer / index work, 1e-5 for fp32 aggregation (tolerance from BASELINE.json's north_star)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import INVALID, load_fixture_graph, rmat_edges

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gigl_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def _check_tree(eng, rowptr, col, roots, fanouts):
    tree = eng.sample_khop(roots, fanouts)
    nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
    for k in range(len(fanouts)):
        assert np.array_equal(tree.cnt[k].cpu().numpy(), cnt_o[k]), f"cnt hop {k}"
        assert np.array_equal(_u32(tree.nbr[k]), nbr_o[k]), f"nbr hop {k}"
    return tree, nbr_o


def test_sampler_fixture_graph(eng, golden_dir):
    n, src, dst, _ = load_fixture_graph(golden_dir)
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=False)
    eng.load_csc(rowptr, col)
    roots = np.arange(n, dtype=np.uint32)
    for f in ([3, 3], [1, 1], [2, 5], [64, 64]):
        _check_tree(eng, rowptr, col, roots, f)


def test_graph_build_from_coo_matches_oracle(eng, golden_dir):
    n, src, dst, _ = load_fixture_graph(golden_dir)
    for directed in (False, True):
        rp_o, col_o = oracle.build_csc(n, src, dst, is_directed=directed)
        eng.build_from_coo(n, src, dst, is_directed=directed)
        rp, cl = eng.graph_to_host()
        assert np.array_equal(rp, rp_o) and np.array_equal(cl, col_o)
    s, d = rmat_edges(12, 60000, seed=11)
    rp_o, col_o = oracle.build_csc(4096, s, d, is_directed=False)
    eng.build_from_coo(4096, s, d, is_directed=False)
    rp, cl = eng.graph_to_host()
    assert np.array_equal(rp, rp_o) and np.array_equal(cl, col_o)
    with pytest.raises(RuntimeError):
        eng.build_from_coo(10, np.array([1, 50], np.uint32), np.array([2, 3], np.uint32), True)


@pytest.mark.parametrize("fanouts", [[10, 5], [25, 10], [15, 10, 5], [7]])
def test_sampler_rmat_parity(eng, fanouts):
    s, d = rmat_edges(13, 200000, seed=7)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    eng.load_csc(rowptr, col)
    rng = np.random.default_rng(1)
    roots = rng.integers(0, n, size=257).astype(np.uint32)  # with repeats, odd size
    _check_tree(eng, rowptr, col, roots, fanouts)


def test_sampler_heavy_rows_and_edge_cases(eng):
    # a star: node 0 has 20000 in-neighbours (> HEAVY_DEG, workgroup path), plus medium rows
    n = 20100
    src = np.concatenate([np.arange(1, 20001), np.arange(1, 300), np.arange(1, 70), np.array([5])]).astype(np.uint32)
    dst = np.concatenate([np.zeros(20000), np.full(299, 1), np.full(69, 2), np.array([3])]).astype(np.uint32)
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    roots = np.array([0, 1, 2, 3, 4, 0, 20099], dtype=np.uint32)  # 4 and 20099: no in-edges
    for f in ([25, 10], [64, 2], [1, 1]):
        _check_tree(eng, rowptr, col, roots, f)
    # empty batch
    t = eng.sample_khop(np.zeros(0, np.uint32), [5, 5])
    assert t.b == 0
    with pytest.raises(RuntimeError):
        eng.sample_khop(roots, [1025, 2])  # GIGL_E_UNSUPPORTED: fanout > GIGL_MAX_FANOUT


@pytest.mark.parametrize("fanouts", [[65, 3], [100, 70], [3, 200], [1024], [300, 2, 80]])
def test_fanouts_beyond_the_wave_resident_selection(eng, fanouts):
    """numNeighborsToSample is any integer in the reference (SGSPureSparkV1Task.scala:313-388): fanouts above 64 take a
    workgroup-per-row selection (sample.hip: expand_wide_kernel) with the same contract — bit-exact vs the oracle on rows
    shorter than f (copied through), a little longer than f, hubs of 20,000 neighbours, empty rows and INVALID parents"""
    rng = np.random.default_rng(5)
    n = 21000
    hub_src = np.arange(1, 20001)
    mid = [rng.choice(n, size=k, replace=False) for k in (66, 99, 101, 130, 257, 1023, 1025, 1500, 3000)]
    s, d = rmat_edges(14, 150000, seed=9)
    src = np.concatenate([hub_src] + mid + [s % n]).astype(np.uint32)
    dst = np.concatenate([np.zeros(20000)] + [np.full(m.size, 1 + i) for i, m in enumerate(mid)] + [d % n]).astype(np.uint32)
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    roots = np.concatenate([np.arange(0, 12), rng.integers(0, n, size=40), np.array([20999])]).astype(np.uint32)
    _check_tree(eng, rowptr, col, roots, fanouts)


def test_wide_fanout_downstream_union_and_forward(eng):
    """a [100, 70] batch through union build (bit-exact vs the oracle) and the GraphSAGE forward (1e-5 vs the fp32
    restatement): the rest of the path takes rows of any length"""
    from gigl_amd.models import GraphSAGE, HipBatch
    from oracle import gnn_ref
    rng = np.random.default_rng(6)
    n = 6000
    s, d = rmat_edges(13, 300000, seed=4)
    rowptr, col = oracle.build_csc(n, s % n, d % n, is_directed=False)
    eng.load_csc(rowptr, col)
    x = rng.standard_normal((n, 24)).astype(np.float32)
    eng.load_features(x)
    roots = rng.integers(0, n, size=48).astype(np.uint32)
    fanouts = [100, 70]
    tree, nbr_o = _check_tree(eng, rowptr, col, roots, fanouts)
    u = eng.union_build(tree)
    o = oracle.union_build(roots, fanouts, nbr_o)
    assert np.array_equal(u.meta.cpu().numpy()[:5], o["meta"][:5])
    nodes, rp_h, col_h = u.to_csr()
    assert np.array_equal(nodes, o["nodes"]) and np.array_equal(rp_h, o["rowptr"]) and np.array_equal(col_h, o["col"])
    torch.manual_seed(2)
    model = GraphSAGE(24, 32, 8, num_layers=2).to(eng.device).eval()
    model.engine = eng
    with torch.no_grad():
        got = model(HipBatch(eng, tree, u))[u.root_local[: roots.size].long()].cpu().numpy()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ei = torch.from_numpy(np.stack([o["col"].astype(np.int64), np.repeat(np.arange(o["rowptr"].size - 1), np.diff(o["rowptr"]))]))
    want = gnn_ref.graphsage_forward(torch.from_numpy(x[o["nodes"]]), ei, sd, 2)[o["root_local"][: roots.size]].numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    # ... and so does the one-call plan (sample -> union -> layers -> one row per root in one library call)
    plan = model.make_plan(eng, roots.size, fanouts)
    r_dev = torch.from_numpy(roots.view(np.int32)).to(eng.device)
    rows = plan.run(r_dev).cpu().numpy()
    plan.close()
    np.testing.assert_allclose(rows, want, rtol=1e-5, atol=1e-5)


def test_int32_wraparound_of_key_sum(eng):
    """the int32 adds of `i + K + seed*counter` wrap exactly like Spark's IntegerType (exercised through
    extreme sampling seeds; ids near 2^31 would need a >16 GB rowptr)"""
    s, d = rmat_edges(10, 20000, seed=3)
    rowptr, col = oracle.build_csc(1024, s, d, is_directed=False)
    eng.load_csc(rowptr, col)
    roots = np.arange(0, 1024, 7, dtype=np.uint32)
    for seed in (2**31 - 1, -(2**31), 1234567891):
        tree = eng.sample_khop(roots, [4, 3], sampling_seed=seed)
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, [4, 3], sampling_seed=seed, canonical=True)
        for k in range(2):
            assert np.array_equal(_u32(tree.nbr[k]), nbr_o[k])


def test_positives_counter_three(eng):
    s, d = rmat_edges(11, 30000, seed=9)
    n = 2048
    # out-edge graph: CSR by source == CSC of the reversed edges
    rowptr, col = oracle.build_csc(n, d, s, is_directed=True)
    eng.load_csc(rowptr, col, out_graph=True)
    roots = np.arange(0, n, 3, dtype=np.uint32)
    pos, cnt = eng.sample_positives(roots, 2)
    nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, [2], first_counter=3, canonical=True)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o[0])
    assert np.array_equal(_u32(pos), nbr_o[0])


@pytest.mark.parametrize("fanouts,b", [([10, 5], 64), ([25, 10], 300), ([4, 3, 2], 50), ([6], 33)])
def test_union_build_bit_exact(eng, fanouts, b):
    s, d = rmat_edges(12, 80000, seed=21)
    n = 1 << 12
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    eng.load_csc(rowptr, col)
    rng = np.random.default_rng(5)
    roots = rng.integers(0, n, size=b).astype(np.uint32)  # duplicates allowed
    tree, nbr_o = _check_tree(eng, rowptr, col, roots, fanouts)
    u = eng.union_build(tree)
    o = oracle.union_build(roots, fanouts, nbr_o)
    meta = u.meta.cpu().numpy()
    hops = len(fanouts)
    assert np.array_equal(meta[: 3 + hops], o["meta"][: 3 + hops]), (meta, o["meta"])
    nn, ne = int(meta[0]), int(meta[1])
    assert np.array_equal(_u32(u.nodes)[:nn], o["nodes"])
    nodes_h, rp_h, col_h = u.to_csr()  # rows keep their pre-dedup capacity on the device; packed here
    assert rp_h[-1] == ne
    assert np.array_equal(rp_h, o["rowptr"])
    assert np.array_equal(col_h, o["col"])
    assert np.array_equal(u.root_local.cpu().numpy()[:b], o["root_local"])
    # device layout invariants: monotone starts, rowend within capacity
    rp_d, re_d = u.rowptr.cpu().numpy()[: nn + 1], u.rowend.cpu().numpy()[: nn + 1]
    assert np.all(np.diff(rp_d) >= 0) and np.all(re_d[:nn] <= rp_d[1:]) and np.all(re_d >= rp_d)


def test_union_isolated_roots_only(eng):
    rowptr = np.zeros(11, dtype=np.int64)
    eng.load_csc(rowptr, np.zeros(0, np.uint32))
    roots = np.array([3, 3, 9], dtype=np.uint32)
    tree = eng.sample_khop(roots, [3, 2])
    assert tree.sampled_edges() == 0
    u = eng.union_build(tree)
    c = u.counts()
    assert c["n_nodes"] == 2 and c["n_edges"] == 0 and c["levels"] == [2, 2, 2]
    assert _u32(u.nodes)[:2].tolist() == [3, 9]
    assert u.root_local.cpu().tolist()[:3] == [0, 0, 1]
    assert u.rowptr.cpu().tolist()[:3] == [0, 0, 0]


def _ref_gather_mean(x, ids, rowptr, col, n_rows):
    d = x.shape[1]
    out = np.zeros((n_rows, 2 * d), dtype=np.float32)
    for i in range(n_rows):
        js = col[rowptr[i]:rowptr[i + 1]]
        if js.size:
            out[i, :d] = x[ids[js]].astype(np.float32).sum(axis=0, dtype=np.float32) / np.float32(js.size)
        out[i, d:] = x[ids[i]]
    return out


@pytest.mark.parametrize("d,dtype", [(100, torch.float32), (16, torch.float32), (256, torch.float32),
                                     (768, torch.float16), (1433, torch.float32), (2, torch.float32)])
def test_gather_mean_against_fp32_reference(eng, d, dtype):
    rng = np.random.default_rng(d)
    n_tab, n_loc = 5000, 700
    x = torch.from_numpy(rng.standard_normal((n_tab, d)).astype(np.float32)).to(dtype)
    ids = rng.choice(n_tab, size=n_loc, replace=False).astype(np.int32)
    deg = rng.integers(0, 40, size=n_loc)
    deg[:5] = [0, 1, 2, 3, 130]
    rowptr = np.zeros(n_loc + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n_loc, size=int(rowptr[-1])).astype(np.int32)
    n_rows = 650  # a prefix
    dev = eng.device
    out = eng.gather_mean(x.to(dev), d, torch.from_numpy(ids).to(dev), torch.from_numpy(rowptr).to(dev), None,
                          torch.from_numpy(col).to(dev), torch.tensor([n_rows], dtype=torch.int32, device=dev), n_loc)
    got = out.cpu().numpy()[:n_rows]
    want = _ref_gather_mean(x.float().numpy(), ids, rowptr, col, n_rows)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    # identity gather (layer >= 2: sources are a dense local matrix)
    h = x[:n_loc].float().contiguous()
    out2 = eng.gather_mean(h.to(dev), d, None, torch.from_numpy(rowptr).to(dev), None, torch.from_numpy(col).to(dev),
                           torch.tensor([n_rows], dtype=torch.int32, device=dev), n_loc)
    want2 = _ref_gather_mean(h.numpy(), np.arange(n_loc), rowptr, col, n_rows)
    np.testing.assert_allclose(out2.cpu().numpy()[:n_rows], want2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,k,n,act", [(1000, 200, 256, 1), (1024, 512, 47, 0), (33, 8, 7, 0), (70, 2866, 16, 1),
                                       (5, 30, 33, 1), (257, 4, 64, 0)])
def test_linear_against_fp32_reference(eng, m, k, n, act):
    g = torch.Generator().manual_seed(m * 31 + k)
    a = torch.randn(m + 9, k, generator=g)
    w = torch.randn(n, k, generator=g) / (k ** 0.5)
    bias = torch.randn(n, generator=g)
    dev = eng.device
    y = eng.linear(a.to(dev), w.to(dev), bias.to(dev), torch.tensor([m], dtype=torch.int32, device=dev), m + 9, act)
    want = a[:m].double() @ w.double().T + bias.double()
    if act:
        want = want.clamp_min(0)
    np.testing.assert_allclose(y.cpu().numpy()[:m], want.float().numpy(), rtol=1e-5, atol=1e-5)
    # no bias
    y2 = eng.linear(a.to(dev), w.to(dev), None, torch.tensor([m], dtype=torch.int32, device=dev), m + 9, 0)
    np.testing.assert_allclose(y2.cpu().numpy()[:m], (a[:m].double() @ w.double().T).float().numpy(), rtol=1e-5, atol=2e-5)


def _ref_collate_from_tree(roots, fanouts, nbr):
    """per-root RootedNodeNeighborhood node/edge lists in the reference's construction order
    (hop-1 ++ hop-2 nodes, distinct, root appended: SGSPureSparkV1Task.scala:721-735,782-815)"""
    from oracle.oracle import tree_edges
    per_root = tree_edges(roots, fanouts, nbr)
    node_lists, edge_lists = [], []
    for r, es in zip(roots.tolist(), per_root):
        es = sorted(es)
        nodes = []
        for s, d in es:
            for v in (s, d):
                if v not in nodes:
                    nodes.append(v)
        if r not in nodes:
            nodes.append(r)
        node_lists.append(np.array(nodes, dtype=np.uint32))
        edge_lists.append((np.array([e[0] for e in es], dtype=np.uint32), np.array([e[1] for e in es], dtype=np.uint32)))
    return node_lists, edge_lists


@pytest.mark.parametrize("d,hid,out,fanouts,b", [(100, 256, 47, [25, 10], 200), (1433, 16, 7, [10, 5], 16),
                                                 (32, 64, 32, [5, 5, 5], 40)])
def test_graphsage_forward_matches_reference_semantics(eng, d, hid, out, fanouts, b):
    """HIP trimmed-schedule root embeddings == fp32 CPU forward of EVERY layer over the WHOLE union graph in
    the reference's own (first-seen) numbering; tolerance 1e-5 (BASELINE.json north_star)."""
    from gigl_amd.models import GraphSAGE, HipBatch
    from oracle import gnn_ref
    s, dd = rmat_edges(12, 60000, seed=33)
    n = 1 << 12
    rowptr, col = oracle.build_csc(n, s, dd, is_directed=False)
    rng = np.random.default_rng(d)
    x = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    roots = rng.integers(0, n, size=b).astype(np.uint32)
    torch.manual_seed(0)
    model = GraphSAGE(d, hid, out, num_layers=len(fanouts)).to(eng.device)
    tree = eng.sample_khop(roots, fanouts)
    u = eng.union_build(tree)
    emb = model(HipBatch(eng, tree, u))[u.root_local[:b].long()].cpu().numpy()
    # reference path on the CPU
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, fanouts)  # permutation order, like the reference
    node_lists, edge_lists = _ref_collate_from_tree(roots, fanouts, nbr_o)
    nodes, ls, ld = oracle.collate_reference(node_lists, edge_lists)
    g2l = {int(g): i for i, g in enumerate(nodes)}
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = gnn_ref.graphsage_forward(torch.from_numpy(x[nodes]), torch.from_numpy(np.stack([ls, ld])), sd, len(fanouts))
    want = ref[[g2l[int(r)] for r in roots]].numpy()
    np.testing.assert_allclose(emb, want, rtol=1e-5, atol=1e-5)


def test_with_replacement_mode_draws_f_valid_neighbours():
    """GIGL_MODE_REPLACE (sampleWithReplacementUDF, SGSPureSparkV1Task.scala:42-50): f draws per parent with in-edges
    (also when deg < f), every draw an in-neighbour, ascending within a parent, reproducible"""
    from gigl_amd._lib import MODE_REPLACE
    from gigl_amd.engine import HipEngine
    from helpers import rmat_edges
    s, d = rmat_edges(10, 6000, seed=5)
    n = 1 << 10
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    roots = np.arange(0, 300, dtype=np.uint32)
    fan = [7, 4]
    t1 = eng.sample_khop(roots, fan, mode=MODE_REPLACE)
    nbr = [x.cpu().numpy().view(np.uint32) for x in t1.nbr]
    cnt = [x.cpu().numpy() for x in t1.cnt]
    t2 = eng.sample_khop(roots, fan, mode=MODE_REPLACE)
    assert all(np.array_equal(nbr[k], t2.nbr[k].cpu().numpy().view(np.uint32)) for k in range(2))
    parents = roots
    saw_repeat = False
    for k, f in enumerate(fan):
        for p, v in enumerate(parents.tolist()):
            seg = nbr[k][p * f:(p + 1) * f]
            if v == 0xFFFFFFFF or rowptr[v + 1] == rowptr[v]:
                assert cnt[k][p] == 0 and np.all(seg == 0xFFFFFFFF)
                continue
            row = set(col[rowptr[v]:rowptr[v + 1]].tolist())
            assert cnt[k][p] == f and all(int(x) in row for x in seg) and np.all(np.diff(seg.astype(np.int64)) >= 0)
            saw_repeat = saw_repeat or len(set(seg.tolist())) < f
        parents = nbr[k]
    assert saw_repeat
    eng.close()


def test_directed_multi_edges_are_sampled_over_the_multiset():
    """GIGL_DIRECTED_MULTI: repeated (src, dst) records stay in the rows (the reference's directed path:
    collect_list, SGSPureSparkV1Task.scala:337,442); positions are drawn over the multiset (a source named k times
    is k times as likely), each drawn id is written once — bit-exact vs the oracle's restatement of that rule"""
    from gigl_amd.engine import HipEngine
    rng = np.random.default_rng(12)
    n = 3000
    src = rng.integers(0, 400, size=60000).astype(np.uint32)  # few sources: plenty of repeated pairs
    dst = rng.integers(0, n, size=60000).astype(np.uint32)
    # a hub whose row is dominated by one source
    src = np.concatenate([src, np.full(500, 7, np.uint32), np.arange(100, 160, dtype=np.uint32)])
    dst = np.concatenate([dst, np.full(560, 5, np.uint32)])
    rowptr_m, col_m = oracle.build_csc(n, src, dst, is_directed=True, keep_multi_edges=True)
    rowptr_s, col_s = oracle.build_csc(n, src, dst, is_directed=True)
    assert col_m.size == src.size and col_s.size < col_m.size
    eng = HipEngine(0)
    eng.build_from_coo(n, src, dst, is_directed=True, keep_multi_edges=True)
    rp, cl = eng.graph_to_host()
    assert np.array_equal(rp, rowptr_m) and np.array_equal(cl, col_m)
    roots = np.concatenate([np.array([5, 5, 0, 1], dtype=np.uint32), rng.integers(0, n, size=400).astype(np.uint32)])
    for fan in ([10, 5], [25, 10], [3], [80, 3], [600]):  # (fanouts > 64: the workgroup-per-row selection)
        tree = eng.sample_khop(roots, fan)
        nbr_o, cnt_o = oracle.sample_khop(rowptr_m, col_m, roots, fan, canonical=True)
        for k in range(len(fan)):
            assert np.array_equal(tree.cnt[k].cpu().numpy(), cnt_o[k]), (fan, k)
            assert np.array_equal(tree.nbr[k].cpu().numpy().view(np.uint32), nbr_o[k]), (fan, k)
            seg = nbr_o[k].reshape(-1, fan[k])
            for row in seg[:200]:  # every parent's ids are distinct
                v = row[row != 0xFFFFFFFF]
                assert v.size == np.unique(v).size
    # the multiset matters: the hub's dominant source is drawn (almost) always, and its row yields fewer than f ids
    t = eng.sample_khop(np.array([5], np.uint32), [10])
    ids = t.nbr[0].cpu().numpy().view(np.uint32)
    assert 7 in ids and int(t.cnt[0].item()) < 10
    # downstream stays consistent: the one-call plan == oracle collate + fp32 forward on the sampled sets
    import torch
    from gigl_amd.models import GraphSAGE
    from oracle import gnn_ref
    x = rng.standard_normal((n, 16)).astype(np.float32)
    eng.load_features(x)
    torch.manual_seed(0)
    model = GraphSAGE(16, 24, 8, num_layers=2).to(eng.device)
    plan = model.make_plan(eng, roots.size, [10, 5])
    out = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
    nbr_o, _ = oracle.sample_khop(rowptr_m, col_m, roots, [10, 5], canonical=True)
    u = oracle.union_build(roots, [10, 5], nbr_o)
    sd = {k_: v_.detach().cpu() for k_, v_ in model.state_dict().items()}
    want = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"]]), gnn_ref.union_edge_index(u["rowptr"], u["col"]),
                                     sd, 2)[u["root_local"]].numpy()
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)
    eng.close()


def test_rows_beyond_the_lds_sort_capacity(eng):
    """a hub that every root of a batch reaches through a different path: its in-edge row in the batch's union graph holds
    ~45,000 distinct sampled neighbours — past the 16,384 the LDS sort takes, sorted and made distinct in global memory
    instead (union.hip: huge_row_sort_distinct; the reference has no such bound).  Generic build bit-exact vs the oracle,
    the one-call plan's rows 1e-5 vs the staged forward over it, no overflow reported."""
    from gigl_amd.models import GraphSAGE, HipBatch
    H, B, deg = 0, 1024, 100_000
    n = 2000 + deg
    # roots 1..B have the hub as their only in-neighbour; the hub's in-neighbours are 2000 .. 2000+deg
    src = np.concatenate([np.full(B, H), np.arange(2000, 2000 + deg)]).astype(np.uint32)
    dst = np.concatenate([np.arange(1, B + 1), np.full(deg, H)]).astype(np.uint32)
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, 8)).astype(np.float32)
    eng.load_features(x)
    roots = np.arange(1, B + 1, dtype=np.uint32)
    fanouts = [2, 64]
    tree, nbr_o = _check_tree(eng, rowptr, col, roots, fanouts)
    u = eng.union_build(tree)
    o = oracle.union_build(roots, fanouts, nbr_o)
    hub_local = int(np.flatnonzero(o["nodes"] == H)[0])
    assert o["rowptr"][hub_local + 1] - o["rowptr"][hub_local] > 16384
    m = u.meta.cpu().numpy()
    assert np.array_equal(m[:5], o["meta"][:5]) and u.counts().get("overflow", 0) == 0
    nodes, rp_h, col_h = u.to_csr()
    assert np.array_equal(nodes, o["nodes"]) and np.array_equal(rp_h, o["rowptr"]) and np.array_equal(col_h, o["col"])
    torch.manual_seed(4)
    model = GraphSAGE(8, 16, 4, num_layers=2).to(eng.device).eval()
    model.engine = eng
    with torch.no_grad():
        want = model(HipBatch(eng, tree, u))[u.root_local[:B].long()].cpu().numpy()
    plan = model.make_plan(eng, B, fanouts)
    rows = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
    plan.close()
    assert np.isfinite(rows).all()
    np.testing.assert_allclose(rows, want, rtol=1e-5, atol=1e-5)
