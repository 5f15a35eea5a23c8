"""gigl_sage_plan_* (one-call pipeline) == the step-by-step path == the reference semantics on the CPU."""
import threading

import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from gigl_amd.engine import HipEngine
    s, d = rmat_edges(13, 150000, seed=8)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = (np.random.default_rng(0).standard_normal((n, 100)) / 10).astype(np.float32)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    yield eng, rowptr, col, x, n
    eng.close()


def test_plan_matches_stepwise_and_oracle(setup):
    from gigl_amd.models import GraphSAGE, HipBatch
    from oracle import gnn_ref
    eng, rowptr, col, x, n = setup
    torch.manual_seed(1)
    model = GraphSAGE(100, 64, 47, num_layers=2).to(eng.device)
    b, fan = 300, [25, 10]
    plan = model.make_plan(eng, b, fan)
    rng = np.random.default_rng(2)
    for it in range(3):
        roots = rng.integers(0, n, size=b).astype(np.uint32)
        r_dev = torch.from_numpy(roots.view(np.int32)).to(eng.device)
        out = plan.run(r_dev).cpu().numpy()
        # step by step through the separate entry points
        tree = eng.sample_khop(roots, fan)
        u = eng.union_build(tree)
        ref_steps = model(HipBatch(eng, tree, u))[u.root_local[:b].long()].cpu().numpy()
        # same kernels; the plan's union keeps pure leaves as global ids (rows of level 1 are summed in global-id
        # order instead of local-id order): equal up to fp32 summation order
        np.testing.assert_allclose(out, ref_steps, rtol=2e-6, atol=2e-6)
        # integer side of the plan's last batch == oracle
        hb = plan.last_batch_to_host()
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
        for k in range(2):
            assert np.array_equal(hb["nbr"][k], nbr_o[k]) and np.array_equal(hb["cnt"][k], cnt_o[k])
        o = oracle.union_build(roots, fan, nbr_o)
        # the plan's union is leaf-global: nodes of level < hops only, same numbering, same unique-edge count
        n1 = int(o["meta"][3])
        assert np.array_equal(hb["meta"][1:4], o["meta"][1:4]) and hb["meta"][0] == n1 and hb["meta"][4] == n1
        n0 = int(o["meta"][2])
        # level 0 (roots) numbered identically; level 1 is the same SET (a level-1 node's first stream position may
        # differ: its leaf occurrences are not inserted, so the order inside level 1 is the plan's own)
        assert np.array_equal(hb["nodes"][:n0], o["nodes"][:n0])
        assert np.array_equal(np.sort(hb["nodes"][n0:n1]), np.sort(o["nodes"][n0:n1]))
        assert np.array_equal(hb["root_local"], o["root_local"])
        # rows as sets of GLOBAL source ids per global destination: level-0 rows hold local ids, level-1 rows the
        # global ids of their sources
        ref_rows = {int(o["nodes"][i]): np.sort(o["nodes"][o["col"][o["rowptr"][i]:o["rowptr"][i + 1]]].astype(np.int64))
                    for i in range(n1)}
        for i in range(n1):
            mine = hb["col"][hb["rowptr"][i]:hb["rowend"][i]].astype(np.int64) & 0xFFFFFFFF
            mine = np.sort(hb["nodes"][mine].astype(np.int64)) if i < n0 else np.sort(mine)
            assert np.array_equal(mine, ref_rows[int(hb["nodes"][i])])
        # fp32 CPU forward over the whole union graph (reference execution order)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
        want = gnn_ref.graphsage_forward(torch.from_numpy(x[o["nodes"]]), ei, sd, 2)[o["root_local"]].numpy()
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)
    plan.close()


def _oracle_rows(rowptr, col, x, roots, fan, model, L=2):
    """oracle sample -> collate -> fp32 CPU forward over the whole union graph (the reference's execution order): the
    roots' rows"""
    from oracle import gnn_ref
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
    o = oracle.union_build(roots, fan, nbr_o)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])
    return gnn_ref.graphsage_forward(torch.from_numpy(x[o["nodes"]].astype(np.float32)), ei, sd, L)[o["root_local"]].numpy()


@pytest.mark.parametrize("aggr,scale,bias", [("mean", 1.0, True), ("sum", 1.0, True), ("mean", 300.0, True),
                                             ("mean", 1e-4, False)])
def test_fused_two_layer_projection_equals_the_separate_layers_and_the_oracle(setup, aggr, scale, bias):
    """100 -> 256 -> 47 (BASELINE configs[1]'s model): the plan applies the last layer's [W_l | W_r] to the hidden rows
    inside the first projection (gigl_sage_plan_fused_layers) — same rows as the layers run apart (GIGL_PLAN_NO_FUSE2) and
    within 1e-5 of the fp32 CPU forward of the reference's execution order; table and weights at several scales (the
    second product's half split is bounded from the first one's scales), mean and sum, groups of batches"""
    import os
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    eng0, rowptr, col, x0, n = setup
    x = (x0 * scale).astype(np.float32)
    eng = HipEngine(0)
    st = torch.cuda.Stream()
    eng.bind_stream(st)  # (hipGraph replay needs a created stream)
    torch.cuda.set_stream(st)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    torch.manual_seed(11)
    model = GraphSAGE(100, 256, 47, num_layers=2, conv_kwargs={"aggr": aggr, "bias": bias}).to(eng.device)
    if scale != 1.0:
        with torch.no_grad():
            for q in model.parameters():
                q.mul_(1.0 / min(scale, 30.0) if scale > 1 else 3.0)
    b, fan, G = 200, [25, 10], 3
    plan = model.make_plan(eng, b, fan, groups=G)
    assert plan.fused_layers()
    os.environ["GIGL_PLAN_NO_FUSE2"] = "1"
    try:
        apart = model.make_plan(eng, b, fan, groups=G)
    finally:
        del os.environ["GIGL_PLAN_NO_FUSE2"]
    assert not apart.fused_layers()
    rng = np.random.default_rng(7)
    for it in range(2):
        roots = rng.integers(0, n, size=G * b).astype(np.uint32)
        r_dev = torch.from_numpy(roots.view(np.int32)).to(eng.device)
        out = plan.run(r_dev).cpu().numpy()
        ref = apart.run(r_dev).cpu().numpy()
        top = np.abs(ref).max()
        assert np.isfinite(out).all() and out.shape == (G * b, 47)
        assert np.abs(out - ref).max() <= 4e-6 * top
        for gi in range(G if aggr == "mean" else 0):  # (the CPU restatement is SAGEConv's default mean)
            want = _oracle_rows(rowptr, col, x, roots[gi * b:(gi + 1) * b], fan, model)
            assert np.abs(out[gi * b:(gi + 1) * b] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-30)
    # replayed as a hipGraph: same bits
    plan.use_graph(True)
    a1 = plan.run(r_dev).clone()
    a2 = plan.run(r_dev).clone()
    assert torch.equal(a1, a2) and np.array_equal(a1.cpu().numpy(), out)
    torch.cuda.set_stream(torch.cuda.default_stream())
    plan.close()
    apart.close()
    eng.close()


def test_bench_size_batch_against_the_oracle_forward():
    """one B = 1024, [25, 10] batch group of the HEADLINE's shape — a products-shaped power-law graph (2^18 nodes here, the
    bench's generator and degree skew), D = 100 fp32, GraphSAGE 100 -> 256 -> 47, 64 batches per call as bench.py runs them
    — against oracle.sample_khop -> union_build -> gnn_ref.graphsage_forward at 1e-5 (homogeneous.py:107-153)"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    s, d_ = rmat_edges(18, 6_000_000, seed=2)
    n = 1 << 18
    rowptr, col = oracle.build_csc(n, s, d_, is_directed=False)
    x = np.random.default_rng(1234).standard_normal((n, 100)).astype(np.float32)
    eng = HipEngine(0)
    st = torch.cuda.Stream()
    eng.bind_stream(st)
    torch.cuda.set_stream(st)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    torch.manual_seed(0)
    model = GraphSAGE(100, 256, 47, num_layers=2).to(eng.device)
    B, G, fan = 1024, 64, [25, 10]
    plan = model.make_plan(eng, B, fan, groups=G)
    plan.use_graph(True)
    roots = np.random.default_rng(42).permutation(n)[:G * B].astype(np.uint32)
    r_dev = torch.from_numpy(roots.view(np.int32)).to(eng.device)
    plan.run(r_dev)
    out = plan.run(r_dev).cpu().numpy()  # (the replayed graph)
    assert plan.fused_layers()
    for gi in (0, 37, G - 1):
        want = _oracle_rows(rowptr, col, x, roots[gi * B:(gi + 1) * B], fan, model)
        err = np.abs(out[gi * B:(gi + 1) * B] - want).max()
        print(f"batch {gi}: max |err| = {err:.3e} of max |row| = {np.abs(want).max():.3e}")
        assert err <= 1e-5 * np.abs(want).max()
    torch.cuda.set_stream(torch.cuda.default_stream())
    plan.close()
    eng.close()


def test_plans_on_several_streams_and_threads(setup):
    """S ctxs sharing one resident graph, each on its own stream and host thread: same results as serial"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    eng, rowptr, col, x, n = setup
    torch.manual_seed(3)
    model = GraphSAGE(100, 32, 16, num_layers=2).to(eng.device)
    b, fan, S, iters = 128, [10, 5], 3, 6
    rng = np.random.default_rng(5)
    all_roots = torch.from_numpy(rng.integers(0, n, size=(S * iters, b)).astype(np.int32)).to(eng.device)
    serial_plan = model.make_plan(eng, b, fan)
    want = [serial_plan.run(all_roots[i]).clone() for i in range(S * iters)]
    torch.cuda.synchronize()
    engines, plans, streams = [], [], []
    for s in range(S):
        e = HipEngine(0)
        e.share_resident(eng)
        st = torch.cuda.Stream(device=eng.device)
        e.bind_stream(st)
        engines.append(e)
        streams.append(st)
        plans.append(model.make_plan(e, b, fan))
    got = [None] * (S * iters)

    def worker(s):
        with torch.cuda.stream(streams[s]):
            for i in range(s, S * iters, S):
                got[i] = plans[s].run(all_roots[i], out=torch.empty((b, 16), device=eng.device))
        streams[s].synchronize()

    ths = [threading.Thread(target=worker, args=(s,)) for s in range(S)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    for i in range(S * iters):
        assert torch.equal(got[i], want[i]), i
    for e in engines:
        e.close()
    serial_plan.close()


def test_plan_reports_batches_that_do_not_fit_its_workspace():
    """a root that is its own sampled neighbour makes the children of that occurrence inner-level nodes: with b = 1,
    fanout [64, 64] the activation workspace holds 1 + 64 rows, the batch needs more -> nothing is computed for it
    and meta[GIGL_META_OVERFLOW] says so (instead of writing past the buffers)"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    n, d = 400, 8
    found = None
    for r in range(1, 200):  # a root whose hop-1 sample contains the root itself (self loop in the CSC)
        nbrs = np.unique(np.concatenate([[r], np.arange(200, 320)])).astype(np.uint32)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        rowptr[r + 1:] = nbrs.size
        nbr_o, _ = oracle.sample_khop(rowptr, nbrs, np.array([r], np.uint32), [64, 64], canonical=True)
        if r in nbr_o[0]:
            u = oracle.union_build(np.array([r], np.uint32), [64, 64], nbr_o)
            if int(u["meta"][3]) > 65:
                found = (r, rowptr, nbrs)
                break
    assert found is not None
    r, rowptr, col = found
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(np.random.default_rng(0).standard_normal((n, d)).astype(np.float32))
    model = GraphSAGE(d, 8, 4, num_layers=2).to(eng.device)
    plan = model.make_plan(eng, 1, [64, 64])
    plan.run(torch.tensor([r], dtype=torch.int32, device=eng.device))
    hb = plan.last_batch_to_host()
    assert hb["meta"][8] != 0  # GIGL_META_OVERFLOW
    # the step-by-step path (public union + per-layer calls sized by the caller) still computes this batch
    tree = eng.sample_khop(np.array([r], np.uint32), [64, 64])
    u = eng.union_build(tree)
    assert u.counts()["levels"][1] > 65
    plan.close()
    eng.close()


@pytest.mark.parametrize("aggr,fan", [("sum", [25, 10]), ("max", [25, 10]), ("max", [6, 4, 3])])
def test_plan_with_sum_and_max_reductions(setup, aggr, fan):
    """SAGEConv aggr sum / max through the one-call plan (gigl_sage_plan_set_aggr) == the staged forward"""
    from gigl_amd.models import GraphSAGE, HipBatch
    eng, rowptr, col, x, n = setup
    torch.manual_seed(4)
    model = GraphSAGE(100, 32, 16, num_layers=len(fan), aggr=aggr).to(eng.device)
    b = 128
    plan = model.make_plan(eng, b, fan)
    roots = np.random.default_rng(9).integers(0, n, size=b).astype(np.uint32)
    out = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
    tree = eng.sample_khop(roots, fan)
    u = eng.union_build(tree)
    want = model(HipBatch(eng, tree, u))[u.root_local[:b].long()].cpu().numpy()
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)


def test_plan_in_two_parts_equals_the_one_call(setup):
    """gigl_sage_plan_run_part: GIGL_PLAN_PART_GRAPH (sample + union) then GIGL_PLAN_PART_LAYERS (layers + rows) on the
    plan's stream == gigl_sage_plan_run, bit for bit (callers pipeline batch sets across two plans with it)"""
    import ctypes as C
    from gigl_amd import _lib
    from gigl_amd.models import GraphSAGE
    eng, rowptr, col, x, n = setup
    torch.manual_seed(1)
    model = GraphSAGE(100, 64, 47, num_layers=2).to(eng.device)
    b, fan = 256, [25, 10]
    plan = model.make_plan(eng, b, fan)
    rng = np.random.default_rng(5)
    for _ in range(2):
        roots = torch.from_numpy(rng.integers(0, n, size=b).astype(np.uint32).view(np.int32)).to(eng.device)
        want = plan.run(roots).clone()
        got = torch.full_like(want, float("nan"))
        for part in (1, 2):
            _lib.check(eng._lib.gigl_sage_plan_run_part(plan._plan, C.c_void_p(roots.data_ptr()), 42, _lib.MODE_SPARK_HASH,
                                                        C.c_void_p(got.data_ptr()), part), eng._ctx)
        eng.synchronize()
        assert torch.equal(want, got)
    plan.close()


def test_first_layer_half_split_is_bounded_by_the_operands(setup):
    """gigl_sage_plan_half_split: the first projection runs over two fp16 planes per operand (three MFMAs instead of
    six), each operand pre-scaled by a power of two (table: from its largest magnitude; weights: found on the device at
    every run) and the scales undone in the epilogue — so the layer stays within 1e-5 of the fp64 forward of the same
    batch whatever the operands' scale: table x 1e-4 / x 1e6, weights x 1e-3 / x 1e6, one column x 1e-5, weights
    rewritten in place between runs.  A table dominated by a few outliers (typical magnitude 2^10 below the largest)
    keeps the bf16 planes, and `half_split()` tells"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    from oracle import gnn_ref
    eng, rowptr, col, x, n = setup
    torch.manual_seed(3)
    b, fan = 256, [15, 10]
    roots = np.random.default_rng(4).integers(0, n, size=b).astype(np.uint32)
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
    o = oracle.union_build(roots, fan, nbr_o)
    ei = gnn_ref.union_edge_index(o["rowptr"], o["col"])

    def run_check(plan, engine, feats, model, rtol):
        out = plan.run(torch.from_numpy(roots.view(np.int32)).to(engine.device)).double().cpu().numpy()
        sd = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
        want = gnn_ref.graphsage_forward(torch.from_numpy(feats[o["nodes"]]).double(), ei, sd, 2)[o["root_local"]].numpy()
        scale = np.abs(want).max()
        assert np.isfinite(out).all()
        assert np.abs(out - want).max() <= rtol * scale, (np.abs(out - want).max(), scale)
        return np.abs(out - want).max() / scale

    def check(engine, feats, model, expect_half, rtol=1e-5):
        plan = model.make_plan(engine, b, fan)
        assert plan.half_split() == expect_half
        err = run_check(plan, engine, feats, model, rtol)
        plan.close()
        return err

    def on_table(feats, model, expect_half, rtol=1e-5):
        e = HipEngine(0)
        try:
            e.load_csc(rowptr, col)
            e.load_features(feats)
            return check(e, feats, model, expect_half, rtol)
        finally:
            e.close()

    def fresh(scale_w=None):
        m = GraphSAGE(100, 256, 64, num_layers=2).to(eng.device)
        if scale_w is not None:
            with torch.no_grad():
                m.conv_layers[0].lin_l.weight.mul_(scale_w)
                m.conv_layers[0].lin_r.weight.mul_(scale_w)
        return m

    model = fresh()
    err_half = check(eng, x, model, True)
    assert err_half < 3e-6  # (what the two-plane split leaves: ~2^-22 per product)
    # weights far above / far below the half range: the scale is found from the weights on the device
    assert check(eng, x, fresh(1e6), True) < 3e-6
    assert check(eng, x, fresh(1e-3), True) < 3e-6
    assert check(eng, x, fresh(1e-7), True) < 3e-6
    # tables far above / far below the half range
    assert on_table((x * 1e6).astype(np.float32), model, True) < 3e-6
    assert on_table((x * 1e-4).astype(np.float32), model, True) < 3e-6
    assert on_table((x * 1e-9).astype(np.float32), model, True) < 3e-6
    # one column five orders of magnitude below the others
    x_mixed = x.copy()
    x_mixed[:, 7] *= 1e-5
    assert on_table(x_mixed, model, True) < 3e-6
    # small table AND small weights together
    assert on_table((x * 1e-4).astype(np.float32), fresh(1e-3), True) < 3e-6
    # a few outliers six orders above the typical magnitude: the bf16 planes (fp32's range and per-element precision)
    x_out = x.copy()
    x_out[5, 3] = 4e6
    on_table(x_out, model, False)
    # weights rewritten IN PLACE between runs of one plan (a training loop writes the buffers the plan borrows; nobody
    # calls set_weights): the device-side scale follows them
    m2 = fresh()
    plan = m2.make_plan(eng, b, fan)
    assert plan.half_split()
    run_check(plan, eng, x, m2, 1e-5)
    w0 = plan._keep[0][0]  # the fused first-layer weight [W_l | W_r] the plan reads
    for factor in (3e-4, 1e9):
        with torch.no_grad():
            w0.mul_(factor)
            m2.conv_layers[0].lin_l.weight.mul_(factor)
            m2.conv_layers[0].lin_r.weight.mul_(factor)
        assert run_check(plan, eng, x, m2, 1e-5) < 3e-6
    plan.close()
    # a sum over up to 15 rows of a table that reaches 5,000: the table's scale accounts for the fan-out
    eng3 = HipEngine(0)
    try:
        eng3.load_csc(rowptr, col)
        x_mid = (x * (5000.0 / np.abs(x).max())).astype(np.float32)
        eng3.load_features(x_mid)
        m_sum = GraphSAGE(100, 256, 64, num_layers=2, aggr="sum").to(eng3.device)
        plan = m_sum.make_plan(eng3, b, fan)
        assert plan.half_split()
        out = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng3.device))
        assert torch.isfinite(out).all()
        from gigl_amd.models import HipBatch
        tree = eng3.sample_khop(roots, fan)
        u = eng3.union_build(tree)
        want = m_sum(HipBatch(eng3, tree, u))[u.root_local[:b].long()]
        assert (out - want).abs().max() <= 1e-5 * want.abs().max()
        plan.close()
    finally:
        eng3.close()


@pytest.mark.parametrize("d_in,n_out", [(4, 1), (12, 47), (36, 48), (128, 16), (260, 5)])
def test_fused_projection_shapes_against_the_separate_layers(setup, d_in, n_out):
    """linear_fused2x_kernel over input widths that leave a ragged last 32-k chunk (12, 36, 260), a single chunk (4: K = 8),
    a [mean | self] boundary inside a chunk and on one (128), and output widths 1 .. 48: the fused plan's rows against the
    layers run apart (GIGL_PLAN_NO_FUSE2: linear_split_kernel twice + the plain gathers) at 4e-6 of the largest row entry,
    row counts that leave a partial last 128-row tile (homogeneous.py:107-153)"""
    import os
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    eng0, rowptr, col, x0, n = setup
    rng = np.random.default_rng(100 + d_in)
    x = rng.standard_normal((n, d_in)).astype(np.float32)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    torch.manual_seed(d_in)
    model = GraphSAGE(d_in, 256, n_out, num_layers=2).to(eng.device)
    b, fan, G = 77, [7, 5], 2
    plan = model.make_plan(eng, b, fan, groups=G)
    assert plan.fused_layers() and plan.fused_planes() in (1, 2)
    os.environ["GIGL_PLAN_NO_FUSE2"] = "1"
    try:
        apart = model.make_plan(eng, b, fan, groups=G)
    finally:
        del os.environ["GIGL_PLAN_NO_FUSE2"]
    assert not apart.fused_layers()
    roots = rng.integers(0, n, size=G * b).astype(np.uint32)
    r_dev = torch.from_numpy(roots.view(np.int32)).to(eng.device)
    out = plan.run(r_dev).cpu().numpy()
    ref = apart.run(r_dev).cpu().numpy()
    assert np.isfinite(out).all() and out.shape == (G * b, n_out)
    assert np.abs(out - ref).max() <= 4e-6 * np.abs(ref).max()
    plan.close()
    apart.close()
    eng.close()
