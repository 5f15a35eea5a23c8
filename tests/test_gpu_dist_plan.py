"""The sharded batch plan (gigl_dist_plan_*: every exchange issued by the library) end to end.

  * every rank of a 2- and 8-rank world played inside one process (in-process communicator group): the sampled
    trees are bit-identical to the oracle on the WHOLE graph and the root embeddings match oracle + gnn_ref
    (sample -> collate -> fp32 forward over the whole union graph) to 1e-5 — raw-row pull and owner-side projection;
  * the same step over a 1-rank RCCL communicator on the test GPU;
  * two PROCESSES sharing the test GPU, exchanging through the callback transport over gloo (the production C++
    path crossing a real process boundary on a 1-GPU box);
  * two RCCL ranks on two GPUs (self-skips below 2 devices)."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges
from oracle import gnn_ref

pytestmark = pytest.mark.gpu

N, D, HID, OUT = 1 << 12, 24, 32, 16
FAN = [6, 4]


def make_graph():
    s, d = rmat_edges(12, 60000, seed=77)
    rowptr, col = oracle.build_csc(N, s, d, is_directed=False)
    x = np.random.default_rng(5).standard_normal((N, D)).astype(np.float32)
    return rowptr, col, x


def make_model():
    from gigl_amd.models import GraphSAGE
    torch.manual_seed(11)
    return GraphSAGE(D, HID, OUT, num_layers=len(FAN))


def reference_rows(rowptr, col, x, model, roots, group_roots):
    """oracle sample -> oracle collate -> fp32 forward over the whole union graph, batch by batch"""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rows = []
    for g0 in range(0, roots.size, group_roots):
        rs = roots[g0:g0 + group_roots]
        nbr, _ = oracle.sample_khop(rowptr, col, rs, FAN, canonical=True)
        u = oracle.union_build(rs, FAN, nbr)
        xs = torch.from_numpy(x[u["nodes"].astype(np.int64)].astype(np.float32))
        out = gnn_ref.graphsage_forward(xs, gnn_ref.union_edge_index(u["rowptr"], u["col"]), sd, len(FAN))
        rows.append(out[torch.from_numpy(u["root_local"].astype(np.int64))])
    return torch.cat(rows).numpy()


def shard_engine(rowptr, col, x, rank, world, dtype, stream=None):
    from gigl_amd.dist import partition_csc, partition_rows
    from gigl_amd.engine import HipEngine
    e = HipEngine(0)
    if stream is not None:
        e.bind_stream(stream)
    e.load_csc(*partition_csc(rowptr, col, rank, world))
    e.load_features(torch.from_numpy(partition_rows(x, rank, world)).to(dtype))
    return e


def rank_roots(rank, b):
    r = np.random.default_rng(100 + rank).integers(0, N, size=b).astype(np.uint32)
    r[3] = r[0]  # a repeated root
    return r


def bound_for(rowptr):
    return int(3 * N + 42 * 2 + np.diff(rowptr).max())


@pytest.mark.parametrize("world,project,dtype", [(2, False, torch.float32), (8, False, torch.float16),
                                                 (2, True, torch.float32), (8, True, torch.float16),
                                                 (3, True, torch.float32), (2, "pre", torch.float32),
                                                 (3, "pre", torch.float16), (8, "pre", torch.float16)])
def test_all_ranks_in_one_process_end_to_end(world, project, dtype):
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    xq = x.astype(np.float16).astype(np.float32) if dtype == torch.float16 else x  # what the shards really hold
    model = make_model()
    w, bs = model.fused_params()
    b, gr = 96, 32
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, dtype, st) for r in range(world)]
    comms = Comm.local(engs)
    if project == "pre":  # rows projected ONCE per rank over its shard; the pull moves W_l x rows
        wdev = w[0].to(engs[0].device)
        tables = [e.project_features(wdev) for e in engs]
        plans = [DistSagePlan(comms[r], w, bs, b, FAN, group_roots=gr, max_window_end=bound_for(rowptr),
                              projected=tables[r]) for r in range(world)]
    else:
        plans = [DistSagePlan(comms[r], w, bs, b, FAN, group_roots=gr, project_on_owner=project,
                              max_window_end=bound_for(rowptr)) for r in range(world)]
    roots = [rank_roots(r, b) for r in range(world)]
    roots_d = [torch.from_numpy(r.view(np.int32)).to(engs[0].device) for r in roots]
    for _ in range(2):  # twice: buffers are reused step to step
        outs = DistSagePlan.run_local(plans, roots_d)
    st.synchronize()
    for r in range(world):
        hb = plans[r].buffers_to_host()
        assert hb["meta"][8] == 0, "bucket overflow"
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots[r], FAN, canonical=True)
        for k in range(len(FAN)):
            assert np.array_equal(hb["nbr"][k], nbr_o[k]), (r, k)
            assert np.array_equal(hb["cnt"][k], cnt_o[k]), (r, k)
        want = reference_rows(rowptr, col, xq, model, roots[r], gr)
        np.testing.assert_allclose(outs[r].cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    acc = torch.zeros(16, dtype=torch.int64, device=engs[0].device)
    with torch.cuda.stream(st):
        plans[0].stats(acc)
    st.synchronize()
    a = acc.cpu().numpy()
    nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots[0], FAN, canonical=True)
    assert a[0] == sum(int(c.sum()) for c in cnt_o) and a[13] == 0 and a[14] > 0 and a[15] > 0
    # the feature-row blocks travel at the size of their request counts, not at the buckets' capacity
    # (gigl_comm_traffic; the id / neighbour exchanges of the hops stay fixed-size)
    for c in comms:
        moved, full = c.traffic()
        assert 0 < moved < full, (moved, full)
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


@pytest.mark.parametrize("world,project,fan", [(2, False, [80, 3]), (3, "pre", [70, 2]), (8, True, [100, 3]), (3, True, [3, 100]),
                                               (2, False, [3, 100])])
def test_fanouts_beyond_the_wave_resident_selection(world, project, fan, monkeypatch):
    """fan-outs past 64 (the reference takes any int: SGSPureSparkV1Task.scala:313-388): the owners answer with the
    workgroup-per-row selection; trees bit-identical to the oracle, rows to 1e-5.  (A second fan-out past 64 takes the
    generic union instead of the leaf-global one, and pre-projected rows are refused there.)"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "FAN", fan)
    test_all_ranks_in_one_process_end_to_end(world, project, torch.float32)


def test_pre_projected_rows_refuse_a_wide_second_fanout():
    from gigl_amd._lib import GiglError
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    w, bs = make_model().fused_params()
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, 2, torch.float32, st) for r in range(2)]
    comms = Comm.local(engs)
    table = engs[0].project_features(w[0].to(engs[0].device))
    with pytest.raises(GiglError, match="second fan-out"):
        DistSagePlan(comms[0], w, bs, 32, [3, 100], max_window_end=bound_for(rowptr), projected=table)
    for c in comms:
        c.close()
    for e in engs:
        e.close()


@pytest.mark.parametrize("world,project,dtype", [(2, False, torch.float32), (3, "pre", torch.float16), (8, "pre", torch.float16)])
def test_in_step_overlap_of_the_own_block_gives_the_same_batches(world, project, dtype, monkeypatch):
    """GIGL_DIST_OVERLAP=1: a rank expands its OWN block of a hop's requests on a side stream while the peers' blocks are
    exchanged, the peers' blocks on the main stream once they arrived (dist.hip: phase_impl) — trees and rows as without"""
    monkeypatch.setenv("GIGL_DIST_OVERLAP", "1")
    test_all_ranks_in_one_process_end_to_end(world, project, dtype)


@pytest.mark.parametrize("world,aggr,pre", [(2, "sum", False), (3, "max", False), (2, "sum", True), (8, "max", False)])
def test_sharded_plan_with_sum_and_max_aggregation(world, aggr, pre):
    """gigl_dist_plan_set_aggr: the sharded step with PyG SAGEConv's other reductions (homogeneous.py:107-153 passes
    `aggr` through) against the single-GPU one-call plan of the same model over the whole graph — itself checked against
    the fp32 restatement in test_gpu_sage_options; max refuses projected rows (lin_l does not commute with it)"""
    from gigl_amd._lib import GiglError
    from gigl_amd.dist import Comm, DistSagePlan
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    rowptr, col, x = make_graph()
    torch.manual_seed(11)
    model = GraphSAGE(D, HID, OUT, num_layers=len(FAN), aggr=aggr)
    w, bs = model.fused_params()
    b, gr = 96, 32
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, torch.float32, st) for r in range(world)]
    comms = Comm.local(engs)
    kw = {}
    tables = None
    if pre:
        tables = [e.project_features(w[0].to(engs[0].device)) for e in engs]
    plans = [DistSagePlan(comms[r], w, bs, b, FAN, group_roots=gr, max_window_end=bound_for(rowptr), aggr=aggr,
                          projected=tables[r] if pre else None) for r in range(world)]
    roots = [rank_roots(r, b) for r in range(world)]
    roots_d = [torch.from_numpy(r.view(np.int32)).to(engs[0].device) for r in roots]
    outs = DistSagePlan.run_local(plans, roots_d)
    st.synchronize()
    full = HipEngine(0)
    full.load_csc(rowptr, col)
    full.load_features(x)
    model = model.to(full.device).eval()
    model.engine = full
    ref_plan = model.make_plan(full, gr, FAN)
    for r in range(world):
        want = torch.cat([ref_plan.run(roots_d[r][g0:g0 + gr].contiguous()).clone() for g0 in range(0, b, gr)])
        torch.cuda.synchronize()
        np.testing.assert_allclose(outs[r].cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    if aggr == "max":
        with pytest.raises(GiglError):
            DistSagePlan(comms[0], w, bs, b, FAN, group_roots=gr, project_on_owner=True, aggr="max")
    ref_plan.close()
    full.close()
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


@pytest.mark.parametrize("world,dtype", [(2, torch.float32), (8, torch.float16)])
def test_replicated_hot_rows_are_not_pulled(world, dtype, pre=False):
    """hub-row replication (gigl_dist_plan_set_hot_rows): the most-referenced nodes' rows are kept on every rank and read
    locally — the results do not change (trees bit-identical, embeddings 1e-5 vs the oracle), the number of rows that
    travel drops, and clearing the set restores the plain pull"""
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    xq = x.astype(np.float16).astype(np.float32) if dtype == torch.float16 else x
    model = make_model()
    w, bs = model.fused_params()
    b, gr = 96, 32
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, dtype, st) for r in range(world)]
    comms = Comm.local(engs)
    tables = [e.project_features(w[0].to(e.device)) for e in engs] if pre else [None] * world
    plans = [DistSagePlan(comms[r], w, bs, b, FAN, group_roots=gr, max_window_end=bound_for(rowptr), projected=tables[r])
             for r in range(world)]
    roots = [rank_roots(r, b) for r in range(world)]
    roots_d = [torch.from_numpy(r.view(np.int32)).to(engs[0].device) for r in roots]
    # hot set: the 5 % of the nodes that occur most often as in-neighbours (the same set on every rank)
    occ = np.bincount(col.astype(np.int64), minlength=N)
    hot = np.argsort(-occ, kind="stable")[: N // 20].astype(np.uint32)
    hot_ids = torch.from_numpy(hot.view(np.int32))
    hot_rows = torch.from_numpy(x[hot.astype(np.int64)]).to(dtype)
    if pre:  # the replicas of a pre-projected plan are W_l x rows: the owners' own projected rows, gathered
        hl = torch.from_numpy(hot.astype(np.int64))
        hot_rows = torch.stack([tables[int(v) % world][int(v) // world, :HID] for v in hl]).contiguous()

    def pulled(plan):
        acc = torch.zeros(16, dtype=torch.int64, device=engs[0].device)
        with torch.cuda.stream(st):
            plan.stats(acc)
        st.synchronize()
        return int(acc[14].item())

    DistSagePlan.run_local(plans, roots_d)
    st.synchronize()
    plain = [pulled(p) for p in plans]
    for p in plans:
        p.set_hot_rows(hot_ids, hot_rows)
    for _ in range(2):
        outs = DistSagePlan.run_local(plans, roots_d)
    st.synchronize()
    with_hot = [pulled(p) for p in plans]
    assert all(h < q for h, q in zip(with_hot, plain)) and sum(with_hot) < 0.85 * sum(plain), (plain, with_hot)
    for r in range(world):
        hb = plans[r].buffers_to_host()
        assert hb["meta"][8] == 0
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots[r], FAN, canonical=True)
        for k in range(len(FAN)):
            assert np.array_equal(hb["nbr"][k], nbr_o[k]) and np.array_equal(hb["cnt"][k], cnt_o[k])
        np.testing.assert_allclose(outs[r].cpu().numpy(), reference_rows(rowptr, col, xq, model, roots[r], gr),
                                   rtol=1e-5, atol=1e-5)
    for p in plans:
        p.set_hot_rows(None, None)
    outs2 = DistSagePlan.run_local(plans, roots_d)
    st.synchronize()
    assert [pulled(p) for p in plans] == plain
    for r in range(world):
        assert torch.equal(outs2[r], outs[r])  # same arithmetic: only where the rows are read from changed
    if not pre:  # the owner-projected plan has no dense bookkeeping: replication is refused there
        from gigl_amd import _lib
        pp = DistSagePlan(comms[0], w, bs, b, FAN, group_roots=gr, project_on_owner=True, max_window_end=bound_for(rowptr))
        with pytest.raises(_lib.GiglError):
            pp.set_hot_rows(hot_ids, hot_rows)
        pp.close()
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


def test_pull_bucket_overflow_is_reported():
    """a row bucket too small for the step fails the batch loudly (meta[GIGL_META_OVERFLOW]) instead of computing on
    missing rows"""
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    model = make_model()
    w, bs = model.fused_params()
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, 2, torch.float32, st) for r in range(2)]
    comms = Comm.local(engs)
    plans = [DistSagePlan(comms[r], w, bs, 64, FAN, pull_cap=8, max_window_end=bound_for(rowptr)) for r in range(2)]
    roots_d = [torch.from_numpy(rank_roots(r, 64).view(np.int32)).to(engs[0].device) for r in range(2)]
    # a good step first, so that stale activations of it sit in the plan's buffers
    ok_plans = [DistSagePlan(comms[r], w, bs, 64, FAN, max_window_end=bound_for(rowptr)) for r in range(2)]
    DistSagePlan.run_local(ok_plans, roots_d)
    outs = DistSagePlan.run_local(plans, roots_d)
    st.synchronize()
    assert plans[0].buffers_to_host()["meta"][8] != 0
    assert plans[0].overflowed()
    assert not ok_plans[0].overflowed()
    # the failed step's rows are NaN, never stale activations that look like embeddings
    assert torch.isnan(outs[0]).all()
    with pytest.raises(RuntimeError, match="overflow"):
        plans[0].raise_on_overflow()
    for p in ok_plans:
        p.close()
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


def test_single_rank_rccl():
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    model = make_model()
    w, bs = model.fused_params()
    st = torch.cuda.Stream()
    eng = shard_engine(rowptr, col, x, 0, 1, torch.float32, st)
    comm = Comm.rccl(eng, 0, 1, Comm.unique_id())
    roots = rank_roots(0, 64)
    for project in (False, True):
        plan = DistSagePlan(comm, w, bs, 64, FAN, project_on_owner=project, max_window_end=bound_for(rowptr))
        out = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device))
        st.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), reference_rows(rowptr, col, x, model, roots, 64), rtol=1e-5,
                                   atol=1e-5)
        plan.close()
    # the communicator alone: an all-to-all with itself is a copy
    a = torch.arange(1024, dtype=torch.int32, device=eng.device)
    bb = torch.zeros_like(a)
    with torch.cuda.stream(st):
        comm.all_to_all(a, bb)
    st.synchronize()
    assert torch.equal(a, bb)
    comm.close()
    eng.close()


def _worker(rank, world, port, transport, q):
    try:
        import torch.distributed as dist
        from gigl_amd.dist import Comm, DistSagePlan, torch_exchange
        dev_idx = rank if transport == "rccl" else 0
        torch.cuda.set_device(dev_idx)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        rowptr, col, x = make_graph()
        model = make_model()
        w, bs = model.fused_params()
        from gigl_amd.dist import partition_csc, partition_rows
        from gigl_amd.engine import HipEngine
        eng = HipEngine(dev_idx)
        st = torch.cuda.Stream(device=eng.device)
        eng.bind_stream(st)
        eng.load_csc(*partition_csc(rowptr, col, rank, world))
        eng.load_features(torch.from_numpy(partition_rows(x, rank, world)))
        comm = Comm.rccl_from_torch(eng) if transport == "rccl" else Comm.callback(eng, rank, world, torch_exchange(eng))
        roots = rank_roots(rank, 64)
        worst = 0.0
        for project in (False, True):
            plan = DistSagePlan(comm, w, bs, 64, FAN, group_roots=32, project_on_owner=project,
                                max_window_end=bound_for(rowptr))
            out = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device))
            st.synchronize()
            hb = plan.buffers_to_host()
            nbr_o, _ = oracle.sample_khop(rowptr, col, roots, FAN, canonical=True)
            assert hb["meta"][8] == 0 and all(np.array_equal(hb["nbr"][k], nbr_o[k]) for k in range(len(FAN)))
            want = reference_rows(rowptr, col, x, model, roots, 32)
            worst = max(worst, float(np.abs(out.cpu().numpy() - want).max()))
            plan.close()
        dist.barrier()
        comm.close()
        eng.close()
        dist.destroy_process_group()
        q.put((rank, "ok", worst))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))


def _spawn(world, transport):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_worker, args=(r, world, port, transport, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        res.append(q.get(timeout=300))
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"
        assert info < 1e-5, (rank, info)


def test_two_processes_one_gpu_callback_transport_over_gloo():
    _spawn(2, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the build boxes have one)")
def test_two_rccl_ranks_on_two_gpus():
    _spawn(2, "rccl")


@pytest.mark.parametrize("world,dtype", [(2, torch.float32), (8, torch.float16)])
def test_replicated_hot_rows_with_pre_projected_rows(world, dtype):
    test_replicated_hot_rows_are_not_pulled(world, dtype, pre=True)


def _gat_reference_rows(rowptr, col, x, model, roots, group_roots):
    """oracle sample -> oracle collate -> fp32 GAT forward (oracle/gnn_ref.gat_conv) over the whole union graph"""
    rows = []
    for g0 in range(0, roots.size, group_roots):
        rs = roots[g0:g0 + group_roots]
        nbr, _ = oracle.sample_khop(rowptr, col, rs, FAN, canonical=True)
        u = oracle.union_build(rs, FAN, nbr)
        h = torch.from_numpy(x[u["nodes"].astype(np.int64)].astype(np.float32))
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        L = len(model.conv_layers)
        for l, c in enumerate(model.conv_layers):
            h = gnn_ref.gat_conv(h, ei, c.lin.weight.detach().cpu(), c.att_src.detach().cpu(), c.att_dst.detach().cpu(),
                                 c.bias.detach().cpu() if c.bias is not None else None, c.heads, concat=c.concat,
                                 negative_slope=c.negative_slope)
            if l < L - 1:
                h = torch.relu(h)
        rows.append(h[torch.from_numpy(u["root_local"].astype(np.int64))])
    return torch.cat(rows).numpy()


@pytest.mark.parametrize("world,dtype", [(2, torch.float32), (3, torch.float16), (8, torch.float32)])
def test_sharded_gat_plan_every_rank_end_to_end(world, dtype):
    """gigl_dist_gat_plan_create (BASELINE configs[4]'s encoder on the hash-partitioned graph): every rank of an emulated
    world samples trees bit-identical to the oracle on the whole graph and gets root embeddings within 1e-5 of the
    oracle's collate + fp32 GAT forward (and of the single-GPU one-call GAT plan)"""
    from gigl_amd.dist import Comm
    from gigl_amd.engine import HipEngine
    from gigl_amd.models_attn import GAT
    rowptr, col, x = make_graph()
    xq = x.astype(np.float16).astype(np.float32) if dtype == torch.float16 else x
    torch.manual_seed(13)
    model = GAT(D, 16, 12, num_layers=len(FAN), heads=2)
    with torch.no_grad():
        for c in model.conv_layers:
            c.bias.normal_(0, 0.1)
    b, gr = 64, 32
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, dtype, st) for r in range(world)]
    comms = Comm.local(engs)
    model = model.to(engs[0].device)
    plans = [model.make_dist_plan(comms[r], b, FAN, group_roots=gr, max_window_end=bound_for(rowptr)) for r in range(world)]
    roots = [rank_roots(r, b) for r in range(world)]
    roots_d = [torch.from_numpy(r.view(np.int32)).to(engs[0].device) for r in roots]
    from gigl_amd.dist import DistSagePlan
    for _ in range(2):
        outs = DistSagePlan.run_local(plans, roots_d)
    st.synchronize()
    # the single-GPU one-call GAT plan over the whole graph
    whole = HipEngine(0)
    whole.load_csc(rowptr, col)
    whole.load_features(torch.from_numpy(x).to(dtype))
    single = model.make_plan(whole, gr, FAN, groups=b // gr)
    for r in range(world):
        hb = plans[r].buffers_to_host()
        assert hb["meta"][8] == 0 and not plans[r].overflowed()
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots[r], FAN, canonical=True)
        for k in range(len(FAN)):
            assert np.array_equal(hb["nbr"][k], nbr_o[k]) and np.array_equal(hb["cnt"][k], cnt_o[k])
        want = _gat_reference_rows(rowptr, col, xq, model, roots[r], gr)
        print(f"sharded GAT world={world} rank {r}: max |err| vs CPU forward {np.abs(outs[r].cpu().numpy() - want).max():.2e} "
              f"(max |row| {np.abs(want).max():.2e})")
        np.testing.assert_allclose(outs[r].cpu().numpy(), want, rtol=1e-5, atol=1e-5)  # (the north star's 1e-5)
        one = single.run(roots_d[r]).cpu().numpy()
        np.testing.assert_allclose(outs[r].cpu().numpy(), one, rtol=1e-5, atol=1e-5)
    single.close()
    whole.close()
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


@pytest.mark.parametrize("world,dtype", [(2, torch.float32), (3, torch.float16), (8, torch.float32)])
def test_staged_plan_hands_out_the_training_batch_of_a_sharded_graph(world, dtype):
    """a STAGED sharded plan (gigl_dist_plan_opts.staged): after the sampling / union / feature-pull phases every rank
    holds its batch union graph with every node numbered — bit-equal to the oracle's sample + collate over the WHOLE
    graph — and the batch's dense feature matrix (gigl_dist_plan_batch_features): own rows and rows pulled from their
    owners, bit-equal to the table's rows.  What a trainer's batch needs when the graph is hash-partitioned."""
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    xq = x.astype(np.float16).astype(np.float32) if dtype == torch.float16 else x
    b = 64
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, dtype, st) for r in range(world)]
    comms = Comm.local(engs)
    dev = engs[0].device
    w = [torch.zeros((4, 2 * D), device=dev), torch.zeros((4, 8), device=dev)]
    plans = [DistSagePlan(comms[r], w, [None, None], b, FAN, max_window_end=bound_for(rowptr), staged=True)
             for r in range(world)]
    roots = [rank_roots(r, b) for r in range(world)]
    roots_d = [torch.from_numpy(r.view(np.int32)).to(dev) for r in roots]
    with torch.cuda.stream(st):
        for _ in range(2):  # twice: buffers are reused step to step
            DistSagePlan.sample_and_pull_local(plans, roots_d)
        got = [p.batch_tensors() for p in plans]
    st.synchronize()
    for r in range(world):
        t = {k: v.cpu().numpy() for k, v in got[r].items()}
        assert t["meta"][8] == 0, "bucket overflow"
        nbr, _ = oracle.sample_khop(rowptr, col, roots[r], FAN, canonical=True)
        u = oracle.union_build(roots[r], FAN, nbr)
        n = int(t["meta"][0])
        assert n == u["nodes"].size
        assert np.array_equal(t["nodes"][:n].view(np.uint32), u["nodes"])
        assert np.array_equal(t["root_local"], u["root_local"])
        for i in range(n):
            assert np.array_equal(t["col"][t["rowptr"][i]:t["rowend"][i]], u["col"][u["rowptr"][i]:u["rowptr"][i + 1]]), i
        assert np.array_equal(t["x"][:n], xq[u["nodes"].astype(np.int64)])
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


@pytest.mark.parametrize("knob", ["pull_cap", "hop_slack"])
def test_staged_plan_reports_bucket_overflow(knob):
    """a staged step never runs the plan's last phase — where the hop / row bucket flags used to be folded into
    meta[GIGL_META_OVERFLOW] — so gigl_dist_plan_batch_graph folds them before it hands meta out: a batch whose requests
    did not fit a bucket (truncated neighbourhoods, NaN feature rows) reads as FAILED, never as a batch"""
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    world, b = 3, 64
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, torch.float32, st) for r in range(world)]
    comms = Comm.local(engs)
    dev = engs[0].device
    w = [torch.zeros((4, 2 * D), device=dev), torch.zeros((4, 8), device=dev)]
    kw = dict(pull_cap=8) if knob == "pull_cap" else dict(hop_slack=1e-6)
    plans = [DistSagePlan(comms[r], w, [None, None], b if knob == "pull_cap" else 4096, FAN,
                          max_window_end=bound_for(rowptr), staged=True, **kw) for r in range(world)]
    nb = plans[0].b
    if knob == "pull_cap":
        roots_h = [rank_roots(r, b) for r in range(world)]
    else:  # every root owned by rank 0: its hop-0 request bucket (nb / world + slack) cannot hold them
        roots_h = [(np.random.default_rng(200 + r).integers(0, N // world, size=nb) * world).astype(np.uint32) for r in range(world)]
    roots_d = [torch.from_numpy(r_.view(np.int32)).to(dev) for r_ in roots_h]
    with torch.cuda.stream(st):
        DistSagePlan.sample_and_pull_local(plans, roots_d)
        got = [p.batch_tensors() for p in plans]
    st.synchronize()
    assert any(int(t["meta"].cpu()[8]) != 0 for t in got), "an overflowed staged step must be flagged in meta"
    for r in range(world):
        if int(got[r]["meta"].cpu()[8]) != 0:
            assert plans[r].overflowed()
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()
