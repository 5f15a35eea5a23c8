"""The PEER-MAPPED route of the sharded step's feature pull (gigl_dist_plan_opts.peer_direct, csrc/dist.hip): the first layer
reads every source row where it lives — row v / world of rank (v % world)'s table, mapped into the reading process — instead
of claim -> id exchange -> owner-side gather -> row exchange -> receive buffer.  Replaces the same reference code as the
bucketed pull (python/gigl/distributed/dist_link_prediction_data_partitioner.py:560-664, the chunked feature scatter;
python/gigl/distributed/distributed_neighborloader.py:162-192, the per-batch feature RPC).

Bars: trees bit-identical to the oracle; embeddings BIT-IDENTICAL to the bucketed route's (same rows, same fp32 sums in the
same order — only where a row is read from changes) and within 1e-5 of the oracle's fp32 CPU forward."""
import os

import numpy as np
import pytest
import torch

import oracle
from test_gpu_dist_plan import (FAN, HID, N, bound_for, make_graph, make_model, rank_roots, reference_rows, shard_engine)

pytestmark = pytest.mark.gpu


def _pulled(plan, st, dev):
    acc = torch.zeros(16, dtype=torch.int64, device=dev)
    with torch.cuda.stream(st):
        plan.stats(acc)
    st.synchronize()
    return int(acc[14].item()), int(acc[13].item())


@pytest.mark.parametrize("world,pre,dtype,hot", [(2, False, torch.float32, False), (8, False, torch.float16, False),
                                                 (2, True, torch.float32, False), (3, True, torch.float16, False),
                                                 (8, True, torch.float16, True), (3, False, torch.float32, True)])
def test_peer_mapped_rows_equal_the_bucketed_route_bit_for_bit(world, pre, dtype, hot):
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    xq = x.astype(np.float16).astype(np.float32) if dtype == torch.float16 else x
    model = make_model()
    w, bs = model.fused_params()
    b, gr = 96, 32
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, dtype, st) for r in range(world)]
    dev = engs[0].device
    comms = Comm.local(engs)
    tables = [e.project_features(w[0].to(dev)) for e in engs] if pre else [None] * world
    kw = dict(group_roots=gr, max_window_end=bound_for(rowptr))
    bucketed = [DistSagePlan(comms[r], w, bs, b, FAN, projected=tables[r], **kw) for r in range(world)]
    peer = [DistSagePlan(comms[r], w, bs, b, FAN, projected=tables[r], peer_direct=True, **kw) for r in range(world)]
    addrs = [p.own_table() for p in peer]
    for p in peer:
        p.set_peer_tables(addrs)
    if hot:
        occ = np.bincount(col.astype(np.int64), minlength=N)
        hid = np.argsort(-occ, kind="stable")[: N // 20].astype(np.uint32)
        hot_ids = torch.from_numpy(hid.view(np.int32))
        if pre:
            hot_rows = torch.stack([tables[int(v) % world][int(v) // world, :HID] for v in hid.astype(np.int64)]).contiguous()
        else:
            hot_rows = torch.from_numpy(x[hid.astype(np.int64)]).to(dtype)
        for p in bucketed + peer:
            p.set_hot_rows(hot_ids, hot_rows)
    roots = [rank_roots(r, b) for r in range(world)]
    roots_d = [torch.from_numpy(r.view(np.int32)).to(dev) for r in roots]
    for _ in range(2):
        outs_b = DistSagePlan.run_local(bucketed, roots_d)
        outs_p = DistSagePlan.run_local(peer, roots_d)
    st.synchronize()
    for r in range(world):
        hb = peer[r].buffers_to_host()
        assert hb["meta"][8] == 0
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots[r], FAN, canonical=True)
        for k in range(len(FAN)):
            assert np.array_equal(hb["nbr"][k], nbr_o[k]) and np.array_equal(hb["cnt"][k], cnt_o[k]), (r, k)
        assert torch.equal(outs_p[r], outs_b[r]), f"rank {r}: the peer-mapped rows differ from the bucketed route's"
        np.testing.assert_allclose(outs_p[r].cpu().numpy(), reference_rows(rowptr, col, xq, model, roots[r], gr),
                                   rtol=1e-5, atol=1e-5)
    # what crosses the links: per OCCURRENCE on this route (no per-call dedup), per unique id on the bucketed one
    for r in range(world):
        pp, ov = _pulled(peer[r], st, dev)
        pb, _ = _pulled(bucketed[r], st, dev)
        assert ov == 0 and pp >= pb > 0, (r, pp, pb)
    if hot:  # clearing the replicas moves their occurrences back onto the links, the rows stay the same
        before = [_pulled(p, st, dev)[0] for p in peer]
        for p in peer:
            p.set_hot_rows(None, None)
        outs_c = DistSagePlan.run_local(peer, roots_d)
        st.synchronize()
        after = [_pulled(p, st, dev)[0] for p in peer]
        assert all(a > q for a, q in zip(after, before)), (before, after)
        for r in range(world):
            assert torch.equal(outs_c[r], outs_p[r])
    for p in bucketed + peer:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


@pytest.mark.parametrize("world,pre,dtype,hot", [(2, False, torch.float32, False), (8, True, torch.float16, True),
                                                 (3, True, torch.float32, False)])
def test_peer_sampled_step_has_no_exchange_and_equals_the_exchange_route(world, pre, dtype, hot):
    """gigl_dist_plan_opts.peer_sample: the graph shards are mapped too and every rank expands its own frontier over them
    (gigl_sample_khop_peer) — trees bit-identical to the oracle's and to the exchange route's, rows bit-identical to the bucketed
    route's, and NOTHING passes through the transport"""
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    xq = x.astype(np.float16).astype(np.float32) if dtype == torch.float16 else x
    model = make_model()
    w, bs = model.fused_params()
    b, gr = 96, 32
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, world, dtype, st) for r in range(world)]
    dev = engs[0].device
    comms = Comm.local(engs)
    tables = [e.project_features(w[0].to(dev)) for e in engs] if pre else [None] * world
    kw = dict(group_roots=gr, max_window_end=bound_for(rowptr))
    bucketed = [DistSagePlan(comms[r], w, bs, b, FAN, projected=tables[r], **kw) for r in range(world)]
    full = [DistSagePlan(comms[r], w, bs, b, FAN, projected=tables[r], peer_direct=True, peer_sample=True, **kw)
            for r in range(world)]
    addrs = [p.own_table() for p in full]
    graphs = [p.own_graph() for p in full]
    for p in full:
        p.set_peer_tables(addrs)
        p.set_peer_graphs([g[0] for g in graphs], [g[1] for g in graphs])
    if hot:
        occ = np.bincount(col.astype(np.int64), minlength=N)
        hid = np.argsort(-occ, kind="stable")[: N // 20].astype(np.uint32)
        hot_ids = torch.from_numpy(hid.view(np.int32))
        hot_rows = (torch.stack([tables[int(v) % world][int(v) // world, :HID] for v in hid.astype(np.int64)]).contiguous()
                    if pre else torch.from_numpy(x[hid.astype(np.int64)]).to(dtype))
        for p in bucketed + full:
            p.set_hot_rows(hot_ids, hot_rows)
    roots = [rank_roots(r, b) for r in range(world)]
    roots_d = [torch.from_numpy(r.view(np.int32)).to(dev) for r in roots]
    outs_b = DistSagePlan.run_local(bucketed, roots_d)
    st.synchronize()
    before = [c.traffic() for c in comms]
    for _ in range(2):
        outs_f = DistSagePlan.run_local(full, roots_d)
    st.synchronize()
    assert [c.traffic() for c in comms] == before, "the peer-sampled step handed bytes to the transport"
    for r in range(world):
        hb, hx = full[r].buffers_to_host(), bucketed[r].buffers_to_host()
        assert hb["meta"][8] == 0
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots[r], FAN, canonical=True)
        for k in range(len(FAN)):
            assert np.array_equal(hb["nbr"][k], nbr_o[k]) and np.array_equal(hb["cnt"][k], cnt_o[k]), (r, k)
            assert np.array_equal(hb["nbr"][k], hx["nbr"][k])
        assert torch.equal(outs_f[r], outs_b[r]), f"rank {r}: rows differ from the bucketed route's"
        np.testing.assert_allclose(outs_f[r].cpu().numpy(), reference_rows(rowptr, col, xq, model, roots[r], gr),
                                   rtol=1e-5, atol=1e-5)
    # what it refuses: a fan-out past the wave-resident selection, no window bound, graphs not set
    from gigl_amd import _lib
    with pytest.raises(_lib.GiglError):
        DistSagePlan(comms[0], w, bs, b, [80, 4], projected=None if not pre else tables[0], peer_direct=True, peer_sample=True, **kw)
    with pytest.raises(_lib.GiglError):
        DistSagePlan(comms[0], w, bs, b, FAN, projected=tables[0], peer_direct=True, peer_sample=True, group_roots=gr)
    with pytest.raises(_lib.GiglError):
        DistSagePlan(comms[0], w, bs, b, FAN, projected=tables[0], peer_sample=True, **kw)  # (needs peer_direct)
    for p in bucketed + full:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


def test_peer_mapped_plan_refuses_what_it_cannot_serve():
    from gigl_amd import _lib
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    model = make_model()
    w, bs = model.fused_params()
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, r, 2, torch.float32, st) for r in range(2)]
    comms = Comm.local(engs)
    with pytest.raises(_lib.GiglError):  # owner-side projection has no table to read in place
        DistSagePlan(comms[0], w, bs, 32, FAN, project_on_owner=True, peer_direct=True)
    with pytest.raises(_lib.GiglError):  # staged (training) batches number every node: generic union
        DistSagePlan(comms[0], w, bs, 32, FAN, staged=True, peer_direct=True)
    plans = [DistSagePlan(comms[r], w, bs, 32, FAN, peer_direct=True, max_window_end=bound_for(rowptr)) for r in range(2)]
    roots_d = [torch.from_numpy(rank_roots(r, 32).view(np.int32)).to(engs[0].device) for r in range(2)]
    with pytest.raises(_lib.GiglError, match="tables were not set"):
        DistSagePlan.run_local(plans, roots_d)
    with pytest.raises(_lib.GiglError):  # tables[rank] must be the plan's own table
        plans[0].set_peer_tables([plans[1].own_table(), plans[0].own_table()])
    for p in plans:
        p.close()
    for c in comms:
        c.close()
    for e in engs:
        e.close()


def test_a_lone_rank_needs_no_tables():
    from gigl_amd.dist import Comm, DistSagePlan
    rowptr, col, x = make_graph()
    model = make_model()
    w, bs = model.fused_params()
    st = torch.cuda.Stream()
    engs = [shard_engine(rowptr, col, x, 0, 1, torch.float32, st)]
    comms = Comm.local(engs)
    kw = dict(group_roots=32, max_window_end=bound_for(rowptr))
    a = DistSagePlan(comms[0], w, bs, 64, FAN, **kw)
    p = DistSagePlan(comms[0], w, bs, 64, FAN, peer_direct=True, **kw)
    roots = rank_roots(0, 64)
    rd = [torch.from_numpy(roots.view(np.int32)).to(engs[0].device)]
    oa = DistSagePlan.run_local([a], rd)[0]
    op = DistSagePlan.run_local([p], rd)[0]
    st.synchronize()
    assert torch.equal(oa, op)
    np.testing.assert_allclose(op.cpu().numpy(), reference_rows(rowptr, col, x, model, roots, 32), rtol=1e-5, atol=1e-5)
    a.close(), p.close(), comms[0].close(), engs[0].close()


# ---- two processes on one GPU: the other process's table through a hipIpc handle
def _worker(rank, world, port, q, transport="gloo"):
    try:
        import torch.distributed as dist
        from gigl_amd.dist import Comm, DistSagePlan, partition_csc, partition_rows, torch_exchange
        from gigl_amd.engine import HipEngine
        dev_idx = rank if transport == "rccl" else 0  # (rccl: one GPU per rank, the tables cross xGMI / PCIe for real)
        torch.cuda.set_device(dev_idx)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        rowptr, col, x = make_graph()
        model = make_model()
        w, bs = model.fused_params()
        eng = HipEngine(dev_idx)
        st = torch.cuda.Stream(device=eng.device)
        eng.bind_stream(st)
        eng.load_csc(*partition_csc(rowptr, col, rank, world))
        eng.load_features(torch.from_numpy(partition_rows(x, rank, world)))
        # the hops' exchanges: RCCL issued by the library, or gloo through the host callback
        comm = Comm.rccl_from_torch(eng) if transport == "rccl" else Comm.callback(eng, rank, world, torch_exchange(eng))
        roots = rank_roots(rank, 64)
        rd = torch.from_numpy(roots.view(np.int32)).to(eng.device)
        worst, opened = 0.0, []
        for pre in (False, True):
            table = eng.project_features(w[0].to(eng.device)) if pre else None
            kw = dict(group_roots=32, max_window_end=bound_for(rowptr), projected=table)
            bucketed = DistSagePlan(comm, w, bs, 64, FAN, **kw)
            peer = DistSagePlan(comm, w, bs, 64, FAN, peer_direct=True, **kw)
            addrs, bases = DistSagePlan.share_tables(eng, peer.own_table())
            opened += bases
            assert addrs[rank] == peer.own_table() and all(a for a in addrs)
            peer.set_peer_tables(addrs)
            # ... and the peer-SAMPLED plan: the other process's CSC shard (two allocations) through ipc handles as well
            full = DistSagePlan(comm, w, bs, 64, FAN, peer_direct=True, peer_sample=True, **kw)
            rp, cl = full.own_graph()
            rps, b1 = DistSagePlan.share_tables(eng, rp)
            cls, b2 = DistSagePlan.share_tables(eng, cl)
            opened += b1 + b2
            full.set_peer_tables(addrs)
            full.set_peer_graphs(rps, cls)
            ob = bucketed.run(rd)
            op = peer.run(rd)
            st.synchronize()
            moved0 = comm.traffic()
            of = full.run(rd)
            st.synchronize()
            assert comm.traffic() == moved0 and torch.equal(of, ob), f"pre={pre}: the peer-sampled step differs / used the transport"
            hb = peer.buffers_to_host()
            nbr_o, _ = oracle.sample_khop(rowptr, col, roots, FAN, canonical=True)
            assert hb["meta"][8] == 0 and all(np.array_equal(hb["nbr"][k], nbr_o[k]) for k in range(len(FAN)))
            assert torch.equal(op, ob), f"pre={pre}: rows read through the ipc mapping differ from the bucketed route's"
            want = reference_rows(rowptr, col, x, model, roots, 32)
            worst = max(worst, float(np.abs(op.cpu().numpy() - want).max()))
            acc = torch.zeros(16, dtype=torch.int64, device=eng.device)
            with torch.cuda.stream(st):
                peer.stats(acc)
            st.synchronize()
            assert int(acc[14]) > 0, "no row was read from the other process's table"
            dist.barrier()  # (nobody unmaps or frees a table somebody still reads)
            bucketed.close()
            peer.close()
            full.close()
        DistSagePlan.close_shared(eng, opened)
        dist.barrier()
        comm.close()
        eng.close()
        dist.destroy_process_group()
        q.put((rank, "ok", worst))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))


def _spawn(transport):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 150
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport)) for r in range(world)]
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


def test_two_processes_one_gpu_read_each_others_tables_through_ipc_handles():
    _spawn("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the build boxes have one)")
def test_two_rccl_ranks_read_each_others_tables_across_gpus():
    """the same with one GPU per rank: hop exchanges over RCCL, the peers' tables opened with hipIpcMemLazyEnablePeerAccess and
    read over the links by the first layer — bit-identical to the bucketed route, 1e-5 of the oracle"""
    _spawn("rccl")
