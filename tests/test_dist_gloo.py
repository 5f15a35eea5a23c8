"""N>1 path on CPU: world_size-2 `gloo` processes.  The hash-partitioned sampler (per-hop all_to_all frontier
exchange) and the feature pull must reproduce exactly what one process computes on the whole graph; the
owner-side expansion is injected (the oracle here, HipEngine.expand_frontier on the GPU box)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from helpers import rmat_edges
    s, d = rmat_edges(10, 12000, seed=13)
    n = 1 << 10
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = np.random.default_rng(1).standard_normal((n, 8)).astype(np.float32)
    return n, rowptr, col, x


def _worker(rank, world, port, fanouts, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from gigl_amd.dist import DistKHopSampler, partition_csc, partition_rows, pull_features, shard_batches
        n, rowptr, col, x = _graph()
        rp_s, col_s = partition_csc(rowptr, col, rank, world)
        x_s = torch.from_numpy(partition_rows(x, rank, world))

        def expand(nodes, ksums, f, hash_add):
            m = nodes.numel()
            nbr = torch.full((m, f), 0xFFFFFFFF, dtype=torch.int64)
            cnt = torch.zeros(m, dtype=torch.int64)
            for i, (v, k) in enumerate(zip(nodes.tolist(), ksums.tolist())):
                assert v % world == rank  # only owned nodes are ever requested
                row = col_s[rp_s[v // world]:rp_s[v // world + 1]]
                if row.size:
                    sel = np.sort(oracle.hash_permutation(row, k, sampling_seed=hash_add, counter=1)[:f])
                    nbr[i, : sel.size] = torch.from_numpy(sel.astype(np.int64))
                    cnt[i] = sel.size
            return nbr, cnt

        sampler = DistKHopSampler(expand, torch.device("cpu"))
        all_roots = np.random.default_rng(7).integers(0, n, size=(6, 40)).astype(np.uint32)
        ok = True
        for bi in shard_batches(6, rank, world):
            roots = all_roots[bi]
            nbr, cnt = sampler.sample_khop(torch.from_numpy(roots.astype(np.int64)), fanouts)
            nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
            for k in range(len(fanouts)):
                ok &= np.array_equal(nbr[k].numpy().astype(np.uint32), nbr_o[k])
                ok &= np.array_equal(cnt[k].numpy().astype(np.int32), cnt_o[k])
            ids = torch.from_numpy(np.unique(np.concatenate([roots] + [a[a != 0xFFFFFFFF] for a in nbr_o])).astype(np.int64))
            rows = pull_features(ids, x_s, world)
            ok &= bool(np.array_equal(rows.numpy(), x[ids.numpy()]))
        # ranks do different numbers of batches? no: 6 batches / 2 ranks each 3 -> collective counts match
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ret[rank] = bool(t.item() == 1.0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fanouts", [[5, 3], [4, 3, 2]])
def test_hash_partitioned_sampling_and_feature_pull_world2(fanouts):
    world = 2
    port = 29500 + (os.getpid() % 2000) + len(fanouts)
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, fanouts, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert dict(ret) == {0: True, 1: True}


def test_partition_helpers():
    sys.path.insert(0, ROOT)
    from gigl_amd.dist import partition_csc, partition_rows, shard_batches
    n, rowptr, col, x = _graph()
    seen = 0
    for r in range(3):
        rp, cl = partition_csc(rowptr, col, r, 3)
        owned = np.arange(r, n, 3)
        assert rp.size == owned.size + 1
        for i, v in enumerate(owned[:50]):
            assert np.array_equal(cl[rp[i]:rp[i + 1]], col[rowptr[v]:rowptr[v + 1]])
        seen += cl.size
        assert np.array_equal(partition_rows(x, r, 3), x[owned])
    assert seen == col.size
    assert sorted(list(shard_batches(7, 0, 2)) + list(shard_batches(7, 1, 2))) == list(range(7))


def _a2a_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from gigl_amd.dist import _gloo_all_to_all
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n = 1000
        # block p of rank r's send buffer = bytes (r, p, i): what must arrive as block r at rank p
        send = torch.cat([(torch.arange(n) * 7 + rank * 31 + p * 5).to(torch.uint8) for p in range(world)])
        recv = torch.zeros_like(send)
        _gloo_all_to_all(recv, send, world)
        want = torch.cat([(torch.arange(n) * 7 + s * 31 + rank * 5).to(torch.uint8) for s in range(world)])
        ret[rank] = bool(torch.equal(recv, want))
    finally:
        dist.destroy_process_group()


def test_host_transport_of_the_callback_communicator_world3():
    """the byte mover behind Comm.callback / torch_exchange (the transport a GPU test drives between two processes
    sharing one device) is an all-to-all: block p of rank r's buffer arrives as block r at rank p"""
    world = 3
    port = 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_a2a_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True, 2: True}
