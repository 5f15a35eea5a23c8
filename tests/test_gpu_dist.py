"""GPU side of the multi-GPU path: gigl_expand_frontier on hash-partitioned shards, and DistKHopSampler driving it
(a 1-rank RCCL group on the single test GPU: same code path as N ranks, the all_to_all is a self-copy)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

import oracle
from helpers import rmat_edges

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def graph():
    s, d = rmat_edges(12, 90000, seed=31)
    n = 1 << 12
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    return n, rowptr, col


@pytest.mark.parametrize("world", [2, 3, 8])
def test_expand_frontier_on_each_shard(graph, world):
    """emulate every rank in turn: requests for owned nodes only, answers == the oracle's per-row selection"""
    from gigl_amd.dist import partition_csc
    from gigl_amd.engine import HipEngine
    n, rowptr, col = graph
    rng = np.random.default_rng(world)
    f, hash_add = 10, 84
    for rank in range(world):
        eng = HipEngine(0)
        rp_s, col_s = partition_csc(rowptr, col, rank, world)
        eng.load_csc(rp_s, col_s)
        owned = np.arange(rank, n, world)
        nodes = rng.choice(owned, size=500).astype(np.uint32)
        nodes[::50] = 0xFFFFFFFF  # empty slots are legal
        ksums = rng.integers(0, 3 * n, size=nodes.size).astype(np.uint32)
        dev = eng.device
        bound = int(3 * n + hash_add + np.diff(rowptr).max())
        for mwe in (bound, -1):
            nbr, cnt = eng.expand_frontier(torch.from_numpy(nodes.view(np.int32)).to(dev),
                                           torch.from_numpy(ksums.view(np.int32)).to(dev), f, hash_add, world, mwe)
            nbr = nbr.cpu().numpy().view(np.uint32).reshape(-1, f)
            cnt = cnt.cpu().numpy()
            for i, (v, k) in enumerate(zip(nodes.tolist(), ksums.tolist())):
                if v == 0xFFFFFFFF:
                    assert cnt[i] == 0 and np.all(nbr[i] == 0xFFFFFFFF)
                    continue
                row = col[rowptr[v]:rowptr[v + 1]]
                want = np.sort(oracle.hash_permutation(row, k, sampling_seed=hash_add, counter=1)[:f]) if row.size else row
                assert cnt[i] == want.size and np.array_equal(nbr[i][: want.size], want)
        eng.close()


def test_dist_sampler_with_hip_expander_single_rank(graph):
    from gigl_amd.dist import DistKHopSampler, hip_expand, pull_features
    from gigl_amd.engine import HipEngine
    n, rowptr, col = graph
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = HipEngine(0)
        eng.load_csc(rowptr, col)  # world 1: the shard is the whole graph
        sampler = DistKHopSampler(hip_expand(eng, 1), eng.device)
        roots = np.random.default_rng(3).integers(0, n, size=257).astype(np.uint32)
        fan = [25, 10]
        nbr, cnt = sampler.sample_khop(torch.from_numpy(roots.astype(np.int64)), fan)
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
        for k in range(2):
            assert np.array_equal(nbr[k].cpu().numpy().astype(np.uint32), nbr_o[k])
            assert np.array_equal(cnt[k].cpu().numpy().astype(np.int32), cnt_o[k])
        x = torch.randn(n, 16, device=eng.device)
        ids = torch.from_numpy(np.unique(nbr_o[1][nbr_o[1] != 0xFFFFFFFF]).astype(np.int64)).to(eng.device)
        assert torch.equal(pull_features(ids, x, 1), x[ids])
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_device_side_frontier_exchange_all_ranks_in_one_process(graph, world):
    """HipDistKHopSampler played for every rank of a `world`-rank job inside one process: the all_to_all is done by
    hand (transpose of the per-rank buffers).  Every rank's tree must be bit-identical to sampling the whole graph in
    one process (gigl_sample_khop) and to the oracle."""
    from gigl_amd.dist import HipDistKHopSampler, partition_csc
    from gigl_amd.engine import HipEngine
    n, rowptr, col = graph
    fan = [25, 10]
    bound = int(3 * n + 42 * 2 + np.diff(rowptr).max())
    engs, samplers = [], []
    for r in range(world):
        e = HipEngine(0)
        e.load_csc(*partition_csc(rowptr, col, r, world))
        engs.append(e)
        samplers.append(HipDistKHopSampler(e, world, max_window_end=bound, slack=1.5))
    dev = engs[0].device
    rng = np.random.default_rng(world)
    roots = [rng.integers(0, n, size=300).astype(np.uint32) for _ in range(world)]
    for r in range(world):
        roots[r][::37] = 0xFFFFFFFF  # empty root slots are legal
    nodes = [torch.from_numpy(x.view(np.int32)).to(dev) for x in roots]
    ksums = [None] * world
    parent_k = list(nodes)
    m = 300
    trees = [([], []) for _ in range(world)]
    for k, f in enumerate(fan):
        reqs = [samplers[r].bucket(k, nodes[r], ksums[r], f) for r in range(world)]
        # all_to_all: rank r receives block [r] of every sender s, in sender order
        gots = [torch.stack([reqs[s][r] for s in range(world)]) for r in range(world)]
        resps = [samplers[r].serve(gots[r], k, f) for r in range(world)]
        backs = [torch.stack([resps[s][r] for s in range(world)]) for r in range(world)]
        for r in range(world):
            nbr = torch.empty(m * f, dtype=torch.int32, device=dev)
            cnt = torch.empty(m, dtype=torch.int32, device=dev)
            child = samplers[r].scatter(k, backs[r], parent_k[r], m, f, nbr, cnt)
            trees[r][0].append(nbr)
            trees[r][1].append(cnt)
            nodes[r], ksums[r], parent_k[r] = nbr, child, child
        m *= f
    whole = HipEngine(0)
    whole.load_csc(rowptr, col)
    for r in range(world):
        assert int(samplers[r].overflow.item()) == 0
        ref = whole.sample_khop(roots[r], fan)
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots[r], fan, canonical=True)
        for k in range(2):
            assert torch.equal(trees[r][0][k], ref.nbr[k]) and torch.equal(trees[r][1][k], ref.cnt[k])
            assert np.array_equal(trees[r][0][k].cpu().numpy().view(np.uint32), nbr_o[k])
    # a bucket that is too small is reported, never silently truncated
    tiny = HipDistKHopSampler(engs[0], world, max_window_end=bound, slack=0.01)
    if world > 2:
        tiny.bucket(0, torch.arange(0, 4000 * world, world, dtype=torch.int32, device=dev), None, 25)  # one owner
        assert int(tiny.overflow.item()) != 0
    for e in engs + [whole]:
        e.close()


def test_device_side_exchange_over_rccl_single_rank(graph):
    from gigl_amd.dist import HipDistKHopSampler
    from gigl_amd.engine import HipEngine
    n, rowptr, col = graph
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29750 + os.getpid() % 200))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = HipEngine(0)
        eng.load_csc(rowptr, col)
        s = HipDistKHopSampler(eng, 1)
        roots = np.random.default_rng(5).integers(0, n, size=257).astype(np.uint32)
        tree = eng.alloc_tree(257, [25, 10])
        nbr, cnt = s.sample_khop(torch.from_numpy(roots.view(np.int32)).to(eng.device), [25, 10], tree=tree)
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, [25, 10], canonical=True)
        for k in range(2):
            assert np.array_equal(nbr[k].cpu().numpy().view(np.uint32), nbr_o[k])
            assert np.array_equal(cnt[k].cpu().numpy(), cnt_o[k])
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 3])
def test_bucketed_feature_pull_all_ranks_in_one_process(world):
    """HipFeaturePuller for every rank of a world inside one process (all_to_all by hand): x == table[ids]"""
    from gigl_amd.dist import HipFeaturePuller
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    dev = eng.device
    rng = np.random.default_rng(7)
    n, d, cap = 5000, 24, 900
    table = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(dev).to(torch.float16)
    shards = [table[r::world].contiguous() for r in range(world)]
    pullers = [HipFeaturePuller(eng, world, shards[r], cap) for r in range(world)]
    ids, n_valid = [], []
    for r in range(world):
        k = int(rng.integers(100, cap))
        v = rng.choice(n, size=k, replace=False).astype(np.uint32)
        buf = np.full(cap, 0xDEADBEEF, dtype=np.uint32)  # garbage past n_valid must be ignored
        buf[:k] = v
        ids.append(torch.from_numpy(buf.view(np.int32)).to(dev))
        n_valid.append(torch.tensor(k, dtype=torch.int32, device=dev))
    reqs = [pullers[r].request(ids[r], n_valid[r]).clone() for r in range(world)]
    sc = [pullers[r].counts[:world].tolist() for r in range(world)]
    gots = [torch.stack([reqs[s][r] for s in range(world)]) for r in range(world)]
    rc = [[sc[s][r] for s in range(world)] for r in range(world)]
    rows = [pullers[r].serve(gots[r], rc[r]) for r in range(world)]
    for r in range(world):
        # what rank r gets back: from each owner s, the block of rows answering r's requests
        parts = []
        for s in range(world):
            off = sum(rc[s][:r])
            parts.append(rows[s][off: off + rc[s][r]])
        back = torch.cat(parts)
        x = pullers[r].place(back, sc[r], int(n_valid[r].item()))
        want = table[(ids[r][: int(n_valid[r].item())].to(torch.int64) & 0xFFFFFFFF)]
        assert torch.equal(x, want)
        # the copy-free variant: rows stay where they arrived, the first layer reads them through an index
        pos = pullers[r].place_index(sc[r])
        assert torch.equal(back[pos[: int(n_valid[r].item())].to(torch.int64)], want)
    eng.close()


def test_forward_over_pulled_rows_through_an_index(graph):
    """the sharded step hands the first layer the feature rows in ARRIVAL order plus an index (HipBatch.x_index) instead
    of copying them into node order; per-root outputs equal the resident-table forward, grouped batches included"""
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE, HipBatch
    n, rowptr, col = graph
    rng = np.random.default_rng(9)
    feats = torch.from_numpy((rng.standard_normal((n, 24)) / 2).astype(np.float32)).to(torch.float16)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(feats)
    torch.manual_seed(0)
    model = GraphSAGE(24, 32, 16, num_layers=2).to(eng.device)
    roots = torch.from_numpy(rng.integers(0, n, 256).astype(np.int32)).to(eng.device)
    tree = eng.sample_khop(roots, [6, 4])
    for group_roots in (None, 64):
        u = eng.union_build(tree, group_roots=group_roots)
        want = model(HipBatch(eng, tree, u))[u.root_local[:256].long()]
        nn = int(u.meta[0])
        ids = u.nodes[:nn].to(torch.int64) & 0xFFFFFFFF
        perm = torch.randperm(nn, device=eng.device)
        back = feats.to(eng.device)[ids[perm]].contiguous()  # row k of `back` = features of local node perm[k]
        pos = torch.zeros(int(u.nodes.numel()), dtype=torch.int32, device=eng.device)
        pos[perm] = torch.arange(nn, dtype=torch.int32, device=eng.device)
        got = model(HipBatch(eng, tree, u, x=back, x_index=pos))[u.root_local[:256].long()]
        assert torch.equal(got, want)
    eng.close()
