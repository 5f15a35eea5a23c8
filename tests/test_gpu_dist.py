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
