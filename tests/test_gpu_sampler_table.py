"""The hash threshold table (sample.hip) must be invisible: table path, direct path and the
out-of-domain fallback all return exactly the oracle's sample."""
import numpy as np
import pytest

import oracle
from helpers import rmat_edges

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gigl_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def _hub_graph(n, hubs, rng):
    """a few hub rows of very different degrees (1 .. 300k) so that windows start/end at every alignment"""
    src, dst = [], []
    for v, deg in hubs:
        s = rng.choice(n, size=deg, replace=False)
        src.append(s)
        dst.append(np.full(deg, v))
    return np.concatenate(src).astype(np.uint32), np.concatenate(dst).astype(np.uint32)


@pytest.mark.parametrize("seed", [42, 0, 7, -5, 2**31 - 1, 123456789])
def test_hub_rows_all_paths(eng, seed):
    rng = np.random.default_rng(17)
    n = 400_000
    hubs = [(0, 300_000), (1, 70_001), (2, 16_384), (3, 4_097), (4, 4_096), (5, 2_049), (6, 2_048), (7, 1_100),
            (8, 1_024), (9, 65), (10, 64), (11, 26), (12, 25), (13, 1), (399_999, 33_333), (123_456, 5_000)]
    src, dst = _hub_graph(n, hubs, rng)
    # second hop: every hub also points at node 20 so that hop-2 parents include the hubs with K = 20 + hub
    src = np.concatenate([src, np.array([h for h, _ in hubs], dtype=np.uint32)])
    dst = np.concatenate([dst, np.full(len(hubs), 20, dtype=np.uint32)])
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    roots = np.array([h for h, _ in hubs] + [20, 20, 0, 14], dtype=np.uint32)
    for fanouts in ([25, 10], [64, 3], [1, 1]):
        tree = eng.sample_khop(roots, fanouts, sampling_seed=seed)
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, fanouts, sampling_seed=seed, canonical=True)
        for k in range(2):
            assert np.array_equal(tree.cnt[k].cpu().numpy(), cnt_o[k]), (seed, fanouts, k)
            assert np.array_equal(_u32(tree.nbr[k]), nbr_o[k]), (seed, fanouts, k)


def test_many_windows_rmat(eng):
    """power-law graph, many roots: windows of the same hub at many different offsets"""
    s, d = rmat_edges(15, 1_500_000, seed=77)
    n = 1 << 15
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    assert np.diff(rowptr).max() > 4096
    eng.load_csc(rowptr, col)
    rng = np.random.default_rng(3)
    roots = rng.integers(0, n, size=1500).astype(np.uint32)
    tree = eng.sample_khop(roots, [25, 10])
    nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, [25, 10], canonical=True)
    for k in range(2):
        assert np.array_equal(tree.cnt[k].cpu().numpy(), cnt_o[k])
        assert np.array_equal(_u32(tree.nbr[k]), nbr_o[k])


def test_million_edge_hub_beyond_2_pow_30(eng):
    """BASELINE configs[3] territory (RMAT scale-30: hubs of ~10^6 in-edges, path sums up to 3 * 2^30): a hub row of
    1.2 M in-edges asked for at window offsets inside the table, straddling 2^30, far beyond it, and wrapping 2^32 —
    with the table allowed to grow past 2^30 (max_window_end says so) and with the bound unknown (-1: windows past
    the table take the workgroup-per-row direct-hash path).  Every answer == the oracle's permutation."""
    import torch
    rng = np.random.default_rng(9)
    n, deg = 1_400_000, 1_200_000
    hub_src = rng.choice(n - 1, size=deg, replace=False).astype(np.uint32) + 1
    small_src = rng.integers(1, n, size=4000).astype(np.uint32)
    src = np.concatenate([hub_src, small_src])
    dst = np.concatenate([np.zeros(deg, np.uint32), rng.integers(1, 50, size=4000).astype(np.uint32)])
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    ksums = np.array([5, (1 << 30) - 600_000, (1 << 30) + 1_000_000, 3 * (1 << 30) - 7, (1 << 32) - 500_000, 17],
                     dtype=np.uint64)
    nodes = np.array([0, 0, 0, 0, 0, 3], dtype=np.uint32)
    f, hash_add = 10, 84
    dev = eng.device
    nd = torch.from_numpy(nodes.view(np.int32)).to(dev)
    kd = torch.from_numpy(ksums.astype(np.uint32).view(np.int32)).to(dev)
    bound = int(3 * (1 << 30) + deg + hash_add)
    for mwe in (bound, -1):
        # a bound promises that no window ends beyond it: the request that wraps 2^32 is only legal without one
        keep = [i for i in range(nodes.size) if mwe < 0 or int(ksums[i]) + hash_add + deg <= mwe]
        nbr, cnt = eng.expand_frontier(nd[keep].contiguous(), kd[keep].contiguous(), f, hash_add, 1, mwe)
        nbr = nbr.cpu().numpy().view(np.uint32).reshape(-1, f)
        cnt = cnt.cpu().numpy()
        for i, (v, k) in enumerate(zip(nodes[keep].tolist(), ksums[keep].tolist())):
            row = col[rowptr[v]:rowptr[v + 1]]
            want = np.sort(oracle.hash_permutation(row, int(k) & 0xFFFFFFFF, sampling_seed=hash_add, counter=1)[:f])
            assert cnt[i] == want.size and np.array_equal(nbr[i][: want.size], want), (mwe, i)
