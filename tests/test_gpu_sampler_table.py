"""The range-top-K hash table (sample.hip) must be invisible: table path, direct path and the
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
