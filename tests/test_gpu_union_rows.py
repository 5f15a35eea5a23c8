"""Union rows of every size class: wave path (<= 64 sampled in-edges), LDS path (> 64), duplicates across
trees, against the oracle's level-ordered union (bit-exact after packing)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gigl_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _check(eng, rowptr, col, roots, fanouts):
    tree = eng.sample_khop(roots, fanouts)
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
    u = eng.union_build(tree)
    o = oracle.union_build(roots, fanouts, nbr_o)
    hops = len(fanouts)
    assert np.array_equal(u.meta.cpu().numpy()[: 3 + hops], o["meta"][: 3 + hops])
    nodes_h, rp_h, col_h = u.to_csr()
    assert np.array_equal(nodes_h, o["nodes"])
    assert np.array_equal(rp_h, o["rowptr"]) and np.array_equal(col_h, o["col"])
    assert np.array_equal(u.root_local.cpu().numpy()[: roots.size], o["root_local"])
    return o


def test_hub_reached_from_every_root(eng):
    """every root's only in-neighbour is hub 0; hub 0 has 5000 in-neighbours -> its union row merges one
    10-sample group per root (thousands of entries with duplicates): the LDS sort path"""
    n, b = 6000, 700
    hub_src = np.arange(1000, 6000, dtype=np.uint32)
    src = np.concatenate([hub_src, np.zeros(b, dtype=np.uint32)])
    dst = np.concatenate([np.zeros(5000, dtype=np.uint32), np.arange(1, b + 1, dtype=np.uint32)])
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    roots = np.arange(1, b + 1, dtype=np.uint32)
    o = _check(eng, rowptr, col, roots, [2, 10])
    hub_local = int(np.nonzero(o["nodes"] == 0)[0][0])
    assert o["rowptr"][hub_local + 1] - o["rowptr"][hub_local] > 64  # the row really is big


def test_rows_between_1_and_200(eng):
    """medium rows: nodes reached from 1..20 roots each"""
    rng = np.random.default_rng(0)
    n = 3000
    mids = np.arange(100, 130, dtype=np.uint32)  # 30 popular middle nodes with 400 in-neighbours each
    src, dst = [], []
    for m in mids:
        s = rng.choice(np.arange(1000, 3000), size=400, replace=False)
        src.append(s)
        dst.append(np.full(400, m))
    b = 400
    for r in range(200, 200 + b):  # each root points back to 3 random middle nodes
        ms = rng.choice(mids, size=3, replace=False)
        src.append(ms)
        dst.append(np.full(3, r))
    src = np.concatenate(src).astype(np.uint32)
    dst = np.concatenate(dst).astype(np.uint32)
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    roots = np.arange(200, 200 + b, dtype=np.uint32)
    _check(eng, rowptr, col, roots, [3, 10])
    _check(eng, rowptr, col, roots[:37], [2, 64])
    _check(eng, rowptr, col, np.concatenate([roots[:50], roots[:50]]), [3, 5])  # duplicate roots


def test_three_hops_levels(eng):
    from helpers import rmat_edges
    s, d = rmat_edges(11, 40000, seed=4)
    n = 1 << 11
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    eng.load_csc(rowptr, col)
    rng = np.random.default_rng(9)
    roots = rng.integers(0, n, size=120).astype(np.uint32)
    _check(eng, rowptr, col, roots, [6, 4, 3])
    _check(eng, rowptr, col, roots, [3, 3, 2, 2])
