"""Split generator hashing on the device (gigl_split_hash_slots) == the C restatement in oracle/ == the host routine
of gigl_amd.split_generator, and the bulk prefill changes nothing about the strategies' results."""
import numpy as np
import pytest
import torch

import oracle.oracle as orc
from gigl_amd import wire
from gigl_amd.split_generator import (HASH_SPACE_GRANULARITY, NodeToDatasetSplitHashingAssigner, SCALA_ARRAY_SEED,
                                      TransductiveEdgeToLinkSplitHashingAssigner, edge_unique_id, murmur3_bytes_hash,
                                      node_unique_id)

pytestmark = pytest.mark.gpu
ARGS = {"train_split": "0.5", "val_split": "0.25", "test_split": "0.25"}


def _slots(eng, a, b=None, t=0, sym=False):
    import ctypes as C
    from gigl_amd._lib import check
    dev = eng.device
    ta = torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).to(dev)
    tb = torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint32).view(np.int32)).to(dev) if b is not None else None
    out = torch.empty(ta.numel(), dtype=torch.int32, device=dev)
    check(eng._lib.gigl_split_hash_slots(eng._ctx, C.c_void_p(ta.data_ptr()),
                                         C.c_void_p(tb.data_ptr()) if tb is not None else None, ta.numel(), t,
                                         1 if sym else 0, C.c_void_p(out.data_ptr())), eng._ctx)
    eng.synchronize()
    return out.cpu().numpy()


def test_device_slots_match_oracle_and_host():
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    rng = np.random.default_rng(4)
    ids = np.concatenate([rng.integers(0, 2**32, size=20000, dtype=np.uint64).astype(np.uint32),
                          np.array([0, 1, 9, 10, 99, 100, 4294967295, 1000000000, 999999999], dtype=np.uint32)])
    for t in (0, 3, 12):
        got = _slots(eng, ids, t=t)
        assert np.array_equal(got, orc.split_slots(ids, condensed_type=t))
        for x, s in zip(ids[:300].tolist(), got[:300].tolist()):
            assert s == murmur3_bytes_hash(node_unique_id(x, t), SCALA_ARRAY_SEED) % HASH_SPACE_GRANULARITY
    src = rng.integers(0, 2**31, size=20000).astype(np.uint32)
    dst = rng.integers(0, 2**31, size=20000).astype(np.uint32)
    for sym in (False, True):
        got = _slots(eng, src, dst, t=2, sym=sym)
        assert np.array_equal(got, orc.split_slots(src, dst, condensed_type=2, symmetric=sym))
        for x, y, s in zip(src[:300].tolist(), dst[:300].tolist(), got[:300].tolist()):
            if sym and x > y:
                x, y = y, x
            assert s == murmur3_bytes_hash(edge_unique_id(x, y, 2), SCALA_ARRAY_SEED) % HASH_SPACE_GRANULARITY
    if_sym = _slots(eng, src, dst, t=0, sym=True)
    assert np.array_equal(if_sym, _slots(eng, dst, src, t=0, sym=True))  # a->b and b->a land together
    eng.close()


def test_prefill_equals_on_demand_hashing():
    from gigl_amd.engine import HipEngine
    eng = HipEngine(0)
    rng = np.random.default_rng(1)
    ids = rng.integers(0, 10**7, size=5000).astype(np.uint32)
    cold, warm = NodeToDatasetSplitHashingAssigner(ARGS), NodeToDatasetSplitHashingAssigner(ARGS)
    assert warm.prefill(eng, ids) == ids.size
    assert [warm.assign_id(int(x)) for x in ids] == [cold.assign_id(int(x)) for x in ids]
    src, dst = ids[:2000], ids[2000:4000]
    e_cold, e_warm = TransductiveEdgeToLinkSplitHashingAssigner(ARGS), TransductiveEdgeToLinkSplitHashingAssigner(ARGS)
    e_warm.prefill(eng, src, dst, condensed_type=0, symmetric=e_warm.symmetric)
    edges = [wire.Edge(src_node_id=int(a), dst_node_id=int(b), condensed_edge_type=0) for a, b in zip(src, dst)]
    assert [e_warm.assign(e) for e in edges] == [e_cold.assign(e) for e in edges]
    eng.close()
