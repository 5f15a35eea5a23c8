"""The library's own radix sort + ordered distinct (csrc/sortscan.h, behind gigl_typed_plan_run) against numpy: bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gigl_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


@pytest.mark.parametrize("n,low,high", [(1, 5, 3), (2047, 8, 8), (2049, 17, 19), (300_000, 18, 18), (3_000_001, 32, 21),
                                        (70_000, 0, 12), (70_000, 9, 0)])
def test_sort_distinct_u64_matches_numpy(eng, n, low, high):
    from gigl_amd import _lib
    rng = np.random.default_rng(n)
    lo = rng.integers(0, max((1 << low) - 1, 1), n, dtype=np.uint64) if low else np.zeros(n, np.uint64)
    hi = rng.integers(0, max((1 << high) - 1, 1), n, dtype=np.uint64) if high else np.zeros(n, np.uint64)
    # many duplicates (a batch's edges repeat) and empty slots
    keys = (hi << np.uint64(32)) | lo
    dup = rng.random(n) < 0.3
    keys[dup] = keys[rng.integers(0, n, int(dup.sum()))]
    pad = np.uint64(0xFFFFFFFFFFFFFFFF)
    keys[rng.random(n) < 0.2] = pad
    want = np.unique(keys[keys != pad])
    k = torch.from_numpy(keys.view(np.int64)).to(eng.device)
    out = torch.empty(n, dtype=torch.int64, device=eng.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=eng.device)
    eng.bind_stream(torch.cuda.current_stream())
    _lib.check(eng._lib.gigl_sort_distinct_u64(eng._ctx, C.c_void_p(k.data_ptr()), n, low, high, C.c_uint64(int(pad)),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(cnt.data_ptr())), eng._ctx)
    c = int(cnt.item())
    assert c == want.size
    assert np.array_equal(out[:c].cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("n,bits", [(1, 1), (2048, 8), (5000, 13), (1_000_003, 22), (250_000, 32)])
def test_sort_distinct_u32_matches_numpy(eng, n, bits):
    from gigl_amd import _lib
    rng = np.random.default_rng(bits)
    top = (1 << bits) - 1 if bits < 32 else 0xFFFFFFFF
    keys = rng.integers(0, max(top, 1), n, dtype=np.uint64).astype(np.uint32)
    pad = np.uint32(0xFFFFFFFF)
    keys[rng.random(n) < 0.25] = pad
    want = np.unique(keys[keys != pad])
    k = torch.from_numpy(keys.view(np.int32)).to(eng.device)
    out = torch.empty(n, dtype=torch.int32, device=eng.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=eng.device)
    eng.bind_stream(torch.cuda.current_stream())
    _lib.check(eng._lib.gigl_sort_distinct_u32(eng._ctx, C.c_void_p(k.data_ptr()), n, bits, C.c_uint32(int(pad)),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(cnt.data_ptr())), eng._ctx)
    c = int(cnt.item())
    assert c == want.size
    assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
