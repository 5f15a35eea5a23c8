"""GIGL_MODE_FAST (the documented non-parity sampler: one counter-based draw inside each of f equal strata of the row,
csrc/sample.hip expand_fast_kernel): not set-equal to the reference — SURVEY.md 8(d) asks for it to be timed, and for it
to be a VALID sample: exactly min(deg, f) distinct in-neighbours per parent, reproducible, a function of the path alone,
every neighbour equally likely; and the one-call plan in fast mode equals the stepwise path over the fast trees."""
import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges

pytestmark = pytest.mark.gpu
INV = 0xFFFFFFFF


@pytest.fixture(scope="module")
def setup():
    from gigl_amd.engine import HipEngine
    s, d = rmat_edges(13, 200000, seed=21)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = (np.random.default_rng(0).standard_normal((n, 64)) / 8).astype(np.float32)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    yield eng, rowptr, col, n
    eng.close()


def test_fast_samples_are_valid_and_reproducible(setup):
    from gigl_amd._lib import MODE_FAST
    eng, rowptr, col, n = setup
    fan = [25, 10]
    rng = np.random.default_rng(1)
    roots = rng.integers(0, n, size=512).astype(np.uint32)
    t = eng.sample_khop(roots, fan, mode=MODE_FAST)
    t2 = eng.sample_khop(roots, fan, mode=MODE_FAST)
    par = eng.sample_khop(roots, fan)  # the parity sample of the same roots: a different (both valid) sample
    deg = np.diff(rowptr)
    parents = roots.astype(np.int64)
    differs = False
    for k, f in enumerate(fan):
        nbr = t.nbr[k].cpu().numpy().view(np.uint32).astype(np.int64).reshape(-1, f)
        cnt = t.cnt[k].cpu().numpy()
        assert np.array_equal(t.nbr[k].cpu().numpy(), t2.nbr[k].cpu().numpy())  # reproducible
        differs |= not np.array_equal(t.nbr[k].cpu().numpy(), par.nbr[k].cpu().numpy())
        valid = parents != INV
        want = np.where(valid, np.minimum(deg[np.minimum(parents, n - 1)], f), 0)
        assert np.array_equal(cnt, want)  # exactly min(deg, f)
        for i in np.flatnonzero(valid):
            got = nbr[i][: cnt[i]]
            row = col[rowptr[parents[i]]: rowptr[parents[i] + 1]].astype(np.int64)
            assert np.all(np.diff(got) > 0) and np.all(np.isin(got, row))  # distinct, ascending, in-edges
            assert np.all(nbr[i][cnt[i]:] == INV)
        parents = nbr.reshape(-1)
    assert differs  # (it IS another sampler: never claimed set-equal)
    # a root's subtree depends on the root alone
    t3 = eng.sample_khop(roots[::-1].copy(), fan, mode=MODE_FAST)
    assert np.array_equal(t3.nbr[1].cpu().numpy().reshape(512, -1)[::-1], t.nbr[1].cpu().numpy().reshape(512, -1))


def test_fast_mode_draws_one_uniform_position_per_stratum(setup):
    """the j-th sample of a row of deg > f lies in stratum j = [j*deg/f, (j+1)*deg/f) of the ascending row (so every
    neighbour is drawn with probability ~ f / deg), and over many seeds the position inside the stratum is uniform:
    quarter-of-stratum counts within 5 sigma"""
    from gigl_amd._lib import MODE_FAST
    eng, rowptr, col, n = setup
    deg = np.diff(rowptr)
    hub = int(np.argmax(deg))
    d = int(deg[hub])
    f = 10
    assert d > 400
    row = col[rowptr[hub]: rowptr[hub + 1]].astype(np.int64)
    lo = np.arange(f) * d // f
    hi = (np.arange(f) + 1) * d // f
    quarters = np.zeros(4, dtype=np.int64)
    calls = 400
    root = np.array([hub], dtype=np.uint32)
    seen = set()
    for s in range(calls):
        t = eng.sample_khop(root, [f], sampling_seed=7 + 13 * s, mode=MODE_FAST)
        ids = t.nbr[0].cpu().numpy().view(np.uint32).astype(np.int64)
        pos = np.searchsorted(row, ids)
        assert np.array_equal(row[pos], ids) and np.all((pos >= lo) & (pos < hi))
        quarters += np.bincount(((pos - lo) * 4 // (hi - lo)).clip(0, 3), minlength=4)
        seen.add(tuple(ids.tolist()))
    assert len(seen) > calls // 2  # the seed moves the sample
    total = calls * f
    sigma = np.sqrt(total * 0.25 * 0.75)
    assert np.all(np.abs(quarters - total / 4) < 5 * sigma), quarters


def test_plan_in_fast_mode_equals_the_stepwise_path(setup):
    from gigl_amd._lib import MODE_FAST
    from gigl_amd.models import GraphSAGE, HipBatch
    eng, rowptr, col, n = setup
    torch.manual_seed(2)
    model = GraphSAGE(64, 32, 16, num_layers=2).to(eng.device)
    b, fan = 256, [25, 10]
    plan = model.make_plan(eng, b, fan)
    roots = np.random.default_rng(3).integers(0, n, size=b).astype(np.uint32)
    r_dev = torch.from_numpy(roots.view(np.int32)).to(eng.device)
    out = plan.run(r_dev, mode=MODE_FAST).cpu().numpy()
    hb = plan.last_batch_to_host()
    tree = eng.sample_khop(roots, fan, mode=MODE_FAST)
    for k in range(2):
        assert np.array_equal(hb["nbr"][k], tree.nbr[k].cpu().numpy().view(np.uint32))
    u = eng.union_build(tree)
    ref = model(HipBatch(eng, tree, u))[u.root_local[:b].long()].cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-6)
