"""Several independent batches in one set of launches (gigl_union_build_groups / gigl_sage_plan_set_groups).

Oracle formulation: the grouped union of G batches is the ordinary union of the SAME trees with every node id
of batch g replaced by id + g*n (disjoint id spaces), so oracle.union_build on the tagged tree is the expected
result; on top of that each batch's sub-union must equal its stand-alone union (same node order, same rows),
and the plan's per-root rows must be bit-identical to the single-batch plan's."""
import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges

pytestmark = pytest.mark.gpu
INVALID = 0xFFFFFFFF


@pytest.fixture(scope="module")
def eng():
    from gigl_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _tag(arr, per_group, n):
    """arr laid out group-major with `per_group` entries per batch -> ids + g*n (INVALID kept)"""
    a = arr.astype(np.uint64)
    g = (np.arange(a.size, dtype=np.uint64) // np.uint64(per_group))
    out = np.where(arr == INVALID, np.uint64(INVALID), a + g * np.uint64(n))
    return out.astype(np.uint32)


def _check_groups(eng, rowptr, col, n, roots, fanouts, group_roots):
    hops = len(fanouts)
    b = roots.size
    G = b // group_roots
    tree = eng.sample_khop(roots, fanouts)
    u = eng.union_build(tree, group_roots=group_roots)
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots, fanouts, canonical=True)
    # expected: ordinary union over tagged ids
    per, nbr_t = group_roots, []
    for k in range(hops):
        per *= fanouts[k]
        nbr_t.append(_tag(nbr_o[k], per, n))
    o = oracle.union_build(_tag(roots, group_roots, n), fanouts, nbr_t)
    assert np.array_equal(u.meta.cpu().numpy()[: 3 + hops], o["meta"][: 3 + hops])
    assert int(u.meta.cpu().numpy()[8]) == 0
    nodes_h, rp_h, col_h = u.to_csr()
    assert np.array_equal(nodes_h, o["nodes"] % np.uint32(n))
    assert np.array_equal(rp_h, o["rowptr"]) and np.array_equal(col_h, o["col"])
    assert np.array_equal(u.root_local.cpu().numpy()[:b], o["root_local"])
    # each batch's sub-union == its stand-alone union (node order and rows)
    grp = (o["nodes"].astype(np.int64) // n)
    for g in range(G):
        r_g = roots[g * group_roots:(g + 1) * group_roots]
        per, nbr_g = group_roots, []
        for k in range(hops):
            per *= fanouts[k]
            nbr_g.append(nbr_o[k][g * per:(g + 1) * per])
        og = oracle.union_build(r_g, fanouts, nbr_g)
        mine = np.nonzero(grp == g)[0]
        assert np.array_equal(nodes_h[mine], og["nodes"])
        relabel = -np.ones(nodes_h.size, dtype=np.int64)
        relabel[mine] = np.arange(mine.size)
        for i_local, i_mega in enumerate(mine[: og["rowptr"].size - 1]):
            row = col_h[rp_h[i_mega]:rp_h[i_mega + 1]]
            want = og["col"][og["rowptr"][i_local]:og["rowptr"][i_local + 1]]
            assert np.array_equal(relabel[row], want), (g, i_local)
    return o


def test_grouped_union_overlapping_batches(eng):
    """small graph so that the batches overlap heavily (same global ids in several batches, duplicate roots)"""
    s, d = rmat_edges(10, 12000, seed=3)
    n = 1 << 10
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    eng.load_csc(rowptr, col)
    rng = np.random.default_rng(5)
    roots = rng.integers(0, n, size=4 * 96).astype(np.uint32)
    roots[96:192] = roots[:96]  # batch 1 == batch 0: identical sub-unions, disjoint local ids
    roots[200] = roots[201]
    _check_groups(eng, rowptr, col, n, roots, [5, 4], 96)
    _check_groups(eng, rowptr, col, n, roots[:192], [3, 2, 2], 64)  # 3 hops (level relaxation), 3 groups
    _check_groups(eng, rowptr, col, n, roots[:7], [4, 3], 1)        # one root per batch
    tree = eng.sample_khop(roots[:7], [4, 3])
    with pytest.raises(RuntimeError):
        eng.union_build(tree, group_roots=5)  # does not divide b = 7


def test_grouped_union_big_rows(eng):
    """hub rows (> 64 sampled in-edges) in every batch: the LDS sort with clustered local ids"""
    n, b = 6000, 4 * 128
    hub_src = np.arange(1000, 6000, dtype=np.uint32)
    src = np.concatenate([hub_src, np.zeros(600, dtype=np.uint32)])
    dst = np.concatenate([np.zeros(5000, dtype=np.uint32), np.arange(1, 601, dtype=np.uint32)])
    rowptr, col = oracle.build_csc(n, src, dst, is_directed=True)
    eng.load_csc(rowptr, col)
    roots = (np.arange(b, dtype=np.uint32) % 600) + 1
    o = _check_groups(eng, rowptr, col, n, roots, [2, 10], 128)
    hubs = np.nonzero(o["nodes"] % np.uint32(n) == 0)[0]
    assert hubs.size == 4  # one local hub per batch
    assert all(o["rowptr"][h + 1] - o["rowptr"][h] > 64 for h in hubs)


def test_grouped_plan_rows_bit_identical(eng):
    from gigl_amd.models import GraphSAGE
    s, d = rmat_edges(13, 160000, seed=18)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = (np.random.default_rng(0).standard_normal((n, 64)) / 8).astype(np.float32)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    torch.manual_seed(2)
    model = GraphSAGE(64, 48, 20, num_layers=2).to(eng.device)
    b, fan, G = 256, [25, 10], 4
    roots = torch.from_numpy(np.random.default_rng(1).integers(0, n, size=(3, G * b)).astype(np.int32)).to(eng.device)
    single = model.make_plan(eng, b, fan)
    grouped = model.make_plan(eng, b, fan, groups=G)
    assert grouped.b == G * b
    for i in range(3):
        want = torch.cat([single.run(roots[i, g * b:(g + 1) * b].contiguous()).clone() for g in range(G)])
        got = grouped.run(roots[i])
        assert torch.equal(got, want), i
    hb = grouped.last_batch_to_host()
    assert hb["meta"][8] == 0 and hb["meta"][2] <= G * b
    # hipGraph replay of the grouped plan
    st = torch.cuda.Stream(device=eng.device)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    torch.cuda.set_stream(st)
    try:
        grouped.use_graph(True)
        want = [grouped.run(roots[i]).clone() for i in range(3)]  # call 0 captures eagerly
        for i in range(3):
            assert torch.equal(grouped.run(roots[i]), want[i])
        ref = torch.cat([single.run(roots[2, g * b:(g + 1) * b].contiguous()).clone() for g in range(G)])
        assert torch.equal(want[2], ref)
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream(eng.device))
        eng.bind_stream(None)


def test_graph_part_on_a_stream_of_its_own_gives_the_same_rows(eng):
    """gigl_sage_plan_set_graph_stream: sample + union of every call on a high-priority stream, the layers behind an event on
    the engine's stream — the same launches in the same order within a call: rows bit-identical to the one-stream plan over
    many calls back to back (each call's graph part also waits for the previous call's layers, which read the union it
    rewrites), with and without kernel timers (timed stages run eagerly between the captured segments)"""
    from gigl_amd.models import GraphSAGE
    s, d = rmat_edges(13, 160000, seed=19)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = (np.random.default_rng(4).standard_normal((n, 64)) / 8).astype(np.float32)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    torch.manual_seed(3)
    model = GraphSAGE(64, 48, 20, num_layers=2).to(eng.device)
    b, fan, G, calls = 256, [25, 10], 4, 12
    roots = torch.from_numpy(np.random.default_rng(2).integers(0, n, size=(calls, G * b)).astype(np.int32)).to(eng.device)
    plan = model.make_plan(eng, b, fan, groups=G)
    want = [plan.run(roots[i]).clone() for i in range(calls)]
    st, hi = torch.cuda.Stream(device=eng.device), torch.cuda.Stream(device=eng.device, priority=-1)
    torch.cuda.synchronize()
    eng.bind_stream(st)
    torch.cuda.set_stream(st)
    try:
        plan.use_graph(True)
        plan.set_graph_stream(hi)
        outs = [torch.empty_like(want[0]) for _ in range(calls)]
        for rep in range(2):
            for i in range(calls):  # (no synchronisation between the calls)
                plan.run(roots[i], out=outs[i])
            st.synchronize()
            for i in range(calls):
                assert torch.equal(outs[i], want[i]), (rep, i)
        eng.profile_enable(["union_insert", "gather_mean"], capacity=256)
        for i in range(calls):
            plan.run(roots[i], out=outs[i])
        st.synchronize()
        assert all(torch.equal(outs[i], want[i]) for i in range(calls))
        assert eng.profile_read("union_insert")[1] > 0 and eng.profile_read("gather_mean")[1] > 0
        eng.profile_enable([], 0)
        plan.set_graph_stream(None)
        assert torch.equal(plan.run(roots[0]), want[0])
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream(eng.device))
        eng.bind_stream(None)
