"""Size-independent properties at the FULL per-GPU sizes of BASELINE.json's sharded configs, as
tests/test_gpu_fullsize.py does for configs[1]:
  rmat-shard  configs[3] / 8: 2^27 nodes, 2e9 directed edges (hubs of > 10^5 in-edges), D = 128 fp16, fanout [15, 10],
              B = 4096
  mag-shard   configs[2] / 8: 30.5 M nodes, 216 M directed edges, D = 768 fp16, fanout [25, 10], B = 1024 — also the
              projected-input first layer (the mode bench.py runs this workload in) against the unprojected one
Checked on the device (the CPU oracle would need minutes per batch at these sizes): every parent gets exactly
min(deg, f) neighbours, ascending, duplicate-free, all of them in-edges of the resident CSC; repeated calls are
identical; a root's subtree depends on the root alone; 200 sampled rows — hubs included — equal the oracle's hash
permutation of the row; the one-call plan equals the step-by-step entry points; on mag-shard also configs[4]'s GAT
encoder: its one-call plan equals the staged forward."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INV = 0xFFFFFFFF


def _device_csc(eng):
    """the resident CSC as device tensors (copied device -> device through the C ABI)"""
    from gigl_amd._lib import LOC_DEVICE, check
    n, e = C.c_int64(), C.c_int64()
    check(eng._lib.gigl_graph_info(eng._graph, C.byref(n), C.byref(e)), eng._ctx)
    rp, cl = C.c_void_p(), C.c_void_p()
    check(eng._lib.gigl_graph_device_ptrs(eng._graph, C.byref(rp), C.byref(cl)), eng._ctx)
    rowptr = torch.empty(n.value + 1, dtype=torch.int64, device=eng.device)
    col = torch.empty(e.value, dtype=torch.int32, device=eng.device)
    check(eng._lib.gigl_memcpy(eng._ctx, C.c_void_p(rowptr.data_ptr()), LOC_DEVICE, rp, LOC_DEVICE, rowptr.numel() * 8), eng._ctx)
    check(eng._lib.gigl_memcpy(eng._ctx, C.c_void_p(col.data_ptr()), LOC_DEVICE, cl, LOC_DEVICE, col.numel() * 4), eng._ctx)
    eng.synchronize()
    return rowptr, col


@pytest.mark.parametrize("workload,fan,b,hid", [("rmat-shard", [15, 10], 4096, 256), ("mag-shard", [25, 10], 1024, 256)])
def test_full_size_shard_invariants(workload, fan, b, hid):
    sys.path.insert(0, ROOT)
    import bench
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE, HipBatch

    class A:
        small = False
    A.workload = workload
    eng = HipEngine(0)
    try:
        n, d = bench.build_workload(eng, A)
        dev = eng.device
        rowptr, col = _device_csc(eng)
        deg = rowptr[1:] - rowptr[:-1]
        assert int(deg.max()) > 50_000  # hubs: the heavy-row paths of the sampler are exercised
        g = torch.Generator().manual_seed(11)
        roots = torch.randint(0, n, (2 * b,), generator=g).to(torch.int32).to(dev)
        tree = eng.sample_khop(roots[:b], fan)
        again = eng.sample_khop(roots[:b], fan)
        parents = roots[:b].to(torch.int64) & INV
        f0, f1 = fan
        for k, f in enumerate(fan):
            nbr = tree.nbr[k].to(torch.int64) & INV
            cnt = tree.cnt[k].to(torch.int64)
            assert torch.equal(tree.nbr[k], again.nbr[k]) and torch.equal(tree.cnt[k], again.cnt[k])
            pvalid = parents != INV
            pc = parents.clamp(max=n - 1)
            want = torch.where(pvalid, torch.minimum(deg[pc], torch.tensor(f, device=dev)), torch.zeros_like(parents))
            assert torch.equal(cnt, want)
            m = nbr.view(-1, f)
            filled = torch.arange(f, device=dev).view(1, -1) < cnt.view(-1, 1)
            assert bool(((m != INV) == filled).all())
            assert bool(((m[:, 1:] > m[:, :-1]) | ~filled[:, 1:]).all())  # ascending, duplicate-free
            # every sampled id is an in-neighbour of its parent: binary search inside the parent's row of the CSC
            lo = rowptr[pc].view(-1, 1).expand(-1, f).clone()
            hi = rowptr[pc + 1].view(-1, 1).expand(-1, f).clone()
            end = hi.clone()
            key = m.to(torch.int64)
            for _ in range(34):
                active = lo < hi
                mid = (lo + hi) >> 1
                v = (col[mid.clamp(max=col.numel() - 1)].to(torch.int64) & INV)
                right = active & (v < key)
                left = active & ~(v < key)
                lo = torch.where(right, mid + 1, lo)
                hi = torch.where(left, mid, hi)
            hit = (lo < end) & ((col[lo.clamp(max=col.numel() - 1)].to(torch.int64) & INV) == key)
            assert bool((hit | ~filled).all())
            parents = nbr
        # a root's subtree is a function of the root alone
        t2 = eng.sample_khop(torch.cat([roots[b // 2:b], roots[b:b + b // 2]]).contiguous(), fan)
        assert torch.equal(t2.nbr[0].view(b, f0)[: b // 2], tree.nbr[0].view(b, f0)[b // 2:])
        assert torch.equal(t2.nbr[1].view(b, f0 * f1)[: b // 2], tree.nbr[1].view(b, f0 * f1)[b // 2:])
        # oracle spot check: 200 hop-2 rows (the longest ones included) against the restated hash permutation
        par = (tree.nbr[0].to(torch.int64) & INV)
        dpar = torch.where(par != INV, deg[par.clamp(max=n - 1)], torch.zeros_like(par))
        cand = torch.nonzero(dpar > f1).view(-1)
        top = cand[torch.topk(dpar[cand], min(20, cand.numel())).indices]
        rest = cand[torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(3))[:180].to(dev)]
        pick = torch.unique(torch.cat([top, rest])).cpu().numpy()
        par_h = par.cpu().numpy()
        rts = np.repeat((roots[:b].to(torch.int64) & INV).cpu().numpy(), f0)
        got = (tree.nbr[1].to(torch.int64) & INV).view(-1, f1).cpu().numpy()
        for i in pick:
            p = int(par_h[i])
            row = (col[int(rowptr[p]): int(rowptr[p + 1])].to(torch.int64) & INV).cpu().numpy().astype(np.uint32)
            want = np.sort(oracle.hash_permutation(row, (int(rts[i]) + p) & INV, sampling_seed=84, counter=1)[:f1])
            assert np.array_equal(got[i], want), (i, p, row.size)
        del rowptr, col, deg
        torch.cuda.empty_cache()
        # the one-call plan == the step-by-step entry points (2e-6), projected input == unprojected (1e-5)
        torch.manual_seed(0)
        model = GraphSAGE(d, hid, hid, num_layers=2).to(dev)
        plan = model.make_plan(eng, b // 2, fan, groups=2)
        out = plan.run(roots[:b].contiguous()).clone()
        assert plan.last_batch_to_host()["meta"][8] == 0
        for gi in range(2):
            r = roots[gi * (b // 2):(gi + 1) * (b // 2)].contiguous()
            tr = eng.sample_khop(r, fan)
            u = eng.union_build(tr)
            ref = model(HipBatch(eng, tr, u))[u.root_local[: b // 2].long()]
            np.testing.assert_allclose(out[gi * (b // 2):(gi + 1) * (b // 2)].cpu().numpy(), ref.cpu().numpy(), rtol=2e-6,
                                       atol=2e-6)
        if workload == "mag-shard":
            assert model.projected_input_pays(eng)
            proj = eng.project_features(model.conv_layers[0].fused_weight())
            plan.set_projected_input(proj)
            out_p = plan.run(roots[:b].contiguous())
            assert plan.last_batch_to_host()["meta"][8] == 0
            np.testing.assert_allclose(out_p.cpu().numpy(), out.cpu().numpy(), rtol=1e-5, atol=1e-5)
            # ... and one batch of it against the CPU restatement END TO END (oracle sample -> collate -> fp32 forward over
            # the whole union graph, homogeneous.py:107-153) at 1e-5: the projected-input plan as bench.py runs mag-shard
            from oracle import gnn_ref
            rowptr_h, col_h = eng.graph_to_host()
            r_h = roots[: b // 2].cpu().numpy().view(np.uint32)
            nbr_o, _ = oracle.sample_khop(rowptr_h, col_h, r_h, fan, canonical=True)
            o = oracle.union_build(r_h, fan, nbr_o)
            del rowptr_h, col_h
            ids = torch.from_numpy(o["nodes"].astype(np.int64)).to(torch.int32).to(dev)
            xs = eng.gather_rows(ids, torch.tensor([ids.numel()], dtype=torch.int32, device=dev), int(ids.numel())).cpu()
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            want = gnn_ref.graphsage_forward(xs, gnn_ref.union_edge_index(o["rowptr"], o["col"]), sd, 2)[o["root_local"]].numpy()
            got_b = out_p[: b // 2].cpu().numpy()
            print(f"mag-shard full size: max |err| vs CPU forward = {np.abs(got_b - want).max():.3e}, max |row| = {np.abs(want).max():.3e}")
            np.testing.assert_allclose(got_b, want, rtol=1e-5, atol=1e-5)
            # configs[4]'s encoder on the same shard (2-layer GAT, heads 2, 768 -> 128 -> 128): the GAT one-call plan
            # (first layer from the input side in one row pass) == the staged forward over the same roots
            from gigl_amd.models_attn import GAT
            plan.close()
            torch.manual_seed(1)
            gat = GAT(d, 128, 128, num_layers=2, heads=2).to(dev)
            gplan = gat.make_plan(eng, b // 2, fan, groups=2)
            got_g = gplan.run(roots[:b].contiguous()).clone()
            for gi in range(2):
                r = roots[gi * (b // 2):(gi + 1) * (b // 2)].contiguous()
                tr = eng.sample_khop(r, fan)
                u = eng.union_build(tr)
                ref = gat(HipBatch(eng, tr, u))[u.root_local[: b // 2].long()]
                np.testing.assert_allclose(got_g[gi * (b // 2):(gi + 1) * (b // 2)].cpu().numpy(), ref.detach().cpu().numpy(),
                                           rtol=1e-5, atol=1e-5)
            assert bool(torch.isfinite(got_g).all())
            gplan.close()
    finally:
        eng.close()
