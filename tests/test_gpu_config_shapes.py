"""The one-call plan at the SHAPES of BASELINE.json's other configs against the oracle (sample -> collate -> fp32 forward
over the whole union graph), on graphs the oracle finishes in seconds:
  configs[3] (RMAT scale-30): directed graph with hubs, fanout [15, 10], B = 4096, D = 128 fp16 features,
             GraphSAGE 128 -> 256 -> 256 — trees bit-identical, root embeddings to 1e-5;
  configs[2] (MAG240M): directed, D = 768 fp16, fanout [25, 10], GraphSAGE 768 -> 256 -> 256."""
import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges
from oracle import gnn_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scale,n_edges,d,hid,out,fan,b", [(15, 400000, 128, 256, 256, [15, 10], 4096),
                                                           (13, 120000, 768, 256, 256, [25, 10], 512)])
def test_plan_matches_oracle_at_config_shapes(scale, n_edges, d, hid, out, fan, b):
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    n = 1 << scale
    s, dst = rmat_edges(scale, n_edges, seed=scale)
    rowptr, col = oracle.build_csc(n, s, dst, is_directed=True)
    assert np.diff(rowptr).max() > 1000  # hubs
    rng = np.random.default_rng(scale)
    x16 = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float16)
    eng = HipEngine(0)
    try:
        eng.load_csc(rowptr, col)
        eng.load_features(torch.from_numpy(x16))
        torch.manual_seed(1)
        model = GraphSAGE(d, hid, out, num_layers=2).to(eng.device)
        roots = rng.integers(0, n, size=b).astype(np.uint32)
        plan = model.make_plan(eng, b, fan)
        got = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
        hb = plan.last_batch_to_host()
        assert hb["meta"][8] == 0
        nbr_o, cnt_o = oracle.sample_khop(rowptr, col, roots, fan, canonical=True)
        for k in range(2):  # the sampled trees: bit-identical
            assert np.array_equal(hb["nbr"][k], nbr_o[k]) and np.array_equal(hb["cnt"][k], cnt_o[k])
        u = oracle.union_build(roots, fan, nbr_o)
        xs = torch.from_numpy(x16[u["nodes"].astype(np.int64)].astype(np.float32))
        sd = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
        ref = gnn_ref.graphsage_forward(xs, gnn_ref.union_edge_index(u["rowptr"], u["col"]), sd, 2)
        want = ref[torch.from_numpy(u["root_local"].astype(np.int64))].numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
        # projected input (gigl_sage_plan_set_projected_input): X W_l^T / X W_r^T computed once over the table, the
        # first layer is a reduction over projected rows — same trees, the same embeddings to 1e-5
        proj = eng.project_features(model.conv_layers[0].fused_weight())
        xw = torch.from_numpy(x16.astype(np.float32)) @ model.conv_layers[0].lin_l.weight.detach().cpu().T
        np.testing.assert_allclose(proj[:, :hid].cpu().numpy(), xw.numpy(), rtol=1e-5, atol=2e-6)
        xr = torch.from_numpy(x16.astype(np.float32)) @ model.conv_layers[0].lin_r.weight.detach().cpu().T
        np.testing.assert_allclose(proj[:, hid:].cpu().numpy(), xr.numpy(), rtol=1e-5, atol=2e-6)
        plan.set_projected_input(proj)
        got_p = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
        hb2 = plan.last_batch_to_host()
        assert hb2["meta"][8] == 0 and all(np.array_equal(hb2["nbr"][k], nbr_o[k]) for k in range(2))
        np.testing.assert_allclose(got_p, want, rtol=1e-5, atol=1e-5)
        plan.set_projected_input(None)  # and back: bit-identical to the first run
        again = plan.run(torch.from_numpy(roots.view(np.int32)).to(eng.device)).cpu().numpy()
        assert np.array_equal(again, got)
        # groups of batches + hipGraph replay with the projected layer (replay needs a created stream)
        g = 4
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        eng.bind_stream(st)
        plan_g = model.make_plan(eng, b // g, fan, groups=g)
        plan_g.set_projected_input(proj)
        r_dev = torch.from_numpy(roots.view(np.int32)).to(eng.device)
        eager = plan_g.run(r_dev)
        st.synchronize()
        eager = eager.clone()
        plan_g.use_graph(True)
        for _ in range(2):
            got_g = plan_g.run(r_dev)
            st.synchronize()
            assert torch.equal(got_g, eager)
        single = model.make_plan(eng, b // g, fan)
        single.set_projected_input(proj)
        one = single.run(r_dev[: b // g].contiguous())
        st.synchronize()
        assert torch.equal(one, eager[: b // g])
    finally:
        eng.close()
