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
    finally:
        eng.close()
