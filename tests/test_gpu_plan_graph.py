"""hipGraph replay of the batch pipeline: identical results to eager launches, with and without timing events."""
import numpy as np
import pytest
import torch

import oracle
from helpers import rmat_edges

pytestmark = pytest.mark.gpu


def test_graph_replay_equals_eager():
    from gigl_amd.engine import HipEngine
    from gigl_amd.models import GraphSAGE
    s, d = rmat_edges(13, 160000, seed=18)
    n = 1 << 13
    rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
    x = (np.random.default_rng(0).standard_normal((n, 64)) / 8).astype(np.float32)
    eng = HipEngine(0)
    eng.load_csc(rowptr, col)
    eng.load_features(x)
    torch.manual_seed(2)
    model = GraphSAGE(64, 48, 20, num_layers=2).to(eng.device)
    b, fan = 256, [25, 10]
    roots = torch.from_numpy(np.random.default_rng(1).integers(0, n, size=(12, b)).astype(np.int32)).to(eng.device)
    eager = model.make_plan(eng, b, fan)
    want = [eager.run(roots[i]).clone() for i in range(12)]
    g = model.make_plan(eng, b, fan)
    with pytest.raises(RuntimeError):
        g.use_graph(True)  # the legacy default stream cannot be captured: loud error, no silent eager fallback
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=eng.device)
    eng.bind_stream(st)
    torch.cuda.set_stream(st)
    g.use_graph(True)
    got = [g.run(roots[i]).clone() for i in range(12)]  # call 0 captures (eager run), 1.. replay the ring
    for i in range(12):
        assert torch.equal(got[i], want[i]), i
    # with timing events inside the graph
    eng.profile_enable(["expand", "linear"], 64)
    got2 = [g.run(roots[i]).clone() for i in range(12)]
    g.flush_profile()
    for i in range(12):
        assert torch.equal(got2[i], want[i]), i
    ms, nl = eng.profile_read("expand")
    ms2, nl2 = eng.profile_read("linear")
    assert nl == 2 * 11 and nl2 == 2 * 11 and ms > 0 and ms2 > 0  # 11 replays x 2 launches (re-capture run is untimed)
    eng.profile_enable([], 0)
    # different seed -> re-capture, still correct
    o1 = g.run(roots[0], sampling_seed=7).clone()
    o2 = eager.run(roots[0], sampling_seed=7).clone()
    assert torch.equal(o1, o2)
    hb = g.last_batch_to_host()
    nbr_o, _ = oracle.sample_khop(rowptr, col, roots[0].cpu().numpy().view(np.uint32), fan, sampling_seed=7, canonical=True)
    assert np.array_equal(hb["nbr"][1], nbr_o[1])
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream(eng.device))
    eng.close()
