"""debug: dense sharded plan with G groups on the overflow graph vs the oracle forward, per group"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch, oracle
from oracle import gnn_ref
from test_gpu_overflow import _graph, FAN, B, N, _overflows
from test_gpu_dist_plan import shard_engine
from gigl_amd.dist import Comm, DistSagePlan
from gigl_amd.models import GraphSAGE
rowptr, col, x = _graph(16)
perm = np.random.default_rng(0).permutation(N)
world, G = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 16
st = torch.cuda.Stream()
engs = [shard_engine(rowptr, col, x, r, world, torch.float32, st) for r in range(world)]
comms = Comm.local(engs)
dev = engs[0].device
torch.manual_seed(5)
model = GraphSAGE(16, 32, 8, num_layers=2)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
w, bs = model.fused_params()
plans = [DistSagePlan(comms[r], w, bs, G * B, list(FAN), group_roots=B) for r in range(world)]
roots = [perm[r * G * B:(r + 1) * G * B].astype(np.uint32) for r in range(world)]
rd = [torch.from_numpy(r.view(np.int32)).to(dev) for r in roots]
with torch.cuda.stream(st):
    outs = DistSagePlan.run_local(plans, rd)
st.synchronize()
for r in range(world):
    print("rank", r, "overflowed flag:", plans[r].overflowed(), "meta", plans[r].buffers_to_host()["meta"][:10])
    o = outs[r].cpu().numpy()
    for g in range(G):
        rs = roots[r][g * B:(g + 1) * B]
        over, u = _overflows(rowptr, col, rs)
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        want = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, sd, 2)[u["root_local"]].numpy()
        got = o[g * B:(g + 1) * B]
        err = np.abs(got - want).max() if np.isfinite(got).all() else float("nan")
        print(f"  group {g}: oracle-overflow={over} level<=1 nodes={int(u['meta'][3])} (cap {B*(1+FAN[0])}) max err {err:.3g}")
