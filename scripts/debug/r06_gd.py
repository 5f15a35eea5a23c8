"""debug: ResidentGraph.graph_data over a staged sharded plan + GraphSAGE(GraphData) vs the oracle forward"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch, oracle
from oracle import gnn_ref
from test_gpu_overflow import _graph, FAN, B, N, _overflows
from test_gpu_dist_plan import shard_engine
from gigl_amd.dist import Comm, DistSagePlan
from gigl_amd.hbm import ResidentGraph
from gigl_amd.models import GraphSAGE
rowptr, col, x = _graph(16)
perm = np.random.default_rng(0).permutation(N)
world = 2
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
engs = [shard_engine(rowptr, col, x, r, world, torch.float32, st) for r in range(world)]
comms = Comm.local(engs)
dev = engs[0].device
torch.manual_seed(5)
model = GraphSAGE(16, 32, 8, num_layers=2).to(dev)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
L = 2
w = [torch.zeros((4, 2 * (16 if l == 0 else 4)), device=dev) for l in range(L)]
plans = [DistSagePlan(comms[r], w, [None] * L, B, list(FAN), staged=True, hop_slack=2.0, pull_cap=1 << 40) for r in range(world)]
for bi in range(3):
    roots = [perm[(2 * bi + r) * B:(2 * bi + r + 1) * B].astype(np.uint32) for r in range(world)]
    rd = [torch.from_numpy(r.view(np.int32)).to(dev) for r in roots]
    DistSagePlan.sample_and_pull_local(plans, rd)
    for r in range(world):
        res = object.__new__(ResidentGraph)
        res.sharded, res.engine, res.device, res.seed, res.world, res.fanouts = True, engs[r], dev, 42, 1, list(FAN)
        pl = plans[r]
        pl_sp = pl.sample_and_pull
        pl.sample_and_pull = lambda roots, sampling_seed=42: None
        res._staged_plan = lambda b, wide=False, pl=pl: pl
        gd, ri = res.graph_data(rd[r], pad_to=B, wide=True)
        pl.sample_and_pull = pl_sp
        model.engine = engs[r]
        with torch.no_grad():
            got = model(gd)[ri].cpu().numpy()
        over, u = _overflows(rowptr, col, roots[r])
        ei = gnn_ref.union_edge_index(u["rowptr"], u["col"])
        want = gnn_ref.graphsage_forward(torch.from_numpy(x[u["nodes"].astype(np.int64)]), ei, sd, 2)[u["root_local"]].numpy()
        print(f"batch {bi} rank {r}: overflow={over} root_local ok={np.array_equal(ri.cpu().numpy(), u['root_local'])} max err {np.abs(got-want).max():.3g}")
