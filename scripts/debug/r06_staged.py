"""debug: staged sharded plan (default / worst-case buckets) on an overflowing batch vs the oracle"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch, oracle
from test_gpu_overflow import _graph, FAN, B, N, _overflows
from test_gpu_dist_plan import shard_engine
from gigl_amd.dist import Comm, DistSagePlan
rowptr, col, x = _graph(16)
perm = np.random.default_rng(0).permutation(N)
world = 2
st = torch.cuda.Stream()
engs = [shard_engine(rowptr, col, x, r, world, torch.float32, st) for r in range(world)]
comms = Comm.local(engs)
dev = engs[0].device
L = 2
w = [torch.zeros((4, 2 * (16 if l == 0 else 4)), device=dev) for l in range(L)]
for wide in (False, True):
    kw = dict(hop_slack=float(world), pull_cap=1 << 40) if wide else {}
    plans = [DistSagePlan(comms[r], w, [None] * L, B, list(FAN), staged=True, **kw) for r in range(world)]
    for bi in (0, 1, 2):
        roots = [perm[(2 * bi + r) * B:(2 * bi + r + 1) * B].astype(np.uint32) for r in range(world)]
        rd = [torch.from_numpy(r.view(np.int32)).to(dev) for r in roots]
        with torch.cuda.stream(st):
            DistSagePlan.sample_and_pull_local(plans, rd)
            ts = [p.batch_tensors() for p in plans]
        st.synchronize()
        for r in range(world):
            over, u = _overflows(rowptr, col, roots[r])
            t = ts[r]; m = t["meta"].cpu().numpy()
            n = int(m[0])
            ok_nodes = n == int(u["meta"][0]) and np.array_equal(t["nodes"][:n].cpu().numpy().view(np.uint32), u["nodes"][:n])
            ok_rp = ok_nodes and np.array_equal(t["rowptr"][:n + 1].cpu().numpy(), u["rowptr"][:n + 1])
            ok_col = ok_rp and np.array_equal(t["col"][:int(m[1])].cpu().numpy(), u["col"][:int(m[1])])
            ok_x = ok_nodes and np.array_equal(t["x"][:n].cpu().numpy(), x[u["nodes"][:n].astype(np.int64)])
            print(f"wide={wide} batch {bi} rank {r}: oracle-overflow={over} meta8={m[8]} n={n}/{int(u['meta'][0])} nodes={ok_nodes} rowptr={ok_rp} col={ok_col} x={ok_x}")
    for p in plans:
        p.close()
