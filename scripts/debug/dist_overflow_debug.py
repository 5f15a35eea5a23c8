"""per rank: meta / stats / row error of the sharded plan in one configuration (debug aid)"""
import sys
sys.path[:0] = ["tests", "."]
import numpy as np
import torch
import test_gpu_dist_plan as T
from gigl_amd.dist import Comm, DistSagePlan

world, fan = int(sys.argv[1]), eval(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "raw"
T.FAN = fan
rowptr, col, x = T.make_graph()
model = T.make_model()
w, bs = model.fused_params()
b, gr = 96, 32
st = torch.cuda.Stream()
engs = [T.shard_engine(rowptr, col, x, r, world, torch.float32, st) for r in range(world)]
comms = Comm.local(engs)
kw = {}
if mode == "pre":
    wdev = w[0].to(engs[0].device)
    tables = [e.project_features(wdev) for e in engs]
plans = [DistSagePlan(comms[r], w, bs, b, fan, group_roots=gr, max_window_end=T.bound_for(rowptr),
                      **({"projected": tables[r]} if mode == "pre" else {"project_on_owner": mode == "owner"})) for r in range(world)]
roots = [T.rank_roots(r, b) for r in range(world)]
roots_d = [torch.from_numpy(r.view(np.int32)).to(engs[0].device) for r in roots]
for _ in range(2):
    outs = DistSagePlan.run_local(plans, roots_d)
st.synchronize()
for r in range(world):
    acc = torch.zeros(16, dtype=torch.int64, device=engs[0].device)
    with torch.cuda.stream(st):
        plans[r].stats(acc)
    st.synchronize()
    hb = plans[r].buffers_to_host()
    want = T.reference_rows(rowptr, col, x, model, roots[r], gr)
    got = outs[r].cpu().numpy()
    bad = np.flatnonzero(~np.isclose(got, want, rtol=1e-5, atol=1e-5).all(axis=1))
    print(r, "meta", hb["meta"][:9].tolist(), "stats", acc.cpu().numpy()[[0, 3, 13, 14, 15]].tolist(), "bad rows", bad[:12].tolist(), len(bad), flush=True)
