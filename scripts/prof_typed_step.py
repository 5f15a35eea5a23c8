import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gigl_amd.graphdb_sampler import INCOMING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG
from gigl_amd.models_hetero import HGT
na, npp, ne, B = 2_000_000, 4_000_000, 40_000_000, 4096
rng = np.random.default_rng(0)
a2p, p2a = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
src = (na * rng.random(ne) ** 2).astype(np.int64); dst = rng.integers(0, npp, ne)
edges = {a2p: (src.astype(np.uint32), dst.astype(np.uint32)), p2a: (dst.astype(np.uint32), src.astype(np.uint32))}
feats = {"author": rng.standard_normal((na, 64)).astype(np.float32), "paper": rng.standard_normal((npp, 128)).astype(np.float32)}
s = HipGraphDBSampler({"author": 0, "paper": 1}, {"author": na, "paper": npp}, edges, {a2p: 0, p2a: 1}, feats)
dag = SamplingOpDAG.from_ops([SamplingOp("h1", a2p, 10, [], INCOMING), SamplingOp("h2", p2a, 5, ["h1"], INCOMING)])
ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
model = HGT({"author": 64, "paper": 128}, {e: 0 for e in ets}, hid_dim=64, out_dim=64, num_layers=2, num_heads=2).cuda().eval()
model.engine = s.engine
roots = rng.integers(0, npp, B)
def step():
    g, ri, _ = s.batch_graph_plan(roots, "paper", dag, b_max=B, edge_type_ids=model.convs[0].edge_types_map)
    with torch.no_grad():
        return model(g, ["paper"], row_subset={"paper": ri})["paper"]
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 10 * 1e3)
def host_only():
    t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / 10 * 1e3
print("host issue ms/step", host_only())
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=60))
s.close()
