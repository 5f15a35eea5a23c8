#!/usr/bin/env python3
"""GAT forward over sampled batches on the bench workload (products-shaped graph, fanout [25,10], B roots):
2-layer GAT 100 -> 4 heads x 64 -> 256, per-kernel HIP-event timing of the aggregation and the projection."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gigl_amd.engine import HipEngine  # noqa: E402
from gigl_amd.models import HipBatch  # noqa: E402
from gigl_amd.models_attn import GAT  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--small", action="store_true")
ap.add_argument("--heads", type=int, default=4)
ap.add_argument("--hid", type=int, default=64)
ap.add_argument("--edge-dim", type=int, default=0)
a = ap.parse_args()
a.workload = "products"
eng = HipEngine(0)
n, d = bench.build_workload(eng, a)
torch.manual_seed(0)
model = GAT(d, a.hid, 256, num_layers=2, heads=a.heads, edge_dim=a.edge_dim or None,
            conv="edge_attr_gat" if a.edge_dim else "gat").to(eng.device)
if a.edge_dim:
    eng._set_edge_table(torch.randn(eng.n_edges, a.edge_dim, device=eng.device))
g = torch.Generator().manual_seed(42)
fan = [25, 10]
roots = torch.randperm(n, generator=g)[: a.batch].to(torch.int32).cuda()
tree = eng.sample_khop(roots, fan)
u = eng.union_build(tree)
batch = HipBatch(eng, tree, u)
if a.edge_dim:
    batch.edge_attr = eng.union_edge_attr(u)
out = model(batch)
torch.cuda.synchronize()
meta = u.meta.cpu().tolist()
rp, re_ = u.rowptr.cpu(), u.rowend.cpu()
lv = [meta[2], meta[3], meta[4]]
agg = int((re_[: lv[1]] - rp[: lv[1]]).sum() + (re_[: lv[0]] - rp[: lv[0]]).sum())
eng.profile_enable(["gather_mean", "linear"], capacity=4096)
t0 = time.perf_counter()
for _ in range(a.iters):
    out = model(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
g_ms, g_n = eng.profile_read("gather_mean")
l_ms, l_n = eng.profile_read("linear")
print(f"GAT forward B={a.batch} heads={a.heads} hid={a.hid} edge_dim={a.edge_dim}: {dt*1e3:.3f} ms/batch wall; "
      f"aggregation {g_ms/a.iters:.3f} ms ({g_n//a.iters} launches), projection {l_ms/a.iters:.3f} ms; "
      f"{agg} aggregated edges -> {agg/(g_ms/a.iters)*1e-6:.2f} G edges/s in the aggregation kernels; "
      f"nodes per level {lv}")
eng.close()
