"""host issue time vs GPU time of the library training step (products-shaped small graph)
usage: python scripts/micro_train_plan.py [prefetch 0|1]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from gigl_amd.engine import HipEngine, SageTrainPlan
from gigl_amd.models import GraphSAGE
import argparse
eng = HipEngine(0)
args = argparse.Namespace(workload="products")
n, d = bench.build_workload(eng, args)
torch.manual_seed(0)
model = GraphSAGE(d, 256, 47, num_layers=2).to(eng.device)
st = torch.cuda.Stream(device=eng.device)
torch.cuda.synchronize(); eng.bind_stream(st)
B, K = 1024, 300
plan = SageTrainPlan(eng, model, B, [25, 10])
g = torch.Generator(device="cpu"); g.manual_seed(1)
roots = torch.randperm(n, generator=g)[: (K + 8) * B].view(-1, B).to(torch.int32).to(eng.device)
labels = torch.randint(0, 47, ((K + 8), B), generator=g).to(eng.device)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.cuda.synchronize()
with torch.cuda.stream(st):
    for i in range(8):
        plan.step(roots[i], labels[i], next_roots=roots[i + 1] if pre else None)
eng.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(st):
    for i in range(8, 8 + K - 1):
        plan.step(roots[i], labels[i], next_roots=roots[i + 1] if pre else None)
t1 = time.perf_counter()
eng.synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"prefetch={pre}: host issue {1e6 * (t1 - t0) / (K - 1):.1f} us/step, total {1e6 * (t2 - t0) / (K - 1):.1f} us/step")
plan.close(); eng.close()
