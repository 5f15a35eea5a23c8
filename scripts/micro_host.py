"""how long does the HOST take to enqueue one step? (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gigl_amd.engine import HipEngine
from gigl_amd.models import GraphSAGE, HipBatch
eng = HipEngine(0); dev = eng.device
n = 200000
src = torch.randint(0, n, (3_000_000,), dtype=torch.int32, device=dev); dst = torch.randint(0, n, (3_000_000,), dtype=torch.int32, device=dev)
eng.build_from_coo(n, src, dst, False)
eng.load_features(torch.randn(n, 100, device=dev))
model = GraphSAGE(100, 256, 47).to(dev)
B = 1024; fan = [25, 10]
roots = torch.randint(0, n, (64, B), dtype=torch.int32, device=dev)
tree = eng.alloc_tree(B, fan); union = eng.alloc_union(B, fan)
def step(i):
    t = eng.sample_khop(roots[i % 64], fan, out=tree); u = eng.union_build(t, out=union); return model(HipBatch(eng, t, u))
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/200:.1f} us/step; total {1e6*(t2-t0)/200:.1f} us/step")
