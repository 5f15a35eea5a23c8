import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from gigl_amd.engine import HipEngine
class A: small=False; workload="products"
eng = HipEngine(0)
n, d = bench.build_workload(eng, A)
g = torch.Generator().manual_seed(42)
perm = torch.randperm(n, generator=g)
for b0 in range(3):
    roots = perm[b0*1024:(b0+1)*1024].to(torch.int32).cuda()
    tree = eng.sample_khop(roots, [25, 10]); u = eng.union_build(tree)
    nn = int(u.meta[0]); ln = (u.rowend[:nn] - u.rowptr[:nn]).cpu()
    lv1 = int(u.meta[3])
    l1 = ln[:lv1]
    print("batch", b0, "rows", lv1, "edges", int(l1.sum()), "max", int(l1.max()), "top5", sorted(l1.tolist())[-5:], ">=128:", int((l1>=128).sum()), "edges in >=128 rows", int(l1[l1>=128].sum()))
eng.close()
