#!/usr/bin/env python3
"""degree mix of the sampler's frontier rows on the bench workload + time per hop (GPU box)"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gigl_amd.engine import HipEngine
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="products")
ap.add_argument("--small", action="store_true")
ap.add_argument("--batch", type=int, default=16384)
a = ap.parse_args()
eng = HipEngine(0)
n, d = bench.build_workload(eng, a)
rowptr, col = eng.graph_to_host()
deg = np.diff(rowptr)
g = torch.Generator().manual_seed(42)
roots = torch.randperm(n, generator=g)[: a.batch].to(torch.int32).cuda()
fan = [25, 10]
tree = eng.sample_khop(roots, fan)
for _ in range(3):
    eng.sample_khop(roots, fan, out=tree)
eng.profile_enable(["expand"], 64)
for _ in range(10):
    eng.sample_khop(roots, fan, out=tree)
ms, nl = eng.profile_read("expand")
eng.profile_enable([], 0)
print(f"expand: {ms/nl*1e3*2:.1f} us per sample_khop call of {a.batch} roots (both hops)")
t1 = eng.alloc_tree(a.batch, fan[:1])
eng.sample_khop(roots, fan[:1], out=t1)
eng.profile_enable(["expand"], 64)
for _ in range(10):
    eng.sample_khop(roots, fan[:1], out=t1)
ms1, nl1 = eng.profile_read("expand")
eng.profile_enable([], 0)
print(f"expand hop 1 alone: {ms1/nl1*1e3:.1f} us")
par = [roots.cpu().numpy().view(np.uint32), tree.nbr[0].cpu().numpy().view(np.uint32)]
edges = [0, 256, 512, 1024, 4096, 16384, 65536, 1 << 30]
for k, p in enumerate(par):
    valid = p != 0xFFFFFFFF
    dg = deg[p[valid]]
    f = fan[k]
    print(f"hop {k+1}: slots {p.size} valid {valid.sum()} copy-through(deg<=f) {(dg<=f).sum()} "
          f"mean deg {dg.mean():.0f} max {dg.max()}")
    sel = dg[dg > f]
    h, _ = np.histogram(sel, bins=[f + 1] + edges[1:])
    for lo, hi, c in zip([f + 1] + edges[1:-1], edges[1:], h):
        m = (sel >= lo) & (sel < hi)
        print(f"   deg [{lo},{hi}): rows {c} ({100*c/max(valid.sum(),1):.1f}% of valid) sum deg {sel[m].sum()}")
eng.close()
