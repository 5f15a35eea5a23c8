"""micro-benchmark: expand time vs row degree (GPU box).  usage: python scripts/micro_expand.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gigl_amd.engine import HipEngine

eng = HipEngine(0)
degs = [10, 26, 64, 65, 128, 256, 300, 512, 1024, 2048, 4096, 16384, 65536, 262144]
n = 600_000
rowptr = np.zeros(n + 1, dtype=np.int64)
for i, d in enumerate(degs):
    rowptr[i + 1] = d
rowptr = np.cumsum(rowptr)
e = int(rowptr[-1])
col = np.concatenate([np.sort(np.random.default_rng(i).choice(n, size=d, replace=False)) for i, d in enumerate(degs)]).astype(np.uint32)
eng.load_csc(rowptr, col)
for B in (1024, 25600):
    for i, d in enumerate(degs):
        roots = torch.full((B,), i, dtype=torch.int32, device=eng.device)
        # distinct K per parent: use hop 2 style? hop-1 K = root id -> same window for all; fine for timing
        tree = eng.alloc_tree(B, [25])
        for _ in range(3):
            eng.sample_khop(roots, [25], out=tree)
        eng.profile_enable(["expand"], 64)
        for _ in range(10):
            eng.sample_khop(roots, [25], out=tree)
        ms, nl = eng.profile_read("expand")
        eng.profile_enable([], 0)
        print(f"B={B:6d} deg={d:7d}  {ms / nl * 1e3:8.1f} us/launch   {ms / nl * 1e6 / B:8.1f} ns/parent")
