import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gigl_amd.export import EmbeddingExporter
import numpy as np
dev = torch.device("cuda", 0)
n, d, calls = 65536, 47, 38
emb = torch.randn(n, d, device=dev)
ids = torch.arange(n, dtype=torch.int64)
ex = EmbeddingExporter("/dev/shm/x_unused", keep_on_device=True)
for _ in range(3):
    ex.add_embedding(ids, emb, "paper")
torch.cuda.synchronize()
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
for c in range(calls):
    ex.add_embedding(ids, emb, "paper")
pr.disable()
t1 = time.perf_counter() - t
torch.cuda.synchronize()
print(f"enqueue {t1*1e3:.1f} ms, with sync {(time.perf_counter()-t)*1e3:.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
