"""EmbeddingExporter throughput alone and next to a busy main thread (the inference loop)"""
import os, shutil, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gigl_amd.export import EmbeddingExporter
dev = torch.device("cuda", 0)
n, d, calls = 65536, 47, 38
emb = torch.randn(n, d, device=dev)
ids = torch.arange(n, dtype=torch.int64)
scratch = tempfile.mkdtemp(prefix="gigl_exp_", dir="/dev/shm")
a = torch.randn(4096, 4096, device=dev)
for busy in (False, True, False, True):
    ex = EmbeddingExporter(os.path.join(scratch, f"e{int(busy)}"))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for c in range(calls):
        if busy:
            for _ in range(20):
                b = a @ a
        ex.add_embedding(ids, emb, "paper")
    t_enq = time.perf_counter() - t
    ex.close()
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    print(f"busy={busy}: {t*1e3:.1f} ms total, enqueue {t_enq*1e3:.1f} ms, {ex.bytes_written/t/1e9:.2f} GB/s, trace {ex.trace}")
    shutil.rmtree(os.path.join(scratch, f"e{int(busy)}"))
shutil.rmtree(scratch)
