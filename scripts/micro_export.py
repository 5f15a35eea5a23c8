"""timing of the device-side Avro embedding encoder (gigl_avro_embeddings_encode) at the products-sized output"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gigl_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0)
for n, d in ((2_449_029, 128), (2_449_029, 32), (1_000_000, 768)):
    emb = torch.randn(n, d, device=eng.device)
    ids = torch.randperm(n, device=eng.device)
    for _ in range(2):
        blocks, off = eng.encode_avro_embeddings(ids, emb, "user", bytes(16))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 5
    for _ in range(R):
        blocks, off = eng.encode_avro_embeddings(ids, emb, "user", bytes(16))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    host = blocks.cpu()
    t1 = time.perf_counter() - t0
    stage = torch.empty(blocks.numel(), dtype=torch.uint8, pin_memory=True)
    stage.copy_(blocks, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stage.copy_(blocks, non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"n={n} d={d}: {blocks.numel() / 1e6:.1f} MB in {dt * 1e3:.2f} ms = {blocks.numel() / dt / 1e9:.1f} GB/s of Avro bytes "
          f"({n / dt / 1e6:.1f} M records/s); device->host copy {t1 * 1e3:.1f} ms pageable, {t2 * 1e3:.1f} ms pinned")
    del emb, ids, blocks, off, host, stage
