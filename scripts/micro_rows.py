"""random-row fetch rate by table size: out = x[idx] (torch index_select) for 70k random rows of 1.5 KB
usage (GPU box): python scripts/micro_rows.py"""
import time
import torch

dev = torch.device("cuda:0")
d = 768
for n in (1 << 20, 1 << 22, 1 << 24, 30_520_062):
    x = torch.empty((n, d), dtype=torch.float16, device=dev)
    x.fill_(1)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for rows in (70_000, 700_000):
        idx = [torch.randint(0, n, (rows,), device=dev, generator=g) for _ in range(8)]
        out = torch.empty((rows, d), dtype=torch.float16, device=dev)
        for i in idx:
            torch.index_select(x, 0, i, out=out)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            for i in idx:
                torch.index_select(x, 0, i, out=out)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / 40
        print(f"table {n * d * 2 / 2**30:6.1f} GiB rows={rows:7d}: {us:8.1f} us  read {rows * d * 2 / us / 1e6:6.2f} TB/s", flush=True)
    del x
    torch.cuda.empty_cache()
