"""where the inference output sink spends its time: D2H through pinned staging, file write to tmpfs / disk"""
import os, time, tempfile, numpy as np, torch
n = 12 << 20
dev = torch.device("cuda", 0)
x = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter(); pin.copy_(x, non_blocking=True); torch.cuda.synchronize(); d2h = time.perf_counter() - t
print(f"D2H 12 MiB pinned: {d2h*1e3:.2f} ms = {n/d2h/1e9:.1f} GB/s")
mv = memoryview(pin.numpy())
for d in ("/dev/shm", "/tmp", "."):
    if not os.path.isdir(d):
        continue
    p = os.path.join(d, "gigl_sink_test.bin")
    t = time.perf_counter()
    with open(p, "wb") as f:
        for _ in range(38):
            f.write(mv)
    w = time.perf_counter() - t
    print(f"write 38 x 12 MiB to {d}: {w*1e3:.1f} ms = {38*n/w/1e9:.2f} GB/s")
    t = time.perf_counter()
    with open(p, "wb", buffering=0) as f:
        for _ in range(38):
            f.write(mv)
    w = time.perf_counter() - t
    print(f"  unbuffered rewrite: {w*1e3:.1f} ms = {38*n/w/1e9:.2f} GB/s")
    os.remove(p)
print("cpus", os.cpu_count())
