"""pieces of the projected-input precompute at the mag-shard shape: allocation, fp16 -> fp32 widening, the product"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gigl_amd.engine import HipEngine, dev_i32
eng = HipEngine(0)
dev = eng.device
n, d, hid = 8_000_000, 768, 256
x = torch.randn((1 << 20, d), device=dev).to(torch.float16).repeat(8, 1)[:n].contiguous()
eng.load_features(x)
w = torch.randn(hid, 2 * d, device=dev) * 0.05
def t(f, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps, r
dt, out = t(lambda: torch.empty((n, 2 * hid), dtype=torch.float32, device=dev)); print(f"alloc {n*2*hid*4/1e9:.1f} GB: {dt*1e3:.1f} ms")
del out
for k in range(2):
    dt, p = t(lambda: eng.project_features(w)); print(f"project_features: {dt*1e3:.1f} ms = {2*n*d*2*hid/dt/1e12:.1f} TF"); del p
m = 1 << 19
a = torch.randn(m, d, device=dev)
for nn in (256, 512):
    wc = torch.randn(nn, d, device=dev)
    y = torch.empty(m, nn, device=dev)
    eng.linear(a, wc, None, dev_i32(dev, m), m, 0, out=y)
    dt, _ = t(lambda: eng.linear(a, wc, None, dev_i32(dev, m), m, 0, out=y), 5); print(f"linear M={m} K={d} N={nn}: {dt*1e3:.2f} ms = {2*m*d*nn/dt/1e12:.1f} TF")
dt, _ = t(lambda: x[:m].float(), 5); print(f"widen {m} rows: {dt*1e3:.2f} ms")
