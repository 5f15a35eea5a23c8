"""micro-benchmark of gigl_linear / gigl_gather_mean (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigl_amd.engine import HipEngine
eng = HipEngine(0)
dev = eng.device
def bench_linear(M, K, N, cap):
    a = torch.randn(cap, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(cap, N, device=dev); m = torch.tensor([M], dtype=torch.int32, device=dev)
    for _ in range(3): eng.linear(a, w, b, m, cap, 1, out=y)
    eng.profile_enable(["linear"], 64)
    for _ in range(20): eng.linear(a, w, b, m, cap, 1, out=y)
    ms, n = eng.profile_read("linear"); eng.profile_enable([], 0)
    us = ms / n * 1e3
    print(f"linear M={M} K={K} N={N} cap={cap}: {us:.1f} us  {2*M*K*N/us/1e6:.1f} TFLOP/s")
for M, cap in [(20000, 20000), (20000, 282624), (27000, 27000), (1024, 1024), (1024, 282624), (282624, 282624)]:
    bench_linear(M, 200, 256, cap)
bench_linear(1024, 512, 47, 1024)
bench_linear(1024, 512, 47, 282624)
bench_linear(100000, 1536, 256, 100000)
