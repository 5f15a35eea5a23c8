"""repro: library training plan on tiny shapes (b = 4, fanout [3,3], dims 2 -> 8 -> 3, 16-node graph)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle
from gigl_amd.engine import HipEngine, SageTrainPlan
from gigl_amd.models import GraphSAGE
n = int(os.environ.get("N", "16"))
rng = np.random.default_rng(0)
src, dst = rng.integers(0, n, 34), rng.integers(0, n, 34)
keep = src != dst
rowptr, col = oracle.build_csc(n, src[keep].astype(np.uint32), dst[keep].astype(np.uint32), is_directed=False)
eng = HipEngine(0)
eng.load_csc(rowptr, col)
eng.load_features(rng.standard_normal((n, 2)).astype(np.float32))
torch.manual_seed(0)
model = GraphSAGE(2, 8, 3, num_layers=2).to(eng.device)
st = torch.cuda.Stream(device=eng.device)
torch.cuda.synchronize()
eng.bind_stream(st)
b = 4
plan = SageTrainPlan(eng, model, b, [3, 3], lr=0.01, weight_decay=5e-4)
ids = torch.arange(14, dtype=torch.int32, device=eng.device)
labels = torch.from_numpy(rng.integers(0, 3, 14)).to(eng.device)
torch.cuda.synchronize()
for ep in range(3):
    with torch.cuda.stream(st):
        for lo in range(0, 14, b):
            nxt = ids[lo + b:lo + 2 * b] if os.environ.get("PREFETCH", "1") == "1" else None
            loss = plan.step(ids[lo:lo + b], labels[lo:lo + b], next_roots=nxt)
            print(ep, lo, "issued", flush=True)
            if os.environ.get("SYNC_EACH"):
                eng.synchronize(); torch.cuda.synchronize(); print("  ok", float(loss), flush=True)
    eng.synchronize()
    print("epoch", ep, float(loss), flush=True)
    ev = os.environ.get("EVAL_BETWEEN")
    if ev:
        plan.store(model)
        eng.bind_stream(torch.cuda.current_stream(eng.device))
        if ev in ("1", "norun", "keep", "noclose"):
            ip = model.make_plan(eng, 8, [3, 3], groups=1)
            if ev != "norun":
                out = ip.run(torch.arange(8, dtype=torch.int32, device=eng.device))
                torch.cuda.synchronize()
                print("  eval ok", float(out.sum()), flush=True)
            if ev in ("1", "norun"):
                ip.close()
            else:
                globals().setdefault("_kept", []).append(ip)
        elif ev == "sample":
            tree = eng.sample_khop(np.arange(8, dtype=np.uint32), [3, 3])
            torch.cuda.synchronize()
            print("  sample ok", flush=True)
        elif ev == "malloc":
            z = [torch.empty(1 << 20, device=eng.device) for _ in range(4)]
            del z
            torch.cuda.empty_cache()
        plan.load(model)
        eng.bind_stream(st)
plan.close(); eng.close(); print("done")
