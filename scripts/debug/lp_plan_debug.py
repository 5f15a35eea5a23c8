"""debug: the link-prediction training plan step by step against autograd (weights after every step)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle
from helpers import rmat_edges
from test_gpu_train_plan import _lp_batches, _lp_loss_torch
from gigl_amd.engine import HipEngine, NablpTrainPlan
from gigl_amd.models import GraphSAGE, HipBatch

s, d = rmat_edges(13, 150000, seed=8); n = 1 << 13
rowptr, col = oracle.build_csc(n, s, d, is_directed=False)
x = (np.random.default_rng(0).standard_normal((n, 100)) / 4).astype(np.float32)
eng = HipEngine(0); eng.load_csc(rowptr, col); eng.load_features(x)
dst = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rowptr).astype(np.int64))
eng.build_from_coo(n, dst, col.astype(np.uint32), is_directed=True, out_graph=True)
b, P, n_rn, fan, steps, temp = int(os.environ.get("B", 128)), int(os.environ.get("P", 1)), int(os.environ.get("NRN", 64)), [10, 5], 3, float(os.environ.get("TEMP", 0.07))
batches = _lp_batches(eng, n, b, P, n_rn, steps, seed=5)
if os.environ.get("SHORT"):
    batches = [(r[: (1 + P) * 40].contiguous(), c[:40].contiguous(), q[:30].contiguous()) for r, c, q in batches]
torch.manual_seed(4)
kw = dict(num_layers=2, should_l2_normalize_embedding_layer_output=os.environ.get("NORM", "1") == "1")
ref = GraphSAGE(100, 32, 16, **kw).to(eng.device); lib = GraphSAGE(100, 32, 16, **kw).to(eng.device)
lib.load_state_dict(ref.state_dict()); ref.train()
LR = float(os.environ.get('LR', 5e-3))
opt = torch.optim.Adam(ref.parameters(), lr=LR, weight_decay=1e-6)
st = torch.cuda.Stream(device=eng.device); torch.cuda.synchronize(); eng.bind_stream(st); torch.cuda.set_stream(st)
plan = NablpTrainPlan(eng, lib, b, P, n_rn, fan, temperature=temp, lr=LR, weight_decay=1e-6)
if os.environ.get('SAME'):
    batches = [batches[0]] * 3
for i, (roots, cnt, rn) in enumerate(batches):
    embs = []
    for r in (roots, rn):
        tree = eng.sample_khop(r, fan); u = eng.union_build(tree)
        embs.append(ref(HipBatch(eng, tree, u, train=True))[u.root_local[: r.numel()].long()])
    na = cnt.numel()
    loss = _lp_loss_torch(embs[0], embs[1], roots, cnt, rn, na, P, temp)
    opt.zero_grad(); loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    opt.step()
    got = plan.step(roots, cnt, rn).clone(); eng.synchronize()
    plan.store(lib)
    lines = []
    for l, conv in enumerate(ref.conv_layers):
        gw, gb = plan.grads(l)
        d_in = conv.in_channels
        rl, rr, rb = grads[f"conv_layers.{l}.lin_l.weight"], grads[f"conv_layers.{l}.lin_r.weight"], grads[f"conv_layers.{l}.lin_l.bias"]
        lines.append(f"   layer {l}: |gW_l - ref| {float((gw[:, :d_in] - rl).abs().max()):.3e} of {float(rl.abs().max()):.3e}   |gW_r - ref| {float((gw[:, d_in:] - rr).abs().max()):.3e} of {float(rr.abs().max()):.3e}"
              f"   |gb - ref| {float((gb - rb).abs().max()):.3e} of {float(rb.abs().max()):.3e}   ratio l {float((gw[:, :d_in] * rl).sum() / (rl * rl).sum()):.4f} r {float((gw[:, d_in:] * rr).sum() / (rr * rr).sum()):.4f} b {float((gb * rb).sum() / (rb * rb).sum()):.4f}")
    print(f"step {i}: loss plan {float(got[0]):.6f} ref {float(loss):.6f} rows {float(got[1])}")
    print("\n".join(lines))
    for (k, a), (_, bb) in zip(lib.state_dict().items(), ref.state_dict().items()):
        print(f"   {k:32s} max|dw| {float((a - bb).abs().max()):.3e}  (|w| {float(bb.abs().max()):.3e}, |grad| {float(grads[k].abs().max()):.3e})")
plan.close()
