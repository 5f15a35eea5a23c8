"""where a typed step's time goes as the batch grows: the one-call plan alone, the wrapper (counts read + feature rows),
the HGT forward alone — for B = 4096 .. 65536 (scripts/prof_typed_step.py profiles one size in detail)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gigl_amd import _lib  # noqa: E402
from gigl_amd.graphdb_sampler import INCOMING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG  # noqa: E402
from gigl_amd.models_hetero import HGT  # noqa: E402

na, npp, ne = 2_000_000, 4_000_000, 40_000_000
rng = np.random.default_rng(0)
a2p, p2a = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
src = (na * rng.random(ne) ** 2).astype(np.int64)
dst = rng.integers(0, npp, ne)
edges = {a2p: (src.astype(np.uint32), dst.astype(np.uint32)), p2a: (dst.astype(np.uint32), src.astype(np.uint32))}
feats = {"author": rng.standard_normal((na, 64)).astype(np.float32), "paper": rng.standard_normal((npp, 128)).astype(np.float32)}
s = HipGraphDBSampler({"author": 0, "paper": 1}, {"author": na, "paper": npp}, edges, {a2p: 0, p2a: 1}, feats)
dag = SamplingOpDAG.from_ops([SamplingOp("h1", a2p, 10, [], INCOMING), SamplingOp("h2", p2a, 5, ["h1"], INCOMING)])
ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
model = HGT({"author": 64, "paper": 128}, {e: 0 for e in ets}, hid_dim=64, out_dim=64, num_layers=2, num_heads=2).cuda().eval()
model.engine = s.engine
eng = s.engine


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for B in (4096, 16384, 32768, 65536):
    roots = rng.choice(npp, size=B, replace=False)
    pl = s.typed_plan("paper", dag, B)
    r_dev = torch.from_numpy(roots).to(torch.int32).cuda()
    t_plan = timed(lambda: _lib.check(eng._lib.gigl_typed_plan_run(pl["plan"], C.c_void_p(r_dev.data_ptr()), B), eng._ctx))
    t_wrap = timed(lambda: s.batch_graph_plan(roots, "paper", dag, b_max=B))
    g, ri, u = s.batch_graph_plan(roots, "paper", dag, b_max=B)

    def fwd():
        with torch.no_grad():
            model(g, ["paper"], row_subset={"paper": ri})
    t_fwd = timed(fwd)
    n_nodes = sum(int(v.numel()) for v in u.values())
    n_edges = sum(int(v.shape[1]) for v in g.edge_index_dict.values())
    print(f"B={B:6d}: plan {t_plan:7.3f} ms  plan+wrapper {t_wrap:7.3f} ms  HGT {t_fwd:7.3f} ms   ({n_nodes} nodes, {n_edges} edges)", flush=True)
s.close()
