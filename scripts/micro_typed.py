"""typed (heterogeneous) sampler throughput on a DBLP-shaped synthetic graph: SamplingOp-DAG execution + device-side
typed record encoding, RootedNodeNeighborhood and NodeAnchorBasedLinkPredictionSample records per second
usage (GPU box): python scripts/micro_typed.py [n_authors n_papers edges_per_type batch]"""
import sys
import time

import numpy as np
import torch

from gigl_amd.graphdb_sampler import INCOMING, EdgeType, HipGraphDBSampler, SamplingOp, SamplingOpDAG

na, npp, ne, B = (int(v) for v in (sys.argv[1:5] + ["500000", "1000000", "8000000", "4096"])[:4])
rng = np.random.default_rng(0)
a2p, p2a = EdgeType("author", "writes", "paper"), EdgeType("paper", "written_by", "author")
src = (na * rng.random(ne) ** 2).astype(np.int64)  # skewed authors: the busiest wrote ~ne / sqrt(na) papers
dst = rng.integers(0, npp, ne)
edges = {a2p: (src.astype(np.uint32), dst.astype(np.uint32)), p2a: (dst.astype(np.uint32), src.astype(np.uint32))}
feats = {"author": rng.standard_normal((na, 64)).astype(np.float32), "paper": rng.standard_normal((npp, 128)).astype(np.float32)}
t0 = time.time()
s = HipGraphDBSampler({"author": 0, "paper": 1}, {"author": na, "paper": npp}, edges, {a2p: 0, p2a: 1}, feats)
print(f"graph resident in {time.time() - t0:.1f} s: {na} authors, {npp} papers, {ne} edges per type", flush=True)
dag_paper = SamplingOpDAG.from_ops([SamplingOp("h1", a2p, 10, [], INCOMING), SamplingOp("h2", p2a, 5, ["h1"], INCOMING)])
dag_author = SamplingOpDAG.from_ops([SamplingOp("h1", p2a, 10, [], INCOMING), SamplingOp("h2", a2p, 5, ["h1"], INCOMING)])


def timed(fn, reps=6):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    tot = 0
    for _ in range(reps):
        tot += fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, tot / reps


roots = rng.integers(0, npp, B)
sec, nbytes = timed(lambda: int(s.encode_records_device(roots, "paper", dag_paper)[0].numel()))
print(f"RootedNodeNeighborhood  [10,5] B={B}: {sec * 1e3:8.2f} ms/batch  {B / sec:10.0f} records/s  {nbytes / sec / 1e6:8.1f} MB/s "
      f"({nbytes / B:.0f} B/record)", flush=True)
sec, nbytes = timed(lambda: int(s.encode_nablp_records_device(roots, p2a, 2, dag_paper, dag_author)[0].numel()))
print(f"NABLP sample, 2 positives B={B}: {sec * 1e3:8.2f} ms/batch  {B / sec:10.0f} records/s  {nbytes / sec / 1e6:8.1f} MB/s "
      f"({nbytes / B:.0f} B/record)", flush=True)
from gigl_amd.subgraph_sampler import _frames_to_host


def to_host():
    buf, off = s.encode_records_device(roots, "paper", dag_paper)
    h = _frames_to_host(buf)
    off.cpu()
    return int(h.size)


sec, nbytes = timed(to_host)
print(f"RootedNodeNeighborhood, frames copied to pinned host memory: {sec * 1e3:8.2f} ms/batch  {B / sec:10.0f} records/s  "
      f"{nbytes / sec / 1e6:8.1f} MB/s", flush=True)
# where the time goes: the DAG alone, the encoder alone
r32 = torch.tensor(roots.astype(np.int32)).cuda()
sec_dag, _ = timed(lambda: (s.run_dag(r32, dag_paper), s.engine.synchronize(), 0)[2])
print(f"  op DAG alone (2 ops): {sec_dag * 1e3:8.2f} ms/batch", flush=True)

# ---- the typed batch graph (what a trainer / inferencer batch needs): staged (one library call per op + torch.unique /
#      searchsorted chains) against the one-call plan (gigl_typed_plan_*: one stream of device work, one host read)


def edges_of(g):
    return sum(int(v.shape[1]) for v in g.edge_index_dict.values())


g0, _, u0 = s.batch_graph(roots, "paper", dag_paper)
g1, _, u1 = s.batch_graph_plan(roots, "paper", dag_paper, b_max=B)
assert all(torch.equal(u0[t], u1[t]) for t in u0) and edges_of(g0) == edges_of(g1)
for label, fn in (("batch graph, staged (torch.unique chains)", lambda: edges_of(s.batch_graph(roots, "paper", dag_paper)[0])),
                  ("batch graph, one-call plan", lambda: edges_of(s.batch_graph_plan(roots, "paper", dag_paper, b_max=B)[0]))):
    sec, ne_ = timed(fn)
    print(f"{label:44s} B={B}: {sec * 1e3:8.2f} ms/batch  {B / sec:10.0f} roots/s  {ne_ / sec / 1e6:8.1f} M distinct edges/s "
          f"({sum(int(v.numel()) for v in u1.values())} nodes, {int(ne_)} edges per batch)")
import ctypes as C  # noqa: E402
from gigl_amd import _lib  # noqa: E402
pl = s.typed_plan("paper", dag_paper, B)
r_dev = torch.from_numpy(np.asarray(roots, dtype=np.int64)).to(torch.int32).cuda()
eng = s.engine


def plan_only():
    _lib.check(eng._lib.gigl_typed_plan_run(pl["plan"], C.c_void_p(r_dev.data_ptr()), B), eng._ctx)
    return 0


plan_only()
eng._stream.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(eng._stream)
for _ in range(20):
    plan_only()
e1.record(eng._stream)
e1.synchronize()
print(f"  gigl_typed_plan_run alone (device time, 20 back to back): {e0.elapsed_time(e1) / 20:8.3f} ms/batch")
# ---- a typed inference step: batch graph through the one-call plan + 2-layer HGT over it + the roots' rows
from gigl_amd.models_hetero import HGT  # noqa: E402
torch.manual_seed(0)
ets = [("author", "writes", "paper"), ("paper", "written_by", "author")]
model = HGT({"author": 64, "paper": 128}, {e: 0 for e in ets}, hid_dim=64, out_dim=64, num_layers=2, num_heads=2).cuda().eval()
model.engine = eng


def infer_step():
    g, ri, _ = s.batch_graph_plan(roots, "paper", dag_paper, b_max=B, edge_type_ids=model.convs[0].edge_types_map)
    with torch.no_grad():
        out = model(g, ["paper"], row_subset={"paper": ri})["paper"]  # the last layer computes the roots' rows only
    return int(out.shape[0])


sec, _ = timed(infer_step)
g_, _, _ = s.batch_graph_plan(roots, "paper", dag_paper, b_max=B)


def fwd_only():
    with torch.no_grad():
        return int(model(g_, ["paper"])["paper"].shape[0])


sec_f, _ = timed(fwd_only)
print(f"typed inference step (one-call plan + 2-layer HGT 64/64, heads 2) B={B}: {sec * 1e3:8.2f} ms/batch  {B / sec:10.0f} roots/s "
      f"(HGT forward alone {sec_f * 1e3:.2f} ms)")
s.close()
