#!/usr/bin/env python3
"""Experiment: two plans on two streams, pipelined by PART instead of free-running — while one plan's layers
(gather + projection: bandwidth / MFMA-bound) run, the other builds the next batch set's graph (sample + union:
latency-bound); events keep two graph parts, and two layer parts, from overlapping each other.
Compares (eager launches throughout): one stream; two streams free-running; two streams pipelined by part."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gigl_amd import _lib  # noqa: E402
from gigl_amd.engine import HipEngine  # noqa: E402
from gigl_amd.models import GraphSAGE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--small", action="store_true")
ap.add_argument("--group", type=int, default=64)
ap.add_argument("--calls", type=int, default=60)
ap.add_argument("--plans", type=int, default=2)
a = ap.parse_args()
a.workload, a.batch, a.fanouts = "products", 1024, "25,10"
eng0 = HipEngine(0)
dev = eng0.device
n, d = bench.build_workload(eng0, a)
_, _, hid, out_dim, _, _ = a._workload
fan, B, G = [25, 10], 1024, a.group
torch.manual_seed(0)
model = GraphSAGE(d, hid, out_dim, num_layers=2).to(dev)
g = torch.Generator().manual_seed(42)
perm = torch.randperm(n, generator=g)
n_calls = a.calls
need = n_calls * G * B
if perm.numel() < need:
    perm = perm.repeat(-(-need // perm.numel()))
my = perm[:need].view(n_calls, G * B).to(torch.int32).to(dev).contiguous()
P = a.plans
engines = [eng0] + [HipEngine(0) for _ in range(P - 1)]
for e in engines[1:]:
    e.share_resident(eng0)
streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
plans, outs = [], []
for e, st in zip(engines, streams):
    e.bind_stream(st)
    plans.append(model.make_plan(e, B, fan, groups=G))
    outs.append(torch.empty((G * B, out_dim), dtype=torch.float32, device=dev))
lib = eng0._lib


def part(k, c, which):
    _lib.check(lib.gigl_sage_plan_run_part(plans[k]._plan, C.c_void_p(my[c].data_ptr()), 42, _lib.MODE_SPARK_HASH,
                                           C.c_void_p(outs[k].data_ptr()), which), engines[k]._ctx)


def whole(k, c):
    plans[k].run(my[c], out=outs[k])


def one_stream():
    for c in range(n_calls):
        whole(0, c)


def free_running():
    for c in range(n_calls):
        whole(c % P, c)


def pipelined(chain_graph=True, chain_layers=True):
    ev_g, ev_l = None, None
    for c in range(n_calls):
        k = c % P
        st = streams[k]
        if chain_graph and ev_g is not None:
            st.wait_event(ev_g)
        part(k, c, 1)
        ev_g = torch.cuda.Event()
        ev_g.record(st)
        if chain_layers and ev_l is not None:
            st.wait_event(ev_l)
        part(k, c, 2)
        ev_l = torch.cuda.Event()
        ev_l.record(st)


ref = None
for label, fn in (("one stream", one_stream), ("%d streams, free-running" % P, free_running),
                  ("%d streams, pipelined by part (graph parts chained, layer parts chained)" % P, pipelined),
                  ("%d streams, layer parts chained only" % P, lambda: pipelined(chain_graph=False))):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    chk = sum(float(o.double().sum()) for o in outs)
    print(f"{label:76s}: {best / (n_calls * G) * 1e6:7.2f} us/step  ({n_calls} calls of {G} batches; checksum {chk:.6e})", flush=True)
for p in plans:
    p.close()
for e in reversed(engines):
    e.close()
