#!/usr/bin/env python3
"""Throughput of the device-side record encoder (gigl_records_encode) on the bench workload:
products-shaped graph, fanout [25,10], D=100 fp32; B roots per call.  Prints bytes/s of finished TFRecord
frames and the HBM-roofline fraction (algorithmic bytes = record bytes written + 4*D per distinct node read +
4 B per tree slot read three times (plan x2, write))."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gigl_amd.engine import HipEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--small", action="store_true")
ap.add_argument("--device-only", action="store_true", help="only the back-to-back device timing (for rocprofv3 runs)")
ap.add_argument("--no-verify", action="store_true", help="ablation runs (GIGL_REC_SKIP): the bytes are not the records'")
a = ap.parse_args()
eng = HipEngine(0)
n, d = bench.build_workload(eng, a)
g = torch.Generator().manual_seed(42)
fan = [25, 10]
roots = torch.randperm(n, generator=g)[: a.batch].to(torch.int32).cuda()
tree = eng.sample_khop(roots, fan)
buf, off = eng.encode_records(tree)
nbytes = buf.numel()
recs = wire_ok = None
from gigl_amd import wire  # noqa: E402
if not a.no_verify:
    recs = list(wire.iter_tfrecords(buf[: int(off[8])].cpu().numpy().tobytes()))  # CRCs verified by the reader
slots = sum(fan[0] * (fan[1] if k else 1) for k in range(2)) + 1
torch.cuda.synchronize()
from gigl_amd.subgraph_sampler import _frames_to_host  # noqa: E402


def to_host():
    b, o = eng.encode_records(eng.sample_khop(roots, fan, out=tree))
    return _frames_to_host(b), o.cpu()




def device_time(calls=20):
    """GPU time of one gigl_records_encode call: `calls` calls issued back to back on the engine's stream into the
    same output buffer, bracketed by events on that stream (no host work in between)"""
    import ctypes as C
    from gigl_amd import _lib
    o = _lib.GiglRecordOpts()
    o.kind, o.trees_per_record, o.tfrecord_frame = _lib.REC_ROOTED_NODE_NEIGHBORHOOD, 1, 1
    o.condensed_node_type = o.condensed_edge_type = 0
    out = torch.empty(nbytes + (64 << 20), dtype=torch.uint8, device="cuda")
    rec_off = torch.empty(a.batch + 1, dtype=torch.int64, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")

    def call():
        _lib.check(eng._lib.gigl_records_encode(eng._ctx, C.c_void_p(tree.roots.data_ptr()), C.byref(tree.c_struct),
                                                eng._feat, C.byref(o), a.batch, C.c_void_p(out.data_ptr()), out.numel(),
                                                C.c_void_p(rec_off.data_ptr()), C.c_void_p(status.data_ptr())), eng._ctx)
    call()
    eng._stream.synchronize()
    assert a.no_verify or (int(status.item()) == 0 and torch.equal(out[:nbytes], buf))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(eng._stream)
    for _ in range(calls):
        call()
    e1.record(eng._stream)
    e1.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / calls
    print(f"encode (device time, {calls} calls back to back) B={a.batch} {dt*1e6:8.1f} us/call  {nbytes/dt/1e9:7.1f} GB/s of "
          f"records  roofline(write+feature read) ~{(2*nbytes + 4*slots*a.batch)/dt/8e12:.3f} of 8 TB/s")


device_time()
if a.device_only:
    eng.close()
    sys.exit(0)

for label, fn in (("sample+encode", lambda: eng.encode_records(eng.sample_khop(roots, fan, out=tree))),
                  ("encode", lambda: eng.encode_records(tree)),
                  ("sample+encode+PCIe", to_host)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"{label:14s} B={a.batch} {dt*1e3:8.3f} ms/call  {nbytes/dt/1e9:7.1f} GB/s of records "
          f"({nbytes/a.batch/1e3:.1f} KB/record, {a.batch/dt/1e6:.2f} M records/s)  "
          f"roofline(write+feature read) ~{(2*nbytes + 3*4*slots*a.batch)/dt/8e12:.3f} of 8 TB/s")
eng.close()
