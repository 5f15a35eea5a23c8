"""bench/entries.py — the reference entry points end to end: --entry inferencer / --entry sampler."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import _LIVE_PMC  # noqa: F401
from .cpu_baseline import run_cpu_records_baseline


def run_entry_inferencer(args, rank, world, local_rank):
    """--entry inferencer: the workload's full inference pass (every node a root, batches of B in the TFRecord route's
    order) through the drop-in entry point's own code: Inferencer.infer_resident -> plugin.infer_batch(HbmRootBatch) ->
    ResidentGraph.encode -> gigl_sage_plan_run, rows handed to the exporter (Avro encoded on the device, written out by
    its writer thread).  The graph is built in HBM by this script (ResidentGraph.from_engine) instead of being read from
    preprocessor tables — ingest is one-time work outside the step.  A replica per GPU at N > 1; a secondary line."""
    import shutil
    import tempfile
    from gigl_amd._lib import MODE_FAST, MODE_SPARK_HASH, STATS, STATS_LEN
    from gigl_amd.engine import HipEngine
    from gigl_amd.hbm import ResidentGraph
    from gigl_amd.inferencer import Inferencer, _RowWriter
    from gigl_amd.task_specs import HipGraphSageNodeClassificationSpec

    torch.cuda.set_device(local_rank)
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    B, G = args.batch, max(1, args.group)
    t0 = time.time()
    n, d = build_workload(eng, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    torch.manual_seed(0)
    spec = HipGraphSageNodeClassificationSpec(out_dim=out_dim, hid_dim=hid, num_layers=len(fanouts))
    from gigl_amd.models import GraphSAGE
    spec.model = GraphSAGE(d, hid, out_dim, num_layers=len(fanouts)).to(dev)
    mode = MODE_SPARK_HASH if args.mode == "parity" else MODE_FAST
    resident = ResidentGraph.from_engine(eng, np.arange(n, dtype=np.int64), fanouts, node_type="paper", mode=mode)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    scratch = tempfile.mkdtemp(prefix="gigl_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)

    class _NullWriter:  # rows stay in HBM
        n_rows = 0

        def add(self, ids, emb, pred, ids_dev=None):
            self.n_rows += int(ids.size)

        def close(self):
            pass

    inf = Inferencer()

    def one_pass(sink):
        w = (_RowWriter({"embeddings": os.path.join(scratch, "emb") + "/"}, "paper", keep_on_device=sink == "avro-device")
             if sink != "none" else _NullWriter())
        inf.infer_resident(spec, dev, resident, w, B, groups=G)
        w.close()
        torch.cuda.synchronize()
        return w

    try:
        one_pass("none")  # warm-up: hash table, plan, allocator
        # exact edge counts of the pass (untimed; sampling is deterministic)
        acc = torch.zeros(STATS_LEN, dtype=torch.int64, device=dev)
        n_steps = 0
        ids = resident.inference_root_order()
        for hb in resident.root_batches(ids, B, G):
            plan = resident._plan_for(spec.model, B, G)
            plan.run(hb.roots, sampling_seed=resident.seed, mode=mode)
            plan.stats(hb.roots, acc)
        torch.cuda.synchronize()
        st = acc.cpu().numpy().astype(np.float64)
        n_steps = -(-n // B)
        # the padding batches of the last call (one repeated root each) are part of the pass; their few edges are in `st`
        edges_pass = float(st[STATS["sampled"]] + st[STATS["aggregated"]])
        res, sink_trace = {}, None
        for sink in ("none", "avro-device", "avro-files"):
            if os.environ.get("GIGL_BENCH_PROFILE") == sink:
                import cProfile
                import pstats
                one_pass(sink)
                pr = cProfile.Profile()
                pr.enable()
                one_pass(sink)
                pr.disable()
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(25)
            one_pass(sink)
            shutil.rmtree(os.path.join(scratch, "emb"), ignore_errors=True)
            reps = []
            t_all = time.perf_counter()
            while time.perf_counter() - t_all < args.min_seconds or len(reps) < 3:
                if world > 1:
                    import torch.distributed as dist
                    dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                w = one_pass(sink)
                reps.append(time.perf_counter() - t1)
                if sink == "avro-files":
                    sink_trace = dict(w.exporter.trace, bytes=w.exporter.bytes_written)
                    shutil.rmtree(os.path.join(scratch, "emb"), ignore_errors=True)  # (untimed)
            res[sink] = np.array(reps)
        # plan level on the same roots and call shape, driven directly (no entry-point code, no rows consumed)
        plan = resident._plan_for(spec.model, B, G)
        batches = list(resident.root_batches(ids, B, G))
        out = torch.empty((G * B, out_dim), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for hb in batches:
            plan.run(hb.roots, out=out, sampling_seed=resident.seed, mode=mode)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t1
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    head = res[args.entry_sink]
    t_med = float(np.median(head))
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([t_med], dtype=torch.float64, device=dev)
        all_reduce(tt, dist.ReduceOp.MAX)
        t_med = float(tt.item())
    if rank == 0:
        line = {
            "metric": "sampled+aggregated edges/s", "value": edges_pass * world / t_med, "unit": "edges/s",
            "n_gpus": world, "steps": int(n_steps * len(head)), "warmup": n_steps, "ms_per_step": t_med / n_steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_label + f" N={n} E={eng.n_edges} D={d} fanout={fanouts} B={B} GraphSAGE "
                                            f"{d}->{hid}->{out_dim}: FULL inference pass over every node through "
                                            "Inferencer.infer_resident (drop-in entry point, in-HBM route), sampler "
                                            "mode=" + args.mode,
                       "entry": "gigl_amd.inferencer.Inferencer.infer_resident -> HipGraphSageNodeClassificationSpec."
                                "infer_batch(HbmRootBatch) -> ResidentGraph.encode -> gigl_sage_plan_run",
                       "sink": {"avro-device": "Avro data blocks encoded on the device, left in HBM (outputs resident)",
                                "avro-files": "Avro shards: device-encoded, copied out and written to tmpfs by the "
                                              "exporter's writer thread (PCIe + file inclusive)",
                                "none": "bare rows, left in HBM"}[args.entry_sink],
                       "pcie_and_file_inclusive_pass_s_median": float(np.median(res["avro-files"])),
                       "pcie_and_file_inclusive_roots_per_s": n * world / float(np.median(res["avro-files"])),
                       "batches_per_call": G, "roots_per_s": n * world / t_med,
                       "pass_s_median": t_med, "pass_s_all": [round(float(v), 4) for v in head],
                       "compute_only_pass_s_median": float(np.median(res["none"])),
                       "compute_only_ms_per_step": float(np.median(res["none"])) / n_steps * 1e3,
                       "plan_level_pass_s": plan_s, "plan_level_ms_per_step": plan_s / n_steps * 1e3,
                       "entry_over_plan": t_med / plan_s, "sink_trace_last_pass": sink_trace,
                       "sampled_edges_per_step": float(st[STATS["sampled"]]) / n_steps,
                       "aggregated_edges_per_step": float(st[STATS["aggregated"]]) / n_steps,
                       "setup_s": round(setup_s, 1)},
            "roofline": None, "cpu_baseline": None,
        }
        emit(line)
    resident.close()
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_entry_sampler(args, rank, world, local_rank):
    """--entry sampler: the Subgraph Sampler job's step on the workload's graph — a batch of B roots sampled k hops
    (gigl_sample_khop, parity mode) and encoded as framed RootedNodeNeighborhood TFRecords on the device
    (gigl_records_encode: per-root dedup, hydration from the resident table, proto3 + TFRecord framing with both
    CRC-32C words), SGSPureSparkV1Task.scala:313-820 + TFRecordIO.scala:53-69.  Inputs and outputs resident in HBM (the
    job's device-to-host copy of finished frames is the PCIe-inclusive figure of scripts/micro_records.py).  Calls are
    issued back to back on the engine's stream into one output buffer; a replica per GPU at N > 1; a secondary line."""
    import ctypes as C
    from gigl_amd import _lib
    from gigl_amd.engine import HipEngine

    torch.cuda.set_device(local_rank)
    eng = HipEngine(local_rank)
    dev = eng.device
    fanouts = [int(v) for v in args.fanouts.split(",")]
    B = args.batch
    t0 = time.time()
    n, d = build_workload(eng, args)
    wl_name, wl_label, hid, out_dim, wl_directed, wl_dtype = args._workload
    g = torch.Generator().manual_seed(42)
    perm = torch.randperm(n, generator=g)
    n_batches = max(8, min(64, n // B // max(world, 1)))
    pool = [perm[(rank + world * i) * B:(rank + world * i + 1) * B].to(torch.int32).to(dev) for i in range(n_batches)]
    trees = [eng.alloc_tree(B, fanouts) for _ in range(2)]
    # sizes and content once, through the public entry (also builds the per-row CRC table: one-time, reported)
    t1 = time.time()
    tbl = C.c_void_p()
    _lib.check(eng._lib.gigl_features_row_crc(eng._ctx, eng._feat, C.byref(tbl)), eng._ctx)
    eng._stream.synchronize()
    row_crc_s = time.time() - t1
    sizes, edges_b, nodes_b = [], [], []
    for r in pool:
        tree = eng.sample_khop(r, fanouts, out=trees[0])
        buf, off = eng.encode_records(tree)
        sizes.append(int(buf.numel()))
        edges_b.append(int(sum(int((t_ != -1).sum().item()) for t_ in tree.nbr)))
    from gigl_amd import wire
    head = buf[: int(off[4].item())].cpu().numpy().tobytes()
    n_ok = sum(1 for _ in wire.iter_tfrecords(head))  # (the reader verifies both CRC words of every frame)
    assert n_ok == 4
    cap = max(sizes) + 4096
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    rec_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    o = _lib.GiglRecordOpts()
    o.kind, o.trees_per_record, o.tfrecord_frame = _lib.REC_ROOTED_NODE_NEIGHBORHOOD, 1, 1
    o.condensed_node_type = o.condensed_edge_type = 0
    fo = (C.c_int32 * len(fanouts))(*fanouts)

    def step(i, encode=True, sample=True):
        tree = trees[i & 1]
        r = pool[i % n_batches]
        if sample:
            tree.roots = r
            _lib.check(eng._lib.gigl_sample_khop(eng._ctx, eng._graph, C.c_void_p(r.data_ptr()), B, fo, len(fanouts), 42,
                                                 _lib.MODE_SPARK_HASH if args.mode == "parity" else _lib.MODE_FAST,
                                                 C.byref(tree.c_struct)), eng._ctx)
        if encode:
            _lib.check(eng._lib.gigl_records_encode(eng._ctx, C.c_void_p(r.data_ptr()), C.byref(tree.c_struct), eng._feat,
                                                    C.byref(o), B, C.c_void_p(out.data_ptr()), cap,
                                                    C.c_void_p(rec_off.data_ptr()), C.c_void_p(status.data_ptr())),
                       eng._ctx)

    for tr in trees:
        tr.c_struct.hops, tr.c_struct.b = len(fanouts), B
        for k, f in enumerate(fanouts):
            tr.c_struct.fanouts[k] = f
    for i in range(max(4, args.warmup // 8)):
        step(i)
    eng._stream.synchronize()
    assert int(status.item()) == 0
    setup_s = time.time() - t0
    K_rep = max(n_batches, -(-max(1, args.steps // 8) // n_batches) * n_batches)

    def timed(reps, **kw):
        ts = []
        for _ in range(reps):
            if world > 1:
                import torch.distributed as dist
                dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_w = time.perf_counter()
            e0.record(eng._stream)
            for i in range(K_rep):
                step(i, **kw)
            e1.record(eng._stream)
            e1.synchronize()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t_w, e0.elapsed_time(e1) * 1e-3))
        return np.array(ts)

    reps = args.min_reps
    while True:
        full = timed(reps)
        if full[:, 0].sum() >= args.min_seconds or reps >= 4096:
            break
        reps *= 2
    enc_only = timed(max(3, reps // 4), sample=False)  # the encoder's share: HIP events on its stream, same calls
    wall = torch.tensor(full[:, 0], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce(wall, torch.distributed.ReduceOp.MAX)
    wall = wall.cpu().numpy()
    elapsed = float(wall.sum())
    steps_total = reps * K_rep
    sampled_per_step = float(np.mean(edges_b))
    bytes_per_step = float(np.mean(sizes))
    # algorithmic bytes of the encoder per call: the finished record bytes written + 4*D read per
    # DISTINCT node of every record (what the payloads are copied from) + the tree slots read once
    slots = 1 + sum(int(np.prod(fanouts[:k + 1])) for k in range(len(fanouts)))
    # node fields of a step, from the record sizes: bytes = fields * (4 D + ~10 header bytes) + edges * ~12.5 + ~30 / record
    fields_per_step = max(0.0, (bytes_per_step - 12.5 * sampled_per_step - 30.0 * B) / (4 * d + 10))
    enc_ms = float(np.median(enc_only[:, 1])) / K_rep * 1e3
    alg_bytes = bytes_per_step + min(fields_per_step, B * slots) * 4 * d + 4.0 * slots * B
    achieved = alg_bytes / (enc_ms * 1e-3) / 1e9
    # HBM traffic of one encode call from the committed counter summary (scripts/pmc_records.sh: FETCH_SIZE / WRITE_SIZE
    # in separate rocprofv3 passes over the same call shape — products-shaped graph, [25,10], 4,096 records): KB units;
    # fetches of 16-byte-per-lane reads are tallied at half their bytes on gfx950 (MI355X_MICROARCH.md), hence x2
    traffic, traffic_src = None, None
    if wl_name == "products" and B == 4096 and fanouts == [25, 10]:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_encoder_pmc.json")))[::-1]:
            try:
                c = json.load(open(f))
                traffic = sum(1024.0 * c[k + ".WRITE_SIZE"]["mean"] + 2048.0 * c[k + ".FETCH_SIZE"]["mean"]
                              for k in ("record_plan", "record_write"))
                traffic_src = os.path.basename(f)
                break
            except Exception:  # noqa: BLE001 — another layout: no traffic figure
                continue
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_records_baseline(eng, pool[0], fanouts, d)
    if rank == 0:
        q = lambda a, p: float(np.percentile(a, p))
        ms_rep = wall / K_rep * 1e3
        line = {
            "metric": "sampled edges/s (sampler job step: sample + encode records)",
            "value": sampled_per_step * steps_total * world / elapsed, "unit": "edges/s", "n_gpus": world,
            "steps": steps_total, "warmup": max(4, args.warmup // 8), "ms_per_step": elapsed / steps_total * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "timing": {"repetitions": reps, "steps_per_repetition": K_rep, "timed_region_s": round(elapsed, 3),
                       "ms_per_step_median": q(ms_rep, 50), "ms_per_step_p10": q(ms_rep, 10),
                       "ms_per_step_p90": q(ms_rep, 90)},
            "config": {"workload": wl_label + f" N={n} E={eng.n_edges} D={d} fp32 features, fanout={fanouts}, B={B} roots "
                                            "per step: k-hop sample (sampler mode=" + args.mode + ") + framed "
                                            "RootedNodeNeighborhood TFRecords encoded on the device, records left in HBM",
                       "entry": "gigl_sample_khop + gigl_records_encode (what SubgraphSampler.run issues per batch)",
                       "records_per_s": B * steps_total * world / elapsed,
                       "record_bytes_per_s": bytes_per_step * steps_total * world / elapsed,
                       "bytes_per_record": bytes_per_step / B, "sampled_edges_per_step": sampled_per_step,
                       "encode_only_ms_per_step": enc_ms, "row_crc_table_build_s": round(row_crc_s, 4),
                       "setup_s": round(setup_s, 1)},
            "roofline": {"bound": "hbm", "kernel": "gigl_records_encode (record_plan + record_scan + record_write)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None if traffic is None else round(traffic),
                         "traffic_source": traffic_src,
                         "alg_bytes_per_launch": round(alg_bytes), "avg_launch_us": round(enc_ms * 1e3, 1),
                         "launches": int(max(3, reps // 4) * K_rep),
                         "timing": "HIP events on the engine's stream around back-to-back encode calls (no sampling "
                                   "in between), median over repetitions",
                         "node_fields_per_step": round(fields_per_step),
                         "bytes": "record bytes written + 4*D read per node field (fields estimated from the record "
                                  "sizes) + 4 B per tree slot read"},
            "cpu_baseline": cpu_baseline,
        }
        emit(line)
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
