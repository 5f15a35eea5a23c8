"""bench/cli.py — the command line of bench.py: flags, launcher / rank environment, dispatch to the workload modules."""
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
from .entries import run_entry_inferencer, run_entry_sampler
from .gat_lp import run_gat_lp, run_gat_lp_train, run_gat_lp_train_plan
from .products import run_products
from .sharded import run_emulated_world, run_sharded
from .train import run_lp_train, run_train
from .typed import run_typed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=960,
                    help="MINIMUM number of timed steps; the timed range is rounded up to whole rounds and repeated "
                         "until --min-seconds (the regime — streams x batches per call — does not depend on it)")
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--min-seconds", type=float, default=2.5, help="lower bound of the timed region")
    ap.add_argument("--min-rounds", type=int, default=10,
                    help="rounds (streams x batches-per-call steps) per timed repetition, at least")
    ap.add_argument("--min-reps", type=int, default=7, help="timed repetitions, at least (median / p10 / p90)")
    ap.add_argument("--batch", type=int, default=0, help="roots per batch (0: the workload's: 1024)")
    ap.add_argument("--fanouts", type=str, default="", help="per-hop fanouts (empty: the workload's: 25,10)")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--group", type=int, default=64,
                    help="batches per library call: G independent batches of B roots share one set of launches "
                         "(each keeps its own union graph; results are bit-identical to G single-batch calls)")
    ap.add_argument("--workload", type=str, default="products",
                    choices=["products", "mag-shard", "mag240m-sharded", "cora", "rmat-shard", "gat-lp", "typed-dblp"],
                    help="products = BASELINE configs[1] (default, the N=1 workload; N>1: a replica per GPU); mag-shard = "
                         "one GPU's 1/8 share of the MAG240M-shaped graph as a self-contained graph (D=768 fp16, SAGE "
                         "768->256->256); mag240m-sharded = BASELINE configs[2]: the MAG240M-shaped graph hash-"
                         "partitioned over the ranks (owner = id %% world), per-hop all_to_all frontier exchange and "
                         "feature pull over RCCL — needs >= 2 GPUs at full size (--shard-scale shrinks it)")
    ap.add_argument("--shard-group", type=int, default=None,
                    help="mag240m-sharded: batches of B roots exchanged per set of collectives (dedup stays per batch); "
                         "default 32, 16 under --emulate-world (eight ranks' workspaces share one GPU's HBM)")
    ap.add_argument("--shard-hot-frac", type=float, default=-1.0,
                    help="mag240m-sharded: fraction of the nodes (the most-referenced ones) whose feature rows are "
                         "replicated on every rank and never pulled (hub-row replication); -1 (default) = auto: on "
                         "whenever world > 1, sized to 4 %% of the free HBM, at most 5 %% of the nodes")
    ap.add_argument("--shard-encoder", type=str, default="sage", choices=["sage", "gat"],
                    help="mag240m-sharded: sage = GraphSAGE 768->256->256 through gigl_dist_plan (dense pull bookkeeping, "
                         "hot rows); gat = BASELINE configs[4]'s encoder, 2-layer GAT heads 2 hid 128 out 128, through "
                         "gigl_dist_gat_plan (raw rows, generic union)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="mag240m-sharded on ONE GPU: all W ranks of the hash-partitioned job as ctxs of this process "
                         "(in-process transport): per-rank pulled rows / bytes / bucket fill / hub-row hit rate / compute "
                         "time are measured, the W-GPU step is projected from them (labelled so)")
    ap.add_argument("--no-emulated-sub", action="store_true",
                    help="N=1 headline: skip the `sharded_emulated` sub-record (an 8-rank emulated world at a reduced scale, "
                         "run in a child process)")
    ap.add_argument("--no-sharded-sub", action="store_true",
                    help="N > 1 headline: skip the `sharded` sub-record (the mag240m-sharded workload at this N)")
    ap.add_argument("--graph-priority", type=str, default="off", choices=["off", "high", "same", "shared", "shared-high"],
                    help="products-family workloads: issue every call's graph part (sample + union) on a stream of its own — "
                         "high: of a higher priority than the layers' stream (gigl_sage_plan_set_graph_stream), same: equal "
                         "priority (A/B of the split alone)")
    ap.add_argument("--shard-route", type=str, default="auto", choices=["auto", "bucketed", "peer", "peer-all", "both"],
                    help="mag240m-sharded: how feature rows reach the first layer — bucketed (claim -> id exchange -> owners "
                         "gather -> row exchange over the transport), peer (rows read in place from the owners' tables mapped "
                         "into the reader: gigl_dist_plan_opts.peer_direct), peer-all (… and the owners' graph shards too: every "
                         "rank expands its own frontier over them, gigl_dist_plan_opts.peer_sample — a step without any "
                         "exchange), both (--emulate-world: measure all of them side by side), auto (both under "
                         "--emulate-world, peer-all when every rank can map its peers' memory)")
    ap.add_argument("--emulate-streams", type=int, default=3,
                    help="--emulate-world: emulated worlds in flight (each W ranks on one stream, a step of all its ranks "
                         "replayed as one hipGraph) for the OVERLAPPED per-rank step; 0 skips that measurement")
    ap.add_argument("--shard-plans", type=int, default=3, help="mag240m-sharded: sharded plans in flight per rank")
    ap.add_argument("--shard-scale", type=float, default=0.0,
                    help="mag240m-sharded: fraction of MAG240M's nodes and edges to generate (0 = world/8, capped at 1: "
                         "every GPU holds the share it has in the 8-GPU job; 1.0 needs 8 GPUs' HBM)")
    ap.add_argument("--project-on-owner", action="store_true",
                    help="mag240m-sharded: owners apply the first layer's weights before sending (256 fp32 per row "
                         "instead of 768 fp16)")
    ap.add_argument("--small", action="store_true", help="200k-node graph (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="counter-collection runs (scripts/gpu_pmc.sh): only warm-up + the timed region, so every "
                         "library launch in the trace is a grouped launch; prints timing without edge counts")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N=1 headline: skip the rocprofv3 counter passes that measure this run's HBM traffic per kernel "
                         "(roofline.traffic then comes from the newest committed profiles/*_pmc*.json of the workload, or is "
                         "null); the passes run as child processes after the timed region, ~1.5 min")
    ap.add_argument("--mode", type=str, default="parity", choices=["parity", "fast"])
    ap.add_argument("--project-input", type=str, default="auto", choices=["auto", "on", "off"],
                    help="first layer over PROJECTED rows (X W_l^T / X W_r^T computed once over the resident table, "
                         "gigl_sage_plan_set_projected_input): auto = when projected rows are narrower than stored rows "
                         "(mag-shard: 768 fp16 -> 256 fp32; not products).  The precompute is timed and charged to "
                         "every step as 1 / (steps of a full inference pass = N / B) of its duration")
    ap.add_argument("--train", action="store_true",
                    help="training step instead of the inference step: a batch sampled in HBM (sample + union graph), "
                         "GraphSAGE forward with autograd over the union graph, cross-entropy on the roots, backward "
                         "(gigl_gather_reduce_backward + the projections' backward GEMMs) and the Adam update — the loop of "
                         "NodeClassificationModelingTaskSpec._train; a secondary line with its own roofline / cpu_baseline")
    ap.add_argument("--gat-train-autograd", action="store_true",
                    help="--workload gat-lp --train: the autograd-driven step (round 4's line) instead of the library plan "
                         "(gigl_gat_nablp_train_plan_*)")
    ap.add_argument("--no-train-prefetch", action="store_true",
                    help="--train --train-task lp: every step samples its own batch (A/B of the next batch's graph part beside "
                         "this step's layers)")
    ap.add_argument("--train-task", type=str, default="snc", choices=["snc", "lp"],
                    help="--train: snc = node classification (gigl_sage_train_plan_*); lp = the link-prediction step of the "
                         "reference's default trainer (GraphSAGE encoder, Retrieval task) as ONE library call "
                         "(gigl_nablp_train_plan_*): --batch anchors (default 2048) with one positive each + 512 random "
                         "negatives per step")
    ap.add_argument("--entry", type=str, default="plan", choices=["plan", "inferencer", "sampler"],
                    help="plan = the library's one-call plan driven by this script (the headline); inferencer = the same "
                         "workload through the drop-in entry point's own loop (gigl_amd.inferencer.Inferencer."
                         "infer_resident -> plugin.infer_batch -> in-HBM route -> Avro shards): one step = one batch of "
                         "the full inference pass over every node; sampler = the Subgraph Sampler job's step (S3-S9): "
                         "k-hop sample of a batch of roots + its RootedNodeNeighborhood TFRecords encoded on the device "
                         "(gigl_sample_khop + gigl_records_encode), records left in HBM; --batch defaults to the job's "
                         "4096 roots")
    ap.add_argument("--entry-sink", type=str, default="avro-device", choices=["avro-device", "avro-files", "none"],
                    help="--entry inferencer: avro-device (the line's value) = rows encoded as Avro data blocks on the "
                         "device, the blocks stay in HBM (outputs resident, like the inputs); avro-files = additionally "
                         "copied out and appended to shard files in a tmpfs scratch directory by the exporter's writer "
                         "thread (PCIe + file inclusive; always measured and reported next to the value); none = bare rows")
    args = ap.parse_args()
    if args.shard_group is None:
        args.shard_group = 16 if getattr(args, "emulate_world", 0) and args.emulate_world > 1 else 32
    wl_fan, wl_b = WORKLOAD_DEFAULTS.get(args.workload, ("25,10", 1024))
    args.fanouts = args.fanouts or wl_fan
    if not args.batch and args.entry == "sampler":
        # the sampler job's own call size (gigl_amd.subgraph_sampler.sampler_call_size: up to 32,768 roots per call)
        from gigl_amd.subgraph_sampler import sampler_call_size
        wl = WORKLOADS.get(args.workload)
        args.batch = sampler_call_size(1 << 30, [int(v) for v in args.fanouts.split(",")], int(wl[3]) if wl else 100)
    args.batch = args.batch or wl_b

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # no launcher: this process becomes one
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GIGL_BENCH_SHARE_GPU") == "1":
        local_rank = 0  # (functional check on a one-GPU box, also under a launcher that numbers the ranks' devices)
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node equal to --gpus "
              "(or without a launcher: bench.py spawns the ranks itself)", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if dist_backend() == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    if args.train and args.workload == "gat-lp":
        if args.gat_train_autograd:
            return run_gat_lp_train(args, rank, world, local_rank)
        return run_gat_lp_train_plan(args, rank, world, local_rank)
    if args.train and args.train_task == "lp":
        return run_lp_train(args, rank, world, local_rank)
    if args.train:
        return run_train(args, rank, world, local_rank)
    if args.entry == "inferencer":
        return run_entry_inferencer(args, rank, world, local_rank)
    if args.entry == "sampler":
        return run_entry_sampler(args, rank, world, local_rank)
    if args.workload == "mag240m-sharded" and args.emulate_world > 1:
        if world != 1:
            print("bench.py: --emulate-world runs in one process on one GPU", file=sys.stderr)
            sys.exit(2)
        return run_emulated_world(args, local_rank)
    if args.workload == "mag240m-sharded":
        return run_sharded(args, rank, world, local_rank)
    if args.workload == "gat-lp":
        return run_gat_lp(args, rank, world, local_rank)
    if args.workload == "typed-dblp":
        return run_typed(args, rank, world, local_rank)
    return run_products(args, rank, world, local_rank)
