#!/usr/bin/env python3
"""bench.py — sampled+aggregated edges/s of the HIP hot path (BASELINE.json metric).

A "step" = one batch of B roots through the whole path, inputs already resident in HBM:
    k-hop sample (parity mode, Spark-hash permutation)  ->  batch union graph (dedup + CSR)
    ->  GraphSAGE forward (gather-mean + fp32 MFMA projection per layer, trimmed schedule)
    ->  one embedding row per root
enqueued by ONE library call (gigl_sage_plan_run).  Workload at N=1 = BASELINE.json configs[1]:
ogbn-products-SHAPED synthetic graph (N=2,449,029, RMAT(.57,.19,.19) power-law, ~61.9M undirected
pairs bidirectionalised, D=100 fp32), fanout [25,10], B=1024, GraphSAGE 100->256->47 (SURVEY.md
§8(d) C2).  MAG240M (the config the metric is quoted on) does not fit one GPU (375 GB of features),
so per the contract the N=1 line is the largest single-GPU configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--group G] [--workload W]

Steps are independent batches.  One library call takes G consecutive batches through ONE set of ~20 launches
(gigl_sage_plan_set_groups: every batch keeps its own union graph, rows are bit-identical to G single-batch
calls — tests/test_gpu_groups.py); calls are pipelined over S HIP streams (one library ctx + one host thread per
stream, all sharing the HBM-resident graph and hash table; defaults S = 3, G = 64 — a sweep of S in 2..6, G in
16..128 stays within 7 % of the best).  The regime does not depend on --steps.  A --steps that is a whole number of rounds
(S*G = 192 steps by default: 192, 384, 960, ...) is honoured exactly: one timed repetition = K steps, `steps` = K,
`steps_honoured`: true; any other value is rounded up to whole rounds (>= --min-rounds) and the line carries the rounded
`steps` with `steps_honoured`: false (likewise --warmup / `warmup_honoured`).  The repetition is repeated until the timed
region lasts >= --min-seconds; median / p10 / p90 over the repetitions are reported next to the aggregate.
N>1: one process per GPU (torch.distributed, RCCL); every rank holds a replica of the graph and takes
its own root batches — the path shards by roots with no data-path collective ("weak" scaling);
time = max over ranks, value = total edges of all ranks / that time.

Counting: sampled edge = one (src->dst) pair emitted by a hop expansion before batch
dedup; aggregated edge = one edge actually consumed by one layer's segmented reduce (sum_l |E_l|,
trimmed schedule — never the inflated L*|E_union|, which is reported in config for context).
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench.cli import main  # noqa: E402  (the package bench/ beside this file: one module per workload family)

if __name__ == "__main__":
    main()
