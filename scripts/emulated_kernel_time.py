"""Kernel time per rank-step of `bench.py --workload mag240m-sharded --emulate-world W` from the rocprofv3 kernel stats of
that run (profiles/r05*_kernel_stats_emulated_world8.csv): every library kernel's total duration / (plan calls x batches
per exchange), grouped — kernels only the SHARDED step has (bucketing, claiming, the owners' row gather, scatter-back),
kernels it shares with the fused single-GPU plan, the in-process transport's device copies (stand-ins for RCCL: not a
rank's kernels) and the bench's setup (graph / table generation, the per-rank table projection).
usage: python scripts/emulated_kernel_time.py <kernel_stats.csv> <batches per exchange> [edges per rank-step]"""
import csv
import json
import re
import sys

SHARDED_ONLY = ("dist_clear_kernel", "bucket_kernel", "claim_bucket_kernel", "scatter_slots_kernel", "pos_from_map_kernel",
                "serve_rows_copy_kernel", "serve_rows_wave_kernel", "fold_overflow_kernel", "hot_unmark_kernel")
SHARED = ("expand_rows_kernel", "plan_rows_kernel", "work_counts_zero_kernel", "expand_heavy", "lg3_", "lg2_", "row_sort",
          "gather_mean_kernel", "linear_split_kernel<2, true, true", "gigl_take_rows_kernel", "find_heavy", "huge_row")
TRANSPORT = ("__amd_rocclr_copyBuffer",)

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*$", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.append((n, int(r["Calls"]), float(r["TotalDurationNs"])))
G = int(sys.argv[2])
calls = [c for n, c, _ in rows if n.startswith("lg3_dedup")][0]
steps = calls * G
grp = {"sharded_only": {}, "shared_with_the_fused_plan": {}, "transport_stand_in": {}, "setup_and_bench": {}}
for n, c, t in rows:
    us = t / 1e3 / steps
    if any(n.startswith(k) for k in SHARDED_ONLY):
        g = "sharded_only"
    elif any(n.startswith(k) or k in n for k in SHARED):
        g = "shared_with_the_fused_plan"
    elif any(n.startswith(k) for k in TRANSPORT):
        g = "transport_stand_in"
    else:
        g = "setup_and_bench"
    key = n[:60]
    grp[g][key] = round(grp[g].get(key, 0.0) + us, 3)
rank = sum(grp["sharded_only"].values()) + sum(grp["shared_with_the_fused_plan"].values())
out = {"plan_calls": calls, "rank_steps": steps,
       "kernel_us_per_rank_step": round(rank, 2),
       "sharded_only_us_per_rank_step": round(sum(grp["sharded_only"].values()), 2),
       "sharded_only_share": round(sum(grp["sharded_only"].values()) / rank, 4),
       "groups": {g: dict(sorted(v.items(), key=lambda kv: -kv[1])) for g, v in grp.items()}}
if len(sys.argv) > 3:
    e = float(sys.argv[3])
    out["projection"] = {"label": "PROJECTION: 8 ranks x edges per rank-step / kernel time per rank-step (no overlap between a "
                                  "rank's kernels, exchanges hidden behind the other plans in flight)",
                         "edges_per_rank_step": e, "whole_node_edges_per_s": round(8 * e / (rank * 1e-6))}
print(json.dumps(out, indent=1))
