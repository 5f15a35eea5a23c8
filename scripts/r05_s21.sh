#!/bin/bash
# round-5 session 21 (GPU box): the GAT link-prediction plan behind the trainer (tests), its bench line and kernel stats
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05t; mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_nablp.py tests/test_gpu_entry_points.py -x -q -m gpu > $o/pytest.log 2>&1
tail -5 $o/pytest.log
timeout 900 python bench.py --workload gat-lp --train --steps 32 --warmup 8 2> $o/bench_gat-lp_train.err | grep '^{' | tail -1 > $o/bench_gat-lp_train.json
tail -3 $o/bench_gat-lp_train.err
python - <<P
import json
try:
    d = json.load(open("$o/bench_gat-lp_train.json"))
    print("gat-lp train", round(d["ms_per_step"], 4), "ms/step", round(d["value"] / 1e9, 4), "G", d["config"]["loss_first_step"], d["config"]["loss_last_step"], "autograd", d["config"]["autograd_driven_ms_per_step"])
except Exception as e:
    print("no line", e)
P
rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o gat -- python bench.py --workload gat-lp --train --steps 32 --warmup 8 > $o/prof.log 2>&1
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_gat-lp_train.csv
find $o/prof -type f -size +8M -delete
python scripts/kstats.py $o/kernel_stats_gat-lp_train.csv "" | head -32
