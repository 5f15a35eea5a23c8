#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05i; mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_train_plan.py -x -q -m gpu > $o/pytest_plan.log 2>&1
tail -4 $o/pytest_plan.log
timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 > $o/bench_lp_train.json 2> $o/bench_lp_train.err
tail -2 $o/bench_lp_train.err
rocprofv3 --kernel-trace --stats -f csv -d $o/prof_lp -o lp -- python bench.py --train --train-task lp --steps 64 --warmup 8 > $o/prof_lp.log 2>&1
f=$(find $o/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_lp_train.csv
find $o/prof_lp -type f -size +8M -delete
python scripts/kstats.py $o/kernel_stats_lp_train.csv "" | head -45
