#!/bin/bash
# round-5 session 10 (GPU box): the link-prediction training plan after the d_cand fix
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05i; mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_train_plan.py -x -q -m gpu > $o/pytest_plan.log 2>&1
tail -8 $o/pytest_plan.log
timeout 2400 python -m pytest tests/test_gpu_nablp.py tests/test_gpu_entry_points.py tests/test_gpu_trainer_ddp.py -x -q -m gpu > $o/pytest_nablp.log 2>&1
tail -8 $o/pytest_nablp.log
timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 > $o/bench_lp_train.json 2> $o/bench_lp_train.err
tail -3 $o/bench_lp_train.err; head -c 900 $o/bench_lp_train.json; echo
rocprofv3 --kernel-trace --stats -f csv -d $o/prof_lp -o lp -- python bench.py --train --train-task lp --steps 64 --warmup 8 > $o/prof_lp.log 2>&1
f=$(find $o/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_lp_train.csv
find $o/prof_lp -type f -size +8M -delete
python scripts/kstats.py $o/kernel_stats_lp_train.csv "" | head -45
