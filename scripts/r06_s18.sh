#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06r; mkdir -p $o
rocprofv3 --kernel-trace --stats -f csv -d $o/prof_train -o train -- python bench.py --train --no-cpu-baseline --steps 512 --warmup 32 > $o/prof_train.log 2>&1
f=$(find $o/prof_train -name '*kernel_stats.csv' | head -1); cp $f $o/kernel_stats_train_fused.csv
find $o/prof_train -type f -size +8M -delete
python scripts/kstats.py $o/kernel_stats_train_fused.csv "" | head -34
