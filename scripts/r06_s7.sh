#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06g; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_nablp.py tests/test_gpu_train_plan.py -x -q -m gpu -s 2>&1 | grep -a "route vs\|passed\|failed\|Error\|assert" | cut -c1-900 | tail -12
prof() {  # prof <name> <bench args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -f csv -d $o/prof_$name -o $name -- python bench.py "$@" > $o/prof_$name.log 2>&1
  grep '^{' $o/prof_$name.log | tail -1 > $o/bench_${name}_under_rocprof.json
  local f=$(find $o/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $o/kernel_stats_$name.csv
  find $o/prof_$name -type f -size +8M -delete
  python scripts/kstats.py $o/kernel_stats_$name.csv "" | head -${KS:-28}
}
prof train --train --no-cpu-baseline --steps 512 --warmup 32
prof gatlp_train --workload gat-lp --train --no-cpu-baseline --steps 64 --warmup 8
