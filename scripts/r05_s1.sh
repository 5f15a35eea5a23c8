#!/bin/bash
# round-5 session 1 (GPU box): where the sharded MAG240M step's time goes.
#   bench lines of mag240m-sharded at 16 / 32 / 64 batches per exchange, a rocprofv3 kernel trace of the world-1 step
#   and of the emulated 8-rank world (kernel time per rank-step; how much of the span has no kernel resident)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05a; mkdir -p $o
for g in 16 32 64; do
  timeout 600 python bench.py --workload mag240m-sharded --no-cpu-baseline --steps 20 --warmup 5 --shard-group $g \
    > $o/bench_sharded_g$g.json 2> $o/bench_sharded_g$g.err
done
prof() {  # prof <name> <bench args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -f csv -d $o/prof_$name -o $name -- python bench.py "$@" > $o/prof_$name.log 2>&1
  grep '^{' $o/prof_$name.log | tail -1 > $o/bench_${name}_under_rocprof.json
  local t=$(find $o/prof_$name -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python scripts/overlap.py $t > $o/overlap_$name.txt 2>&1
  local f=$(find $o/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $o/kernel_stats_$name.csv
  find $o/prof_$name -type f -size +8M -delete
}
prof sharded_w1 --workload mag240m-sharded --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 1.0
prof emulated_w8 --workload mag240m-sharded --emulate-world 8 --no-cpu-baseline --steps 256
ls -la $o
