#!/bin/bash
# round-5 session 2 (GPU box): the sharded step after the one-clear / self-in-place / tiled-layers changes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05b; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_dist_plan.py tests/test_gpu_dist.py tests/test_gpu_hbm_route.py tests/test_gpu_trainer_ddp.py tests/test_bench_launcher.py -x -q -m gpu > $o/pytest_dist.log 2>&1
tail -15 $o/pytest_dist.log
for g in 16 32 64; do
  timeout 600 python bench.py --workload mag240m-sharded --no-cpu-baseline --steps 20 --warmup 5 --shard-group $g \
    > $o/bench_sharded_g$g.json 2> $o/bench_sharded_g$g.err
done
timeout 600 python bench.py --workload mag-shard --no-cpu-baseline --no-live-pmc --steps 20 --warmup 5 > $o/bench_mag-shard.json 2> $o/bench_mag-shard.err
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
prof sharded_w1 --workload mag240m-sharded --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 1.0 --shard-group 32
prof emulated_w8 --workload mag240m-sharded --emulate-world 8 --no-cpu-baseline --steps 256
ls -la $o
