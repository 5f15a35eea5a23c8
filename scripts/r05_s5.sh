#!/bin/bash
# round-5 session 5 (GPU box): the sharded-only kernels after the serve / claim changes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05e; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_dist_plan.py tests/test_gpu_dist.py tests/test_gpu_hbm_route.py "tests/test_gpu_entry_points.py::test_bench_emulated_world_line" tests/test_bench_launcher.py -x -q -m gpu > $o/pytest_dist.log 2>&1
tail -8 $o/pytest_dist.log
timeout 600 python bench.py --workload mag240m-sharded --no-cpu-baseline --steps 20 --warmup 5 --shard-group 32 > $o/bench_sharded_g32.json 2> $o/bench_sharded_g32.err
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
prof emulated_w8 --workload mag240m-sharded --emulate-world 8 --no-cpu-baseline --steps 256
python scripts/emulated_kernel_time.py $o/kernel_stats_emulated_w8.csv 16 > $o/emulated_world8_kernel_time.txt
head -30 $o/emulated_world8_kernel_time.txt
timeout 900 python bench.py --workload mag240m-sharded --emulate-world 8 --no-cpu-baseline --steps 256 > $o/bench_emulated_w8.json 2> $o/bench_emulated_w8.err
tail -2 $o/*.err
