#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06p; mkdir -p $o
prof() {  # prof <name> <bench args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -f csv -d $o/prof_$name -o $name -- python bench.py "$@" > $o/prof_$name.log 2>&1
  grep '^{' $o/prof_$name.log | tail -1 > $o/bench_${name}_under_rocprof.json
  local f=$(find $o/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $o/kernel_stats_$name.csv
  find $o/prof_$name -type f -size +8M -delete
}
prof emulated_w8_peer_all --workload mag240m-sharded --emulate-world 8 --shard-route peer-all --emulate-streams 0 --steps 256
python scripts/emulated_kernel_time.py $o/kernel_stats_emulated_w8_peer_all.csv 16 > $o/emulated_world8_kernel_time_peer_all.txt
head -36 $o/emulated_world8_kernel_time_peer_all.txt
