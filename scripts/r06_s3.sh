#!/bin/bash
# round-6 session 3: lone-rank hops, dist tests, emulated world sweeps (G, streams) on the peer route, rocprof of both routes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06c; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_dist_peer.py tests/test_gpu_dist_plan.py tests/test_gpu_dist.py "tests/test_gpu_entry_points.py::test_bench_emulated_world_line" tests/test_bench_launcher.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --workload mag240m-sharded > $o/bench_mag240m-sharded.json 2> $o/sh.err; tail -2 $o/sh.err; head -c 300 $o/bench_mag240m-sharded.json; echo
show() { python - "$1" <<PY
import json,sys
j=json.load(open(sys.argv[1]))
print(sys.argv[1], j["value"], j["ms_per_step"], j.get("route"))
for t,e in j["emulated"].items():
    if not isinstance(e,dict) or "route" not in e: continue
    print(" ", t, "kernel_ms", round(e["kernel_ms_per_step_per_rank"],5), "share", e["sharded_only_kernel_share"], "overl", e["overlapped"] and e["overlapped"]["ms_per_rank_step_all"], "pulled", round(e["pulled_rows_per_step_mean"]), "link_ms", round(e["projection"]["link_ms_per_step"],5))
    print("     ", e["kernel_ms_by_group"])
PY
}
for g in 16 32; do for st in 3 4; do
  timeout 600 python bench.py --workload mag240m-sharded --emulate-world 8 --shard-route peer --shard-group $g --emulate-streams $st > $o/emu_peer_g${g}_s${st}.json 2> $o/emu.err || tail -3 $o/emu.err
  show $o/emu_peer_g${g}_s${st}.json
done; done
prof() {  # prof <name> <bench args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -f csv -d $o/prof_$name -o $name -- python bench.py "$@" > $o/prof_$name.log 2>&1
  grep '^{' $o/prof_$name.log | tail -1 > $o/bench_${name}_under_rocprof.json
  local f=$(find $o/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $o/kernel_stats_$name.csv
  find $o/prof_$name -type f -size +8M -delete
}
prof emulated_w8_peer --workload mag240m-sharded --emulate-world 8 --shard-route peer --emulate-streams 0 --steps 256
python scripts/emulated_kernel_time.py $o/kernel_stats_emulated_w8_peer.csv 16 > $o/emulated_world8_kernel_time_peer.txt
head -40 $o/emulated_world8_kernel_time_peer.txt
prof emulated_w8_bucketed --workload mag240m-sharded --emulate-world 8 --shard-route bucketed --emulate-streams 0 --steps 256
python scripts/emulated_kernel_time.py $o/kernel_stats_emulated_w8_bucketed.csv 16 > $o/emulated_world8_kernel_time_bucketed.txt
head -24 $o/emulated_world8_kernel_time_bucketed.txt
