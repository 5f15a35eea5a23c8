#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06k; mkdir -p $o
for g in 16 32; do
  t0=$(date +%s); timeout 600 python bench.py --workload mag240m-sharded --emulate-world 8 --shard-scale 0.08 --fanouts 25,10 --batch 1024 --shard-group $g --steps 256 --no-cpu-baseline --no-live-pmc > $o/sub_g$g.json 2> $o/sub_g$g.err
  echo "G=$g child wall $(( $(date +%s) - t0 )) s"
  python - $o/sub_g$g.json <<PY
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1], round(j["value"]/1e9,2), "G", round(j["ms_per_step"]*1e3,2), "us", j.get("route"))
for t,e in j["emulated"].items():
    if isinstance(e,dict) and e.get("overlapped"): print("  ", t, round(e["kernel_ms_per_step_per_rank"]*1e3,1), round(e["overlapped"]["ms_per_rank_step"]*1e3,1))
PY
done
t0=$(date +%s); python bench.py --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "default bench wall $(( $(date +%s) - t0 )) s"
