#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06n; mkdir -p $o
timeout 900 python -m pytest "tests/test_gpu_entry_points.py::test_bench_emulated_world_line" -x -q -m gpu 2>&1 | tail -5
t0=$(date +%s); timeout 600 python bench.py --workload mag240m-sharded --emulate-world 8 --shard-scale 0.08 --fanouts 25,10 --batch 1024 --shard-group 32 --steps 256 --no-cpu-baseline --no-live-pmc > $o/sub_g32.json 2> $o/sub.err; echo "child wall $(( $(date +%s) - t0 )) s"; tail -2 $o/sub.err
python - <<PY
import json
j=json.load(open("$o/sub_g32.json")); print(round(j["value"]/1e9,2), "G", j["route"]); print(j["world1_reference"])
for t,e in j["emulated"].items():
    if isinstance(e,dict) and e.get("overlapped"): print(t, {k:round(v,2) for k,v in e["projection"].items() if k.startswith("scaling")})
PY
