#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06f; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_groups.py tests/test_gpu_train_plan.py tests/test_gpu_overflow.py -x -q -m gpu -s 2>&1 | grep -a "Adam state\|passed\|failed\|Error\|error\|assert" | tail -30
line() { python - "$1" <<PY
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print(sys.argv[1], round(j["value"]/1e9,3), "G", round(j["ms_per_step"]*1e3,2), "us", j.get("steps"), j.get("steps_honoured"))
else: print(sys.argv[1], "NO LINE")
PY
}
for gp in off high same; do
  timeout 600 python bench.py --graph-priority $gp --no-cpu-baseline --no-live-pmc --no-emulated-sub > $o/bench_gp_$gp.json 2> $o/bench_gp_$gp.err; line $o/bench_gp_$gp.json
done
for gp in off high; do
  timeout 600 python bench.py --workload mag-shard --graph-priority $gp --no-cpu-baseline --no-live-pmc > $o/bench_mag_gp_$gp.json 2> $o/e.err; line $o/bench_mag_gp_$gp.json
done
timeout 900 python bench.py --entry inferencer --no-cpu-baseline > $o/bench_entry_inferencer.json 2> $o/e2.err; line $o/bench_entry_inferencer.json
