#!/bin/bash
# round-5 session 24 (GPU box): transposed lists built in the graph parts — tests and bench lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05z; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_nablp.py -x -q -m gpu > $o/pytest_plan.log 2>&1
grep -a "passed\|failed" $o/pytest_plan.log | tail -2
b() { local name=$1; shift; timeout 900 python bench.py "$@" 2> $o/bench_$name.err | grep '^{' | tail -1 > $o/bench_$name.json; }
b train --train --steps 64 --warmup 8
b lp_train --train --train-task lp --steps 64 --warmup 8
python - <<P
import json
for n in ("train", "lp_train"):
    try:
        d = json.load(open("$o/bench_%s.json" % n))
        print(n, round(d["ms_per_step"], 4), "ms/step", round(d["value"] / 1e9, 4), "G")
    except Exception as e:
        print(n, "no line", e)
P
