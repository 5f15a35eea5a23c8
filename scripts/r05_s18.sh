#!/bin/bash
# round-5 session 18 (GPU box): the link-prediction plan's second workspace (next batch's graph part beside this step's
# layers): tests, A/B bench lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05q; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_nablp.py -x -q -m gpu > $o/pytest_plan.log 2>&1
tail -5 $o/pytest_plan.log
timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 2> $o/bench_lp_train.err | grep '^{' | tail -1 > $o/bench_lp_train.json
timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 --no-train-prefetch 2> $o/bench_lp_train_noprefetch.err | grep '^{' | tail -1 > $o/bench_lp_train_noprefetch.json
python - <<P
import json
for n in ("lp_train", "lp_train_noprefetch"):
    try:
        d = json.load(open("$o/bench_%s.json" % n))
        print(n, round(d["ms_per_step"], 4), "ms/step", round(d["value"] / 1e9, 4), "G", d["config"].get("loss_last_step"), d["config"].get("autograd_driven_ms_per_step"))
    except Exception as e:
        print(n, "no line", e)
P
tail -3 $o/bench_lp_train.err
