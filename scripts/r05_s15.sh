#!/bin/bash
# round-5 session 15 (GPU box): the transposed backward gather in the training plans — tests, then A/B bench lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05m; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_train_plan.py -x -q -m gpu > $o/pytest_plan.log 2>&1
tail -5 $o/pytest_plan.log
for v in gather atomic; do
  ev=""; [ $v = atomic ] && ev="GIGL_TRAIN_BWD_ATOMIC=1"
  env $ev timeout 600 python bench.py --train --steps 64 --warmup 8 2> $o/bench_train_$v.err | grep '^{' | tail -1 > $o/bench_train_$v.json
  env $ev timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 2> $o/bench_lp_train_$v.err | grep '^{' | tail -1 > $o/bench_lp_train_$v.json
  python - <<P
import json
for n in ("train", "lp_train"):
    try:
        d = json.load(open("$o/bench_%s_$v.json" % n))
        print("$v", n, round(d["ms_per_step"], 4), "ms/step", round(d["value"] / 1e9, 4), "G")
    except Exception as e:
        print("$v", n, "no line", e)
P
done
