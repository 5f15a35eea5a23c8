#!/bin/bash
# round-5 session 16 (GPU box): training batches as root groups of the union pass — tests, then bench lines per group count
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05n; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_dist_plan.py -x -q -m gpu -k "train or library or transposed or refuse" > $o/pytest_plan.log 2>&1
tail -5 $o/pytest_plan.log
for g in 1 8 32 64 128; do
  GIGL_TRAIN_GROUPS=$g timeout 600 python bench.py --train --steps 64 --warmup 8 2> $o/bench_train_g$g.err | grep '^{' | tail -1 > $o/bench_train_g$g.json
  GIGL_TRAIN_GROUPS=$g timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 2> $o/bench_lp_train_g$g.err | grep '^{' | tail -1 > $o/bench_lp_train_g$g.json
  python - <<P
import json
for n in ("train", "lp_train"):
    try:
        d = json.load(open("$o/bench_%s_g$g.json" % n))
        print("groups $g", n, round(d["ms_per_step"], 4), "ms/step", round(d["value"] / 1e9, 4), "G", d["config"].get("loss_last_step"))
    except Exception as e:
        print("groups $g", n, "no line", e)
P
done
