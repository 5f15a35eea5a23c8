#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06q; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_train_plan.py -x -q -m gpu -s 2>&1 | grep -a "node-classification plan\|passed\|failed\|Error\|assert" | cut -c1-300 | tail -12
for uf in 0 1; do
  if [ $uf = 1 ]; then export GIGL_TRAIN_PLAN_UNFUSED=1; else unset GIGL_TRAIN_PLAN_UNFUSED; fi
  for rep in 1; do
  timeout 600 python bench.py --train --no-cpu-baseline --no-live-pmc > $o/train_unfused_${uf}_$rep.json 2> $o/e.err
  python - $o/train_unfused_${uf}_$rep.json <<PY
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(j["value"]/1e9,3), "G", round(j["ms_per_step"],4), "ms")
PY
  done
done
