#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06j; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_attn.py tests/test_gpu_train_plan.py -x -q -m gpu -s 2>&1 | grep -a "GAT plan\|passed\|failed\|Error\|assert" | cut -c1-400 | tail -12
for tw in 0 1; do
  if [ $tw = 1 ]; then export GIGL_GAT_BWD_TWO_SWEEPS=1; else unset GIGL_GAT_BWD_TWO_SWEEPS; fi
  timeout 600 python bench.py --workload gat-lp --train --no-cpu-baseline --steps 64 --warmup 8 > $o/gatlp_train_two_sweeps_$tw.json 2> $o/e.err
  python - $o/gatlp_train_two_sweeps_$tw.json <<PY
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], j["value"], j["ms_per_step"])
PY
done
