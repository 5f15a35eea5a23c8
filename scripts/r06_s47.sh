#!/bin/bash
# GAT link-prediction training plan: weight-gradient partial sums inside the Adam kernel, W^T once per step
mkdir -p gpurun_out/r06ay
timeout 1200 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_overflow.py tests/test_gpu_nablp.py tests/test_gpu_attn.py -x -q > gpurun_out/r06ay/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r06ay/tests.log | tail -1
grep -E "^FAILED|Error" gpurun_out/r06ay/tests.log | head -5
for v in 0 1; do
  if [ $v = 1 ]; then export GIGL_TRAIN_PLAN_UNFUSED=1; else unset GIGL_TRAIN_PLAN_UNFUSED; fi
  timeout 600 python bench.py --workload gat-lp --train > gpurun_out/r06ay/bench_unfused${v}_$RANDOM.json 2> gpurun_out/r06ay/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ay/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j['value']/1e9, j['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
