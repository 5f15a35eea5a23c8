#!/bin/bash
# LP training plan: weight-grad partial sums inside lp_adam, one W^T per step
mkdir -p gpurun_out/r06u
timeout 900 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_overflow.py tests/test_gpu_nablp.py -x -q > gpurun_out/r06u/tests.log 2>&1
tail -5 gpurun_out/r06u/tests.log
for i in 1 2; do
timeout 300 python bench.py --train --train-task lp > gpurun_out/r06u/lp_fused_$i.json 2> gpurun_out/r06u/lp_fused_$i.err
GIGL_TRAIN_PLAN_UNFUSED=1 timeout 300 python bench.py --train --train-task lp > gpurun_out/r06u/lp_unfused_$i.json 2> gpurun_out/r06u/lp_unfused_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06u/lp_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j['value'], j['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
