#!/bin/bash
# GAT link-prediction plan: the random negatives' encode on a stream of its own (default; GIGL_LP_FORK=0 off)
mkdir -p gpurun_out/r06bg
for i in 1 2; do
timeout 1200 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_overflow.py tests/test_gpu_nablp.py -q > gpurun_out/r06bg/tests_$i.log 2>&1
grep -E "passed|failed" gpurun_out/r06bg/tests_$i.log | tail -1; grep -E "^FAILED" gpurun_out/r06bg/tests_$i.log | head -5
done
for v in 1 0 1 0; do
  if [ $v = 0 ]; then export GIGL_LP_FORK=0; else unset GIGL_LP_FORK; fi
  timeout 600 python bench.py --workload gat-lp --train > gpurun_out/r06bg/bench_fork${v}_$RANDOM.json 2> gpurun_out/r06bg/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06bg/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j['value']/1e9, j['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
