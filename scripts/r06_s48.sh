#!/bin/bash
# every projection-kernel variant still passes the fused-plan tests (A/B knobs stay usable)
mkdir -p gpurun_out/r06az
for v in 0 4 7 8 9; do
  GIGL_F2_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_plan.py -x -q -k fused > gpurun_out/r06az/tests_v$v.log 2>&1
  echo "variant $v: $(grep -E 'passed|failed' gpurun_out/r06az/tests_v$v.log | tail -1)"
done
GIGL_F2_ALL_WR=1 timeout 600 python -m pytest tests/test_gpu_plan.py -x -q -k fused > gpurun_out/r06az/tests_allwr.log 2>&1; echo "all_wr: $(grep -E 'passed|failed' gpurun_out/r06az/tests_allwr.log | tail -1)"
GIGL_EXPAND_ITERS=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r06az/tests_iters0.log 2>&1; echo "iters0: $(grep -E 'passed|failed' gpurun_out/r06az/tests_iters0.log | tail -1)"
GIGL_TRAIN_PLAN_UNFUSED=1 timeout 900 python -m pytest tests/test_gpu_train_plan.py -x -q > gpurun_out/r06az/tests_unfused.log 2>&1; echo "unfused: $(grep -E 'passed|failed' gpurun_out/r06az/tests_unfused.log | tail -1)"
