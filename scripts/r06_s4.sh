#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06d; mkdir -p $o
timeout 2400 python -m pytest tests/test_gpu_overflow.py -x -q -m gpu 2>&1 | tail -40
timeout 2400 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_hbm_route.py tests/test_gpu_plan.py tests/test_bench_launcher.py -x -q -m gpu 2>&1 | tail -15
