#!/bin/bash
# round-5 session 7 (GPU box): link-prediction trainer at world 2 (replica vs sharded), then the products profile round
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05g; mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_trainer_ddp.py tests/test_gpu_nablp.py tests/test_gpu_dist_plan.py -x -q -m gpu > $o/pytest_lp.log 2>&1
tail -12 $o/pytest_lp.log
bash scripts/prof_round.sh r05g products > $o/prof_round.log 2>&1
tail -5 $o/prof_round.log
ls gpurun_out/r05g
