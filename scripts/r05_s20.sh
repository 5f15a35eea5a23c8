#!/bin/bash
# round-5 session 20 (GPU box): the GAT link-prediction training plan against autograd
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05s; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_train_plan.py -x -q -m gpu -k "gat_link" -s > $o/pytest_gat_plan.log 2>&1
tail -40 $o/pytest_gat_plan.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
