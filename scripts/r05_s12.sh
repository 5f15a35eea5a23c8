#!/bin/bash
# round-5 session 12 (GPU box): the whole -m gpu suite on the tree as committed, then the link-prediction training line
# with the autograd-driven comparison
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05j; mkdir -p $o
timeout 3300 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1
tail -5 $o/pytest_gpu.log
timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 > $o/bench_lp_train.json 2> $o/bench_lp_train.err
tail -3 $o/bench_lp_train.err; cat $o/bench_lp_train.json
