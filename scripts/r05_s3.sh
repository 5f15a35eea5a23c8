#!/bin/bash
# round-5 session 3 (GPU box): the fused two-layer projection — parity, then the headline with / without it
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05c; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_plan.py -x -q -m gpu -s > $o/pytest_plan.log 2>&1
tail -25 $o/pytest_plan.log
timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 > $o/bench_products_fused.json 2> $o/bench_products_fused.err
GIGL_PLAN_NO_FUSE2=1 timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 > $o/bench_products_apart.json 2> $o/bench_products_apart.err
timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 --streams 1 > $o/bench_products_fused_s1.json 2> $o/bench_products_fused_s1.err
tail -3 $o/*.err
