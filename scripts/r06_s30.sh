#!/bin/bash
# round-6 session 30: full -m gpu suite + smoke on the tree with linear_fused2x_kernel, then the products profile set
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06ag
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06ag/pytest_gpu.log 2>&1
echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r06ag/pytest_gpu_tail.txt
tail -4 gpurun_out/r06ag/pytest_gpu.log | tee -a gpurun_out/r06ag/pytest_gpu_tail.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06ag/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06ag/smoke.log
bash scripts/prof_round.sh r06ag products "linear_fused2x_kernel|gather_mean_kernel|lg3_dedup_kernel" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06ag/bench_products_driver_flags.json 2> gpurun_out/r06ag/bench_df.err
tail -2 gpurun_out/r06ag/bench_df.err; head -c 300 gpurun_out/r06ag/bench_products_driver_flags.json; echo
