#!/bin/bash
# round-6 session 38: full -m gpu suite + smoke + driver-flag bench on the tree with retiring expand waves
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06ao
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06ao/pytest_gpu.log 2>&1
echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r06ao/pytest_gpu_tail.txt
grep -E "passed|failed" gpurun_out/r06ao/pytest_gpu.log | tail -2 | tee -a gpurun_out/r06ao/pytest_gpu_tail.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06ao/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06ao/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06ao/bench_products_driver_flags.json 2> gpurun_out/r06ao/bench_df.err
tail -2 gpurun_out/r06ao/bench_df.err; head -c 300 gpurun_out/r06ao/bench_products_driver_flags.json; echo
for wl in mag240m-sharded mag-shard; do
timeout 900 python bench.py --workload $wl --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-emulated-sub > gpurun_out/r06ao/bench_$wl.json 2> gpurun_out/r06ao/bench_$wl.err
head -c 250 gpurun_out/r06ao/bench_$wl.json; echo
done
