set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02y
# 1. driver-flag bench line (with cpu baseline)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02y/bench_driver_flags.json 2> gpurun_out/r02y/bench_driver_flags.err
# 2. kernel stats under rocprof
scripts/gpu_profile.sh r02y --steps 20 --warmup 5 > gpurun_out/r02y/profile.log 2>&1
f=$(find gpurun_out/prof_r02y -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/r02y/kernel_stats.csv
cp gpurun_out/bench_r02y.json gpurun_out/r02y/bench_under_rocprof.json
t=$(find gpurun_out/prof_r02y -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python scripts/trace_summary.py $t > gpurun_out/r02y/trace_by_grid.txt 2>&1; [ -n "$t" ] && python scripts/overlap.py $t > gpurun_out/r02y/overlap.txt 2>&1
# 3. PMC
scripts/gpu_pmc.sh r02y --no-cpu-baseline > gpurun_out/r02y/pmc.log 2>&1
cp gpurun_out/pmc_r02y.json gpurun_out/r02y/pmc.json
# 4. SQ for expand
scripts/gpu_sq.sh r02y "expand_rows_kernel|plan_rows_kernel|gather_mean_kernel|linear_split_kernel" --no-cpu-baseline > gpurun_out/r02y/sq.log 2>&1
cp gpurun_out/sq_r02y.json gpurun_out/r02y/sq.json
ls -la gpurun_out/r02y
