set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
# 1. driver-flag bench line (with cpu baseline)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02p/bench_driver_flags.json 2> gpurun_out/r02p/bench_driver_flags.err
# 2. kernel stats under rocprof
scripts/gpu_profile.sh r02p --steps 20 --warmup 5 > gpurun_out/r02p/profile.log 2>&1
f=$(find gpurun_out/prof_r02p -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/r02p/kernel_stats.csv
cp gpurun_out/bench_r02p.json gpurun_out/r02p/bench_under_rocprof.json
t=$(find gpurun_out/prof_r02p -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python scripts/trace_summary.py $t > gpurun_out/r02p/trace_by_grid.txt 2>&1; [ -n "$t" ] && python scripts/overlap.py $t > gpurun_out/r02p/overlap.txt 2>&1
# 3. PMC
scripts/gpu_pmc.sh r02p --no-cpu-baseline > gpurun_out/r02p/pmc.log 2>&1
cp gpurun_out/pmc_r02p.json gpurun_out/r02p/pmc.json
# 4. SQ for expand
scripts/gpu_sq.sh r02p "expand_rows_kernel|plan_rows_kernel|gather_mean_kernel|linear_split_kernel" --no-cpu-baseline > gpurun_out/r02p/sq.log 2>&1
cp gpurun_out/sq_r02p.json gpurun_out/r02p/sq.json
ls -la gpurun_out/r02p
