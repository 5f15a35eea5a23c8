#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof_round.sh <tag> <workload> [sq kernel regex]
# one profile round of a bench workload: the driver-flag bench line, rocprofv3 kernel stats of the same command, the
# FETCH_SIZE / WRITE_SIZE passes (scripts/gpu_pmc.sh) and, when a regex is given, the SQ counters of those kernels.
# Summaries land in gpurun_out/<tag>/ with the workload in their names; copy what should be judged into profiles/.
set -u
tag=$1; wl=$2; re=${3:-}
cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; mkdir -p $o
extra=""; [ "$wl" != "products" ] && extra="--workload $wl"
timeout 900 python bench.py --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 $extra > $o/bench_${wl}.json 2> $o/bench_${wl}.err
scripts/gpu_profile.sh ${tag}_${wl} --steps 20 --warmup 5 $extra > $o/profile_${wl}.log 2>&1
f=$(find gpurun_out/prof_${tag}_${wl} -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_${wl}.csv
cp gpurun_out/bench_${tag}_${wl}.json $o/bench_${wl}_under_rocprof.json
t=$(find gpurun_out/prof_${tag}_${wl} -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python scripts/overlap.py $t > $o/overlap_${wl}.txt 2>&1
scripts/gpu_pmc.sh ${tag}_${wl} --no-cpu-baseline $extra > $o/pmc_${wl}.log 2>&1
cp gpurun_out/pmc_${tag}_${wl}.json $o/pmc_${wl}.json
if [ -n "$re" ]; then
  scripts/gpu_sq.sh ${tag}_${wl} "$re" --no-cpu-baseline $extra > $o/sq_${wl}.log 2>&1
  cp gpurun_out/sq_${tag}_${wl}.json $o/sq_${wl}.json
fi
ls -la $o
