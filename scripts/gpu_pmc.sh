#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/gpu_pmc.sh <tag>
# HBM-traffic counters for every kernel of one bench step, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one TCC pass), each with --kernel-trace only
# (no sys/hip/hsa trace domains).  Each pass profiles (a) scripts/pmc_calib.py — known byte counts — and
# (b) a short single-stream, eager-launch bench.py run of grouped calls only (--timed-only: no probe / counting pass,
# so a kernel's mean over its dispatches is the mean over the launches roofline.achieved is quoted on).  scripts/pmc_summary.py folds the four CSVs into
# gpurun_out/pmc_<tag>.json (per-kernel mean counter per dispatch + calibration factors).
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${tag}_${ctr}
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -f csv -d "$out/calib" -o calib -- python scripts/pmc_calib.py \
      > gpurun_out/pmc_${tag}_${ctr}_calib.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -f csv -d "$out/bench" -o bench -- python bench.py --no-live-pmc --no-emulated-sub --streams 1 \
      --no-graph --steps 64 --min-rounds 2 --warmup 32 --timed-only "$@" > gpurun_out/pmc_${tag}_${ctr}_bench.log 2>&1
done
python scripts/pmc_summary.py "$tag"
# the raw per-dispatch CSVs are large; keep the summary only
find gpurun_out -name '*counter_collection.csv' -size +8M -delete
find gpurun_out -name '*kernel_trace.csv' -size +8M -delete
