#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/gpu_profile.sh <tag> [bench args...]
# runs bench.py under rocprofv3 --kernel-trace --stats and leaves the per-kernel summary in
# gpurun_out/prof_<tag>/ (copy the *_kernel_stats.csv you want judged into profiles/).
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -f csv -d "$out" -o "$tag" -- python bench.py --no-live-pmc --no-emulated-sub "$@" --no-cpu-baseline > gpurun_out/bench_$tag.log 2>&1
grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json
# keep only the small summaries (the raw trace can be 100s of MB)
find "$out" -name '*kernel_trace.csv' -size +20M -delete
find "$out" -type f | head -20
