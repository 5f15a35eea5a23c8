#!/bin/bash
# usage (GPU box): scripts/prof_train.sh <tag> — rocprofv3 kernel stats of `bench.py --train` (the library training plan)
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -f csv -d "$out" -o "$tag" -- python bench.py --no-live-pmc --train --no-cpu-baseline --steps 128 "$@" > gpurun_out/bench_$tag.log 2>&1
grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json
find "$out" -name '*kernel_trace.csv' -size +20M -delete
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
python scripts/kstats.py $f "" | sort -t= -k3 -n -r | head -40
