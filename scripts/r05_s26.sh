#!/bin/bash
# round-5 session 26 (GPU box): whole -m gpu suite on the tree as committed (no -x: every failure listed), typed-dblp kernel stats
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05ab; mkdir -p $o
timeout 3300 python -m pytest tests -q -m gpu > $o/pytest_gpu.log 2>&1
grep -a "passed\|failed\|^FAILED" $o/pytest_gpu.log | tail -8
rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ty -- python bench.py --workload typed-dblp --steps 20 --warmup 5 > $o/prof.log 2>&1
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_typed-dblp.csv
find $o/prof -type f -size +8M -delete
python scripts/kstats.py $o/kernel_stats_typed-dblp.csv "" | head -24
grep '^{' $o/prof.log | tail -1 > $o/bench_typed-dblp_under_rocprof.json
