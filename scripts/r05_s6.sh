#!/bin/bash
# round-5 session 6 (GPU box): the whole GPU suite, then one bench line per workload
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05f; mkdir -p $o
timeout 2400 python -m pytest tests -x -q -m gpu -s > $o/pytest_gpu.log 2>&1
tail -15 $o/pytest_gpu.log
grep -E "max \|err\||full size" $o/pytest_gpu.log > $o/measured_errors.txt
b() { local name=$1; shift; timeout 900 python bench.py "$@" > $o/bench_$name.json 2> $o/bench_$name.err; }
b products_driver_flags --steps 20 --warmup 5
b mag-shard --workload mag-shard --no-live-pmc --steps 20 --warmup 5
b rmat-shard --workload rmat-shard --no-live-pmc --steps 20 --warmup 5
b mag240m-sharded --workload mag240m-sharded --steps 20 --warmup 5
b cora --workload cora --no-live-pmc --steps 20 --warmup 5
b gat-lp --workload gat-lp --steps 20 --warmup 5
b train --train --steps 20 --warmup 5
ls -la $o
