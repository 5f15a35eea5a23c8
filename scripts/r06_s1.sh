#!/bin/bash
# round-6 session 1: encoder tests after the wave fence fix + the round's starting lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06a; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_records.py -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 > $o/bench_products_driver_flags.json 2> $o/bench.err
tail -2 $o/bench.err; head -c 1500 $o/bench_products_driver_flags.json; echo
timeout 600 python bench.py --workload mag240m-sharded > $o/bench_mag240m-sharded.json 2> $o/bench_sh.err
tail -2 $o/bench_sh.err; head -c 1500 $o/bench_mag240m-sharded.json; echo
