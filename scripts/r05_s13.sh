#!/bin/bash
# round-5 session 13 (GPU box): fan-outs past 64 in the record encoder, the sharded plan and the typed plan
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05k; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_records.py tests/test_gpu_dist_plan.py tests/test_gpu_graphdb_sampler.py -q -m gpu \
  -k "long_streams or beyond or fanouts" > $o/pytest_fanout.log 2>&1
tail -40 $o/pytest_fanout.log
