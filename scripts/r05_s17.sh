#!/bin/bash
# round-5 session 17 (GPU box): whole -m gpu suite on the tree as it stands
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05o; mkdir -p $o
timeout 3300 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.log 2>&1
tail -6 $o/pytest_gpu.log
