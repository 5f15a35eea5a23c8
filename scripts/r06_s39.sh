#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06ap
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06ap/pytest_gpu.log 2>&1
echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r06ap/pytest_gpu_tail.txt
grep -E "passed|failed" gpurun_out/r06ap/pytest_gpu.log | tail -2 | tee -a gpurun_out/r06ap/pytest_gpu_tail.txt
grep -E "^FAILED" gpurun_out/r06ap/pytest_gpu.log | head
