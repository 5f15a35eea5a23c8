#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06bd
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06bd/pytest_gpu.log 2>&1
echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r06bd/pytest_gpu_tail.txt
grep -E "passed|failed" gpurun_out/r06bd/pytest_gpu.log | tail -2 | tee -a gpurun_out/r06bd/pytest_gpu_tail.txt
grep -E "^FAILED" gpurun_out/r06bd/pytest_gpu.log | head
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06bd/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06bd/smoke.log
