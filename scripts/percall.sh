#!/bin/bash
# per-kernel time of one 16-batch call, single stream, eager launches (GPU box): scripts/percall.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_pc -o pc -- python bench.py --no-live-pmc --no-emulated-sub --streams 1 --group 16 --steps 320 --warmup 32 --no-graph --no-cpu-baseline > gpurun_out/pc.log 2>&1
grep "^{" gpurun_out/pc.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=1 G=16 us/step', round(d['ms_per_step']*1e3,2))"
python scripts/percall.py gpurun_out/prof_pc/pc_kernel_trace.csv 20 2>/dev/null | head -24
