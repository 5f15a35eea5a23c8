#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06bl
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06bl/pytest_gpu.log 2>&1
echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r06bl/pytest_gpu_tail.txt
grep -E "passed|failed" gpurun_out/r06bl/pytest_gpu.log | tail -2 | tee -a gpurun_out/r06bl/pytest_gpu_tail.txt
grep -E "^FAILED" gpurun_out/r06bl/pytest_gpu.log | head
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06bl/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --workload gat-lp --train > gpurun_out/r06bl/bench_gat-lp_train.json 2> gpurun_out/r06bl/err
timeout 600 python bench.py --train --train-task lp > gpurun_out/r06bl/bench_train_lp.json 2> gpurun_out/r06bl/err
timeout 600 python bench.py --train > gpurun_out/r06bl/bench_train.json 2> gpurun_out/r06bl/err
python - <<'PY'
import json
for n in ('gat-lp_train','train_lp','train'):
    j=json.loads(open(f'gpurun_out/r06bl/bench_{n}.json').read().strip().splitlines()[-1]); print(n, j['value']/1e9, j['ms_per_step'])
PY
