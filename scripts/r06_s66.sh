#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06bu
for i in 1 2; do
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06bu/pytest_gpu_$i.log 2>&1
echo "whole suite $i: rc=$? $(grep -E 'passed|failed|Segmentation' gpurun_out/r06bu/pytest_gpu_$i.log | tail -1 | cut -c1-90)" | tee -a gpurun_out/r06bu/pytest_gpu_tail.txt
grep -E "^FAILED" gpurun_out/r06bu/pytest_gpu_$i.log | head -3
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06bu/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --workload gat-lp --train > gpurun_out/r06bu/bench_gat-lp_train.json 2> gpurun_out/r06bu/err
timeout 600 python bench.py --train --train-task lp > gpurun_out/r06bu/bench_train_lp.json 2> gpurun_out/r06bu/err
timeout 600 python bench.py --train > gpurun_out/r06bu/bench_train.json 2> gpurun_out/r06bu/err
python - <<'PY'
import json
for n in ('gat-lp_train','train_lp','train'):
    j=json.loads(open(f'gpurun_out/r06bu/bench_{n}.json').read().strip().splitlines()[-1]); print(n, j['value']/1e9, j['ms_per_step'])
PY
