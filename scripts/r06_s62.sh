#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06bq
for i in 1 2; do
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06bq/pytest_gpu_$i.log 2>&1
echo "default (fork on, eager layers) $i: rc=$? $(grep -E 'passed|failed|Segmentation' gpurun_out/r06bq/pytest_gpu_$i.log | tail -1 | cut -c1-90)" | tee -a gpurun_out/r06bq/pytest_gpu_tail.txt
grep -E "^FAILED" gpurun_out/r06bq/pytest_gpu_$i.log | head -3
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06bq/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --workload gat-lp --train > gpurun_out/r06bq/bench_gat-lp_train.json 2> gpurun_out/r06bq/err
timeout 600 python bench.py --train --train-task lp > gpurun_out/r06bq/bench_train_lp.json 2> gpurun_out/r06bq/err
GIGL_LP_FORK=0 timeout 600 python bench.py --workload gat-lp --train > gpurun_out/r06bq/bench_gat-lp_train_nofork.json 2> gpurun_out/r06bq/err
GIGL_LP_FORK=0 timeout 600 python bench.py --train --train-task lp > gpurun_out/r06bq/bench_train_lp_nofork.json 2> gpurun_out/r06bq/err
python - <<'PY'
import json
for n in ('gat-lp_train','gat-lp_train_nofork','train_lp','train_lp_nofork'):
    j=json.loads(open(f'gpurun_out/r06bq/bench_{n}.json').read().strip().splitlines()[-1]); print(n, j['value']/1e9, j['ms_per_step'])
PY
