#!/bin/bash
# round-6 session 43: FINAL tree — whole -m gpu suite, smoke, the products profile set, the secondary lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06at
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06at/pytest_gpu.log 2>&1
echo "pytest rc=$? in $(( $(date +%s) - t0 )) s" | tee gpurun_out/r06at/pytest_gpu_tail.txt
grep -E "passed|failed" gpurun_out/r06at/pytest_gpu.log | tail -2 | tee -a gpurun_out/r06at/pytest_gpu_tail.txt
grep -E "^FAILED" gpurun_out/r06at/pytest_gpu.log | head
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06at/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06at/smoke.log
bash scripts/prof_round.sh r06at products "linear_fused2x_kernel|gather_mean_kernel|lg3_dedup_kernel" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06at/bench_products_driver_flags.json 2> gpurun_out/r06at/bench_df.err
tail -1 gpurun_out/r06at/bench_df.err; head -c 300 gpurun_out/r06at/bench_products_driver_flags.json; echo
timeout 600 python bench.py --train > gpurun_out/r06at/bench_train.json 2> gpurun_out/r06at/bench_train.err
timeout 600 python bench.py --train --train-task lp > gpurun_out/r06at/bench_train_lp.json 2> gpurun_out/r06at/bench_train_lp.err
timeout 600 python bench.py --entry inferencer --no-live-pmc --no-cpu-baseline > gpurun_out/r06at/bench_entry_inferencer.json 2> gpurun_out/r06at/bench_entry_inferencer.err
python - <<'PY'
import json
for n in ('train','train_lp','entry_inferencer'):
    try:
        j=json.loads(open(f'gpurun_out/r06at/bench_{n}.json').read().strip().splitlines()[-1]); print(n, j['value']/1e9, j['ms_per_step'])
    except Exception as e: print(n,'ERR',e)
PY
