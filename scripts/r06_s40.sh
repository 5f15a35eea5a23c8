#!/bin/bash
# flakiness check: the whole -m gpu suite twice more, and a longer driver-flag run
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06aq
for i in 1 2; do
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06aq/pytest_gpu_$i.log 2>&1
grep -E "passed|failed" gpurun_out/r06aq/pytest_gpu_$i.log | tail -1
grep -E "^FAILED" gpurun_out/r06aq/pytest_gpu_$i.log | head
done
timeout 900 python bench.py --steps 20 --warmup 5 --min-seconds 30 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06aq/bench_long.json 2> gpurun_out/r06aq/bench_long.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06aq/bench_long.json').read().strip().splitlines()[-1])
print(j['value']/1e9, j['ms_per_step']*1e3, j['timing'].get('repetitions'), j['timing'].get('ms_per_step_p10'), j['timing'].get('ms_per_step_p90'), j.get('overflowed_calls'), j['config'].get('overflowed_calls'))
PY
