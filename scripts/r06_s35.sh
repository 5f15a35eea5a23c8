#!/bin/bash
# expand_rows_kernel: persistent waves (0) against waves that retire after N steps
mkdir -p gpurun_out/r06al
GIGL_EXPAND_ITERS=8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sampler_ties.py tests/test_gpu_plan.py -x -q > gpurun_out/r06al/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r06al/tests.log | tail -1
for it in 0 4 8 16 32 0 8; do
  GIGL_EXPAND_ITERS=$it timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06al/bench_${it}_$RANDOM.json 2> gpurun_out/r06al/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06al/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'expand alone', g['expand']['ms_per_step_alone'], 'ovl', g['expand']['ms_per_step_overlapped'], 'gather ovl', g['gather_mean']['ms_per_step_overlapped'], 'linear ovl', g['linear']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
