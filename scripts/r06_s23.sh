#!/bin/bash
# linear_fused2x: register-staged images (7) against LDS-direct loads (8)
mkdir -p gpurun_out/r06z
for v in 7 8; do
GIGL_F2_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_plan.py -x -q -k fused > gpurun_out/r06z/tests_v$v.log 2>&1
tail -1 gpurun_out/r06z/tests_v$v.log
done
for v in 8 7 8 7; do
  GIGL_F2_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06z/bench_v${v}_$RANDOM.json 2> gpurun_out/r06z/bench_v$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06z/bench_v*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear alone', g['linear']['ms_per_step_alone'], 'ovl', g['linear']['ms_per_step_overlapped'], 'gather alone', g['gather_mean']['ms_per_step_alone'], 'ovl', g['gather_mean']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
