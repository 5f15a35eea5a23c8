#!/bin/bash
# linear_fused2x: bias staged in LDS; 7 = W2 rounds LDS-direct, 9 = register-staged rounds, 8 = everything LDS-direct
mkdir -p gpurun_out/r06af
for v in 7 8 9; do
GIGL_F2_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_plan.py -x -q -k fused > gpurun_out/r06af/tests_v$v.log 2>&1
tail -1 gpurun_out/r06af/tests_v$v.log
done
for v in 7 9 8 7 9; do
  GIGL_F2_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06af/bench_v${v}_$RANDOM.json 2> gpurun_out/r06af/bench_v$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06af/bench_v*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear alone', g['linear']['ms_per_step_alone'], 'ovl', g['linear']['ms_per_step_overlapped'], 'gather alone', g['gather_mean']['ms_per_step_alone'])
    except Exception as e: print(f, 'ERR', e)
PY
