#!/bin/bash
# linear_fused2: chunks of operands in flight (GIGL_F2_VARIANT 0: 1 chunk / 3 waves, 1: 2/2, 2: 3/2, 3: 2/3)
mkdir -p gpurun_out/r06w
for v in 0 4 5 6; do
  GIGL_F2_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_plan.py -x -q -k fused > gpurun_out/r06w/tests_v$v.log 2>&1
  tail -1 gpurun_out/r06w/tests_v$v.log
done
for v in 0 4 5 6 0 4; do
  GIGL_F2_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06w/bench_v${v}_$RANDOM.json 2> gpurun_out/r06w/bench_v$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06w/bench_v*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear alone', g['linear']['ms_per_step_alone'], 'ovl', g['linear']['ms_per_step_overlapped'], 'gather alone', g['gather_mean']['ms_per_step_alone'])
    except Exception as e: print(f, 'ERR', e)
PY
