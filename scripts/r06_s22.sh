#!/bin/bash
# linear_fused2x (variant 7: whole hidden width per workgroup) against 4 (fused2w) and 0 (round 5)
mkdir -p gpurun_out/r06y
GIGL_F2_VARIANT=7 timeout 900 python -m pytest tests/test_gpu_plan.py -x -q > gpurun_out/r06y/tests_v7.log 2>&1
tail -3 gpurun_out/r06y/tests_v7.log
for v in 7 4 0 7 4; do
  GIGL_F2_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06y/bench_v${v}_$RANDOM.json 2> gpurun_out/r06y/bench_v$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06y/bench_v*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear alone', g['linear']['ms_per_step_alone'], 'ovl', g['linear']['ms_per_step_overlapped'], 'gather alone', g['gather_mean']['ms_per_step_alone'], 'ovl', g['gather_mean']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
