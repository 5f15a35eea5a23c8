#!/bin/bash
mkdir -p gpurun_out/r06am
for it in 16 0 16 0 16 0 12 24; do
  GIGL_EXPAND_ITERS=$it timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06am/bench_${it}_$RANDOM.json 2> gpurun_out/r06am/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06am/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'expand alone', g['expand']['ms_per_step_alone'], 'ovl', g['expand']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
