#!/bin/bash
# fused2x default: streams in flight
mkdir -p gpurun_out/r06ab
for s in 2 3 4 3 4; do
  timeout 600 python bench.py --steps 20 --warmup 5 --streams $s --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06ab/bench_s${s}_$RANDOM.json 2> gpurun_out/r06ab/bench_s$s.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ab/bench_s*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear ovl', g['linear']['ms_per_step_overlapped'], 'gather ovl', g['gather_mean']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
