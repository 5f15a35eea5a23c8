#!/bin/bash
# fused2x default: streams x batches per call
mkdir -p gpurun_out/r06ac
for cfg in "2 64" "3 64" "2 128" "2 32" "2 64" "3 64" "2 96" "3 128"; do
  set -- $cfg
  timeout 600 python bench.py --steps 20 --warmup 5 --streams $1 --group $2 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06ac/bench_s$1_g$2_$RANDOM.json 2> gpurun_out/r06ac/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ac/bench_s*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2))
    except Exception as e: print(f, 'ERR', e)
PY
