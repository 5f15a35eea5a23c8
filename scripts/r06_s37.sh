#!/bin/bash
# LG3 table size beside the 64-KB projection workgroups
mkdir -p gpurun_out/r06an
for cap in 16384 8192 16384 8192 4096; do
  GIGL_LG3_CAP=$cap timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06an/bench_${cap}_$RANDOM.json 2> gpurun_out/r06an/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06an/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'insert alone', g['union_insert']['ms_per_step_alone'], 'ovl', g['union_insert']['ms_per_step_overlapped'], 'linear ovl', g['linear']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
