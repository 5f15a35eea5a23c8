#!/bin/bash
# float halves on ONE stream, graph halves on the plans' own streams
mkdir -p gpurun_out/r06ad
for cfg in "float-shared 3" "float-shared 4" "float-shared 2" "off 3" "float-shared 6"; do
  set -- $cfg
  timeout 600 python bench.py --steps 20 --warmup 5 --streams $2 --graph-priority $1 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06ad/bench_$1_s$2.json 2> gpurun_out/r06ad/bench_$1_s$2.err
  tail -1 gpurun_out/r06ad/bench_$1_s$2.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ad/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear ovl', g['linear']['ms_per_step_overlapped'], 'gather ovl', g['gather_mean']['ms_per_step_overlapped'], 'expand ovl', g['expand']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
