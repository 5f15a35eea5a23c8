#!/bin/bash
# linear_fused2w ablations (timing only): which part of the kernel does the time go to?
mkdir -p gpurun_out/r06x
for ab in 0 1 2 4 8 16 32 3 12 63; do
  GIGL_F2_ABLATE=$ab timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06x/bench_ab${ab}.json 2> gpurun_out/r06x/bench_ab$ab.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06x/bench_ab*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear alone', g['linear']['ms_per_step_alone'], 'ovl', g['linear']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
