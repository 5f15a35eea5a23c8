#!/bin/bash
# linear_fused2x_kernel: the W_r-only output block skipped for tiles past the roots (GIGL_F2_ALL_WR=1: every tile computes it)
mkdir -p gpurun_out/r06ar
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_groups.py tests/test_gpu_fullsize.py tests/test_gpu_plan_graph.py -x -q > gpurun_out/r06ar/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r06ar/tests.log | tail -1
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export GIGL_F2_ALL_WR=1; else unset GIGL_F2_ALL_WR; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06ar/bench_allwr${v}_$RANDOM.json 2> gpurun_out/r06ar/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ar/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear alone', g['linear']['ms_per_step_alone'], 'ovl', g['linear']['ms_per_step_overlapped'])
    except Exception as e: print(f, 'ERR', e)
PY
