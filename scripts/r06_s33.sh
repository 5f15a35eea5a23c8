#!/bin/bash
# 256-wide layers on linear_fused2x_kernel's first product (GIGL_LINEAR_ROWS256=0: linear_split_kernel)
mkdir -p gpurun_out/r06aj
timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_dist.py tests/test_gpu_dist_peer.py tests/test_gpu_hbm_route.py -x -q > gpurun_out/r06aj/tests.log 2>&1
tail -3 gpurun_out/r06aj/tests.log
for wl in rmat-shard mag-shard mag240m-sharded; do
for on in 1 0; do
  GIGL_LINEAR_ROWS256=$on timeout 900 python bench.py --workload $wl --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-emulated-sub > gpurun_out/r06aj/bench_${wl}_$on.json 2> gpurun_out/r06aj/bench_${wl}_$on.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06aj/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); g=(j.get('roofline') or {}).get('groups') or {}
        lin=g.get('linear',{})
        print(f, round(j['value']/1e9,3), round(j['ms_per_step']*1e3,2), 'linear alone', lin.get('ms_per_step_alone'), 'ovl', lin.get('ms_per_step_overlapped'), 'frac', lin.get('frac_alone'))
    except Exception as e: print(f, 'ERR', e)
PY
