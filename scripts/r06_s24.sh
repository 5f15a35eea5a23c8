#!/bin/bash
# round-6 session 24: linear_fused2x_kernel as the default — tests, the products profile set, the kernel's ablation table
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06aa
timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_entry_points.py tests/test_gpu_hbm_route.py tests/test_gpu_overflow.py -x -q > gpurun_out/r06aa/tests.log 2>&1
tail -2 gpurun_out/r06aa/tests.log
bash scripts/prof_round.sh r06aa products "linear_fused2x_kernel|gather_mean_kernel|lg3_dedup_kernel" 2>&1 | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06aa/bench_products_driver_flags.json 2> gpurun_out/r06aa/bench_df.err
tail -2 gpurun_out/r06aa/bench_df.err; head -c 400 gpurun_out/r06aa/bench_products_driver_flags.json; echo
for ab in 0 1 2 4 8 16 32 63; do
  GIGL_F2_ABLATE=$ab timeout 600 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-sharded-sub > gpurun_out/r06aa/bench_ab${ab}.json 2> gpurun_out/r06aa/bench_ab$ab.err
done
python - <<'PY' | tee gpurun_out/r06aa/fused2x_ablations.txt
import json,glob
print("linear_fused2x_kernel with parts switched off (GIGL_F2_ABLATE; rows wrong, timing only): bench.py --steps 20 --warmup 5")
names={0:'whole kernel',1:'no A loads',2:'no W image copies',4:'one MFMA of three in the first product',8:'no second product',16:'no output stores',32:'no barriers',63:'all of the above'}
for ab in (0,1,2,4,8,16,32,63):
    try:
        j=json.loads(open(f'gpurun_out/r06aa/bench_ab{ab}.json').read().strip().splitlines()[-1]); g=j['roofline']['groups']
        print(f"{ab:3d} {names[ab]:42s} linear alone {g['linear']['ms_per_step_alone']*1e3:6.2f} us/step  overlapped {g['linear']['ms_per_step_overlapped']*1e3:6.2f}  step {j['ms_per_step']*1e3:6.2f} us")
    except Exception as e: print(ab, 'ERR', e)
PY
