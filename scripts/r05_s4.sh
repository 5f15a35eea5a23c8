#!/bin/bash
# round-5 session 4 (GPU box): new tests, then the LG3 dedup workgroup shape under three streams
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05d; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_union_lg3.py tests/test_gpu_train_plan.py "tests/test_gpu_dist_plan.py::test_staged_plan_reports_bucket_overflow" -x -q -m gpu -s > $o/pytest_a.log 2>&1
tail -12 $o/pytest_a.log
GIGL_LG3_NT=512 GIGL_LG3_CAP=8192 timeout 900 python -m pytest tests/test_gpu_union_lg3.py tests/test_gpu_plan.py -x -q -m gpu > $o/pytest_nt512.log 2>&1
tail -3 $o/pytest_nt512.log
run() {  # run <name> <env...>
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 > $o/bench_$name.json 2> $o/bench_$name.err
}
run nt1024_cap16384 GIGL_X=1
run nt1024_cap8192 GIGL_LG3_CAP=8192
run nt512_cap8192 GIGL_LG3_NT=512 GIGL_LG3_CAP=8192
run nt512_cap4096 GIGL_LG3_NT=512 GIGL_LG3_CAP=4096
run nt1024_cap4096 GIGL_LG3_CAP=4096
run nt512_cap16384 GIGL_LG3_NT=512
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05d/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); g=d['roofline']['groups']
        print(f.split('bench_')[1], round(d['value']/1e9,3), round(d['ms_per_step']*1e3,2), 'union_insert alone/ovl', g['union_insert']['ms_per_step_alone']*1e3, g['union_insert']['ms_per_step_overlapped']*1e3)
    except Exception as e: print(f, 'ERR', e)
P
