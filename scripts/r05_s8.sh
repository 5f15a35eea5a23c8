#!/bin/bash
# round-5 session 8 (GPU box): half split for layers >= 1, 32-KB long-row pass, parallel fused2 prepare; record encoder trace
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05h; mkdir -p $o
timeout 2400 python -m pytest tests/test_gpu_plan.py tests/test_gpu_union_lg3.py tests/test_gpu_parity.py tests/test_gpu_fullsize_shards.py tests/test_gpu_fullsize.py tests/test_gpu_groups.py tests/test_gpu_sage_options.py -x -q -m gpu -s > $o/pytest_a.log 2>&1
tail -6 $o/pytest_a.log; grep -E "max \|err\|" $o/pytest_a.log | tail -8
run() { local name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 $EXTRA > $o/bench_$name.json 2> $o/bench_$name.err; }
EXTRA=""; run products GIGL_X=1
EXTRA="--streams 4"; run products_s4 GIGL_X=1
EXTRA="--streams 2"; run products_s2 GIGL_X=1
EXTRA="--group 128"; run products_g128 GIGL_X=1
EXTRA="--workload rmat-shard"; run rmat GIGL_X=1
EXTRA="--workload rmat-shard"; run rmat_bf16_layers GIGL_PLAN_HS_LAYERS=0
EXTRA="--workload cora"; run cora GIGL_X=1
rocprofv3 --kernel-trace --stats -f csv -d $o/prof_sampler -o sampler -- python bench.py --entry sampler --no-cpu-baseline --steps 20 --warmup 5 > $o/prof_sampler.log 2>&1
f=$(find $o/prof_sampler -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_entry_sampler.csv
grep '^{' $o/prof_sampler.log | tail -1 > $o/bench_entry_sampler_under_rocprof.json
find $o/prof_sampler -type f -size +8M -delete
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05h/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        g=r.get('groups') or {}
        print(f.split('bench_')[1], round(d['value']/1e9,3), round(d['ms_per_step']*1e3,2), r.get('kernel'), r.get('frac'), {k:(round(v['ms_per_step_alone']*1e3,2),round(v['ms_per_step_overlapped']*1e3,2)) for k,v in g.items()})
    except Exception as e: print(f,'ERR',e)
P
python scripts/kstats.py $o/kernel_stats_entry_sampler.csv | head -20
