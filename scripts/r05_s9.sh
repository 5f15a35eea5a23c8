#!/bin/bash
# round-5 session 9 (GPU box): the link-prediction training plan — parity, the trainer through it, its bench line + trace
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05i; mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_train_plan.py -x -q -m gpu > $o/pytest_plan.log 2>&1
tail -30 $o/pytest_plan.log
timeout 1800 python -m pytest tests/test_gpu_nablp.py tests/test_gpu_entry_points.py -x -q -m gpu > $o/pytest_nablp.log 2>&1
tail -12 $o/pytest_nablp.log
timeout 600 python bench.py --train --train-task lp --steps 64 --warmup 8 > $o/bench_lp_train.json 2> $o/bench_lp_train.err
tail -3 $o/bench_lp_train.err; head -c 600 $o/bench_lp_train.json
rocprofv3 --kernel-trace --stats -f csv -d $o/prof_lp -o lp -- python bench.py --train --train-task lp --steps 64 --warmup 8 > $o/prof_lp.log 2>&1
f=$(find $o/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_lp_train.csv
find $o/prof_lp -type f -size +8M -delete
python scripts/kstats.py $o/kernel_stats_lp_train.csv "" | head -40
# A/B on one box: the long-row pass's LDS shape
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 > $o/bench_products_cap4096_$i.json 2>/dev/null
  GIGL_LG_BIG_CAP=16384 timeout 600 python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub --steps 20 --warmup 5 > $o/bench_products_cap16384_$i.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05i/bench_products_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); g=d['roofline']['groups']
        print(f.split('bench_')[1], round(d['value']/1e9,3), round(d['ms_per_step']*1e3,2), {k:(round(v['ms_per_step_alone']*1e3,2),round(v['ms_per_step_overlapped']*1e3,2)) for k,v in g.items()})
    except Exception as e: print(f,'ERR',e)
P
