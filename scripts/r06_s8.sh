#!/bin/bash
# round-6 session 8: the round's products profile set (driver flags, rocprof kernel stats, overlap, PMC, SQ of the projection)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scripts/prof_round.sh r06h products "linear_fused2_kernel|gather_mean_kernel|lg3_dedup_kernel" 2>&1 | tail -30
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06h/bench_products_driver_flags.json 2> gpurun_out/r06h/bench_df.err
tail -2 gpurun_out/r06h/bench_df.err; head -c 600 gpurun_out/r06h/bench_products_driver_flags.json; echo
python -c "
import json
j=json.load(open('gpurun_out/r06h/sq_products.json'))
print(json.dumps(j, indent=0)[:6000])
"
