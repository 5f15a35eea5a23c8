#!/bin/bash
# streams x batches-per-call sweep of bench.py (GPU box), round 4 kernels
for sg in "2 64" "3 64" "4 64" "5 64" "3 32" "3 128" "4 128" "4 32" "6 64"; do set -- $sg
python bench.py --no-live-pmc --no-emulated-sub --streams $1 --group $2 --steps 1920 --warmup 128 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$1 G=$2', round(d['value']/1e9,3), 'Gedges/s', round(d['ms_per_step']*1e3,2), 'us/step')"
done
