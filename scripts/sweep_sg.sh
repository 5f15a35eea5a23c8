#!/bin/bash
# streams x batches-per-call sweep of bench.py (GPU box)
for sg in "2 16" "3 16" "4 16" "2 32" "3 8" "4 8" "3 32"; do set -- $sg
python bench.py --no-live-pmc --no-emulated-sub --streams $1 --group $2 --steps 1920 --warmup 128 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$1 G=$2', round(d['value']/1e9,3), 'Gedges/s', round(d['ms_per_step']*1e3,2), 'us/step')"
done
