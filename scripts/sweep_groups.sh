for cfg in "8 1" "8 2" "8 4" "8 8" "4 4" "4 8" "2 8" "2 16" "4 16" "1 8"; do set -- $cfg; python bench.py --no-live-pmc --no-emulated-sub --streams $1 --group $2 --steps 480 --warmup 32 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('S=$1 G=$2', 'ms/step', round(d['ms_per_step'],4), 'Gedges/s', round(d['value']/1e9,3), d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
