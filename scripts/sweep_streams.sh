#!/bin/bash
# usage (GPU box): scripts/sweep_streams.sh — bench.py's headline workload at 1..4 streams: value and the dominant kernel's fractions
for s in 1 2 3 4; do
  python bench.py --no-live-pmc --no-emulated-sub --streams $s --no-cpu-baseline 2>/dev/null | python scripts/show_streams.py
done
