#!/bin/bash
# round-5 session 25 (GPU box): streams x batches-per-call sweep of the headline on the final build
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05aa; mkdir -p $o; rm -f $o/sweep.txt
for s in 2 3 4 5 6; do for g in 32 64 128; do
  v=$(timeout 300 python bench.py --no-live-pmc --no-emulated-sub --no-cpu-baseline --streams $s --group $g --min-seconds 1.5 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f G  %.2f us' % (d['value']/1e9, d['ms_per_step']*1e3))")
  echo "streams $s group $g: $v" >> $o/sweep.txt
done; done
cat $o/sweep.txt
