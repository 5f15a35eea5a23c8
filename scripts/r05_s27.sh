#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05ae; mkdir -p $o; rm -f $o/enc.txt
for wg in 0 1; do for b in 1024 4096 8192 32768; do
  echo "== plan workgroup-per-record $wg, B=$b: $(GIGL_REC_PLAN_WG=$wg timeout 300 python scripts/micro_records.py --device-only --batch $b 2>&1 | grep 'encode (device' | cut -c1-140)" >> $o/enc.txt
done; done
cat $o/enc.txt
timeout 900 python -m pytest tests/test_gpu_records.py tests/test_gpu_nablp.py tests/test_gpu_edge_features.py -q -m gpu 2>&1 | grep -a "passed\|failed" | tail -2
