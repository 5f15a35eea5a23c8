#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05x; mkdir -p $o; rm -f $o/enc.txt
for wp in 4 2 1; do
  echo "== waves per plan workgroup $wp" >> $o/enc.txt
  GIGL_REC_WP=$wp timeout 300 python scripts/micro_records.py --device-only 2>&1 | grep "encode (device" >> $o/enc.txt
  GIGL_REC_WP=$wp timeout 300 python scripts/micro_records.py --device-only --batch 32768 2>&1 | grep "encode (device" >> $o/enc.txt
done
cat $o/enc.txt
