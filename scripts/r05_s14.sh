#!/bin/bash
# round-5 session 14 (GPU box): record encoder variants (scripts/micro_records.py --device-only verifies every variant's
# bytes against the committed path's before timing it; the ablations skip that)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05l; mkdir -p $o; rm -f $o/encoder_variants.txt
run() {  # run <label> <flags> <env...>
  local label=$1 flags=$2; shift 2
  echo "== $label" >> $o/encoder_variants.txt
  env "$@" timeout 300 python scripts/micro_records.py --device-only $flags 2>&1 | grep "encode (device\|Error\|error" >> $o/encoder_variants.txt
}
run "fused" "" X=1
run "fused + prefetch in the plan pass" "" GIGL_REC_PREFETCH=1
run "plan + scan only, prefetch" --no-verify GIGL_REC_SKIP=both GIGL_REC_PREFETCH=1
run "plan + scan + rows, prefetch" --no-verify GIGL_REC_SKIP=fields GIGL_REC_PREFETCH=1
run "fused + prefetch, 32768 records" "--batch 32768" GIGL_REC_PREFETCH=1
run "fused, 32768 records" "--batch 32768" X=1
cat $o/encoder_variants.txt
