#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
show() { python - "$1" <<PY
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); g=j["roofline"]["groups"]
print(sys.argv[1], round(j["value"]/1e9,3), "G", round(j["ms_per_step"]*1e3,2), "us | linear alone/ovl", g["linear"]["ms_per_step_alone"], g["linear"]["ms_per_step_overlapped"], "| gather", g["gather_mean"]["ms_per_step_alone"], g["gather_mean"]["ms_per_step_overlapped"])
PY
}
o=gpurun_out/r06l; mkdir -p $o
cp gigl_amd/libgigl_hip.so /tmp/keep.so
for rep in 1 2; do
  cp /tmp/keep.so gigl_amd/libgigl_hip.so
  python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub > $o/real_$rep.json 2>/dev/null; show $o/real_$rep.json
  cp gigl_amd/libgigl_hip_nosplit.so gigl_amd/libgigl_hip.so
  python bench.py --no-cpu-baseline --no-live-pmc --no-emulated-sub > $o/nosplit_$rep.json 2>/dev/null; show $o/nosplit_$rep.json
done
cp /tmp/keep.so gigl_amd/libgigl_hip.so
