#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06e; mkdir -p $o
timeout 5000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $o/pytest_gpu_tail.txt
cat $o/pytest_gpu_tail.txt
for a in "" "--train" "--train --train-task lp" "--entry inferencer" "--workload gat-lp --train"; do
  n=$(echo "bench$a" | tr ' ' '_' | tr -d '-')
  timeout 900 python bench.py $a --no-cpu-baseline > $o/$n.json 2> $o/$n.err
  python - $o/$n.json <<PY
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print(sys.argv[1], j["value"], j["ms_per_step"], j.get("steps"), j.get("steps_honoured"))
else: print(sys.argv[1], "NO LINE")
PY
done
