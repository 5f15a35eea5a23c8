#!/bin/bash
# usage (GPU box): scripts/pmc_sq.sh <tag> [bench args]   — SQ issue/stall counters per kernel (own pass, --kernel-trace only)
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$tag
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -f csv -d "$out" -o sq -- python bench.py "$@" --no-cpu-baseline > gpurun_out/sq_$tag.log 2>&1
python - "$out" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:36]
        key = (name, int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[key] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])
print(f"{'kernel':38s} {'grid':>9s} {'calls':>5s} {'VALU/call':>10s} {'SALU/call':>10s} {'waveCyc/call':>12s} {'actVALU%':>8s} {'wait%':>6s} {'waitInst%':>9s}")
for (name, g), c in rows[:28]:
    if name.startswith("at::") or "rocprim" in name: continue
    n = max(calls[(name, g)], 1); wc = max(c["SQ_WAVE_CYCLES"], 1)
    print(f"{name:38s} {g:9d} {n:5d} {c['SQ_INSTS_VALU']/n:10.0f} {c['SQ_INSTS_SALU']/n:10.0f} {wc/n:12.0f} {100*c['SQ_ACTIVE_INST_VALU']/wc:8.1f} {100*c['SQ_WAIT_ANY']/wc:6.1f} {100*c['SQ_WAIT_INST_ANY']/wc:9.1f}")
PY
find gpurun_out -name '*counter_collection.csv' -size +8M -delete; find gpurun_out -name '*kernel_trace.csv' -size +8M -delete
