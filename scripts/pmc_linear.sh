#!/bin/bash
# SQ counters of the projection kernel on the micro-benchmark (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
rm -rf gpurun_out/sq_lin
QUICK=1 timeout 600 rocprofv3 --kernel-trace --pmc $set -f csv -d gpurun_out/sq_lin -o lin -- python scripts/micro_linear.py > gpurun_out/sq_lin.log 2>&1
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("gpurun_out/sq_lin/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear" not in r["Kernel_Name"]: continue
        k=int(r["Grid_Size"]); acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
for g,c in acc.items():
    print("grid",g,"calls",n[g],{k:round(v/max(n[g],1)/1e6,2) for k,v in c.items()})
PY
done
tail -3 gpurun_out/sq_lin.log
