#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/pmc_records.sh <tag>
# counters of the record encoder (scripts/micro_records.py --device-only), one rocprofv3 pass per counter group
# (--kernel-trace + --pmc only); leaves gpurun_out/pmc_records_<tag>.txt
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_records_$tag
mkdir -p $out
i=0
for grp in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_FLAT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -f csv -d $out/p$i -o p -- python scripts/micro_records.py --device-only \
      > $out/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $out/p$i.log)"
done
python - <<PY > gpurun_out/pmc_records_$tag.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$out/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        short = next((n for n in ("record_plan", "record_write", "record_scan", "row_crc") if n in k), None)
        if short is None:
            continue
        name = short + "." + row["Counter_Name"]
        acc[name][0] += float(row["Counter_Value"]); acc[name][1] += 1
# a dispatch reports one row per counter (summed over XCDs / instances by the tool's _sum names)
disp = collections.Counter()
import json
out = {}
for name, (v, n) in sorted(acc.items()):
    print(f"{name:60s} total {v:16.0f}  dispatches {n}  mean {v / max(n, 1):14.1f}")
    out[name] = {"total": v, "dispatches": n, "mean": v / max(n, 1)}
json.dump(out, open("gpurun_out/pmc_records_$tag.json", "w"), indent=1)
PY
cat gpurun_out/pmc_records_$tag.txt
find $out -name '*.csv' -size +4M -delete
