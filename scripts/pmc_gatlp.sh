#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_gatlp.sh <tag>
# HBM-side traffic of the gat-lp step's kernels (FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only, as
# scripts/gpu_pmc.sh): per-kernel mean counter per dispatch -> gpurun_out/pmc_gatlp_<tag>.txt
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_gatlp_${tag}_${ctr}
  GIGL_BENCH_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc $ctr -f csv -d "$out" -o b -- python bench.py \
      --workload gat-lp --steps 20 --warmup 5 --min-seconds 0.3 > gpurun_out/pmc_gatlp_${tag}_${ctr}.log 2>&1
done
python - "$tag" <<'PY' | tee gpurun_out/pmc_gatlp_$1.txt
import csv, glob, sys, collections
tag = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_gatlp_{tag}_{ctr}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != ctr:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        res[k][ctr] = v / n
        res[k]["n"] = n
print("kernel, dispatches, mean FETCH_SIZE, mean WRITE_SIZE per dispatch (raw counter units: see MI355X_MICROARCH.md)")
for k, d in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:14]:
    print(f"{k[:70]:70s} n={d['n']:5d} fetch={d.get('FETCH_SIZE', 0):14.0f} write={d.get('WRITE_SIZE', 0):14.0f}")
PY
find gpurun_out -name '*counter_collection.csv' -size +4M -delete
find gpurun_out -name '*kernel_trace.csv' -size +4M -delete
