"""Fold the rocprofv3 counter_collection CSVs written by scripts/gpu_pmc.sh into one JSON:

  calibration: what FETCH_SIZE / WRITE_SIZE report for launches with KNOWN byte counts (scripts/pmc_calib.py)
               -> bytes-per-counter-unit factors for a streaming access and for the row-gather access
  kernels:     per kernel of the bench run, mean FETCH_SIZE / WRITE_SIZE per dispatch (raw counter units, KB as
               rocprofv3 defines them) and the HBM bytes after the streaming-copy correction

usage: python scripts/pmc_summary.py <tag>   (reads gpurun_out/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE}/..., writes
gpurun_out/pmc_<tag>.json)"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def per_kernel(dirpath, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = acc[row["Kernel_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:96]


def main():
    tag = sys.argv[1]
    base = f"gpurun_out/pmc_{tag}"
    known = None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            lines = [l for l in open(f"{base}_{ctr}_calib.log") if l.startswith("{")]
            known = json.loads(lines[-1])
        except (OSError, IndexError):
            pass
    shape = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):  # the launch shape of the profiled bench run (its --timed-only line)
        try:
            lines = [l for l in open(f"{base}_{ctr}_bench.log") if l.startswith("{")]
            shape = json.loads(lines[-1])
        except (OSError, IndexError):
            pass
    out = {"tag": tag, "workload": shape.get("workload", "products"), "projected_input": shape.get("projected_input"),
           "batches_per_call": shape.get("batches_per_call"), "streams": shape.get("streams"),
           "steps_executed": shape.get("steps_executed"), "calls_executed": shape.get("calls_executed"),
           "counter_unit": "KB (rocprofv3 FETCH_SIZE / WRITE_SIZE)", "known_bytes": known,
           "calibration": {}, "kernels": {}}
    corr = {}
    for ctr, key in (("FETCH_SIZE", "read"), ("WRITE_SIZE", "write")):
        cal = per_kernel(f"{base}_{ctr}/calib", ctr)
        copy = [(k, v) for k, v in cal.items() if "elementwise" in k and v[1] >= 4 and v[0] > 1e5]
        gath = [(k, v) for k, v in cal.items() if "gather_rows" in k]
        entry = {}
        if copy and known:
            k, (mean, calls) = max(copy, key=lambda kv: kv[1][0])
            entry["streaming_copy"] = {"kernel": short(k), "calls": calls, "counter_mean": mean,
                                       "known_bytes": known[f"copy_{key}_bytes"],
                                       "bytes_per_unit": known[f"copy_{key}_bytes"] / mean,
                                       "factor_vs_1024": known[f"copy_{key}_bytes"] / (mean * 1024.0)}
            corr[ctr] = entry["streaming_copy"]["bytes_per_unit"]
        if gath and known:
            k, (mean, calls) = gath[0]
            entry["row_gather"] = {"kernel": short(k), "calls": calls, "counter_mean": mean,
                                   "algorithmic_bytes": known[f"gather_rows_{key}_bytes"],
                                   "counter_bytes_with_streaming_factor": mean * corr.get(ctr, 1024.0)}
        out["calibration"][ctr] = entry
    out["bytes_per_unit_used"] = {c: corr.get(c, 1024.0) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    kern = defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, (mean, calls) in per_kernel(f"{base}_{ctr}/bench", ctr).items():
            e = kern[short(k)]
            e[ctr + "_mean"] = mean
            e[ctr + "_calls"] = calls
            e[ctr + "_bytes"] = mean * out["bytes_per_unit_used"][ctr]
    for k, e in kern.items():
        e["hbm_bytes_per_launch"] = e.get("FETCH_SIZE_bytes", 0.0) + e.get("WRITE_SIZE_bytes", 0.0)
    out["kernels"] = dict(sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]))
    json.dump(out, open(f"{base}.json", "w"), indent=1)
    print(json.dumps(out["calibration"], indent=1))
    for k, e in list(out["kernels"].items())[:24]:
        print(f"{e['hbm_bytes_per_launch'] / 1e6:10.3f} MB/launch  {k}")


if __name__ == "__main__":
    main()
