"""print per-kernel average durations of a rocprofv3 *_kernel_stats.csv (library kernels only)
usage: python scripts/kstats.py <kernel_stats.csv> [substring ...]"""
import csv
import sys

keys = sys.argv[2:] or ["anonymous namespace", "copyBuffer", "fillBuffer", "rccl"]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in keys):
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{n[:64]:64s} calls={r['Calls']:>6s} avg={float(r['AverageNs']) / 1e3:9.1f} us  {r['Percentage']:>6s}%")
