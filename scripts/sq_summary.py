"""Fold the counter_collection CSVs of scripts/gpu_sq.sh into gpurun_out/sq_<tag>.json: for every kernel and dispatch
shape (grid size), the mean of every SQ counter per dispatch."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def main():
    tag = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"gpurun_out/sq_{tag}/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
                key = f'{name[:60]} grid={row.get("Grid_Size")}'
                a = acc[key][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    out = {k: dict({c: v[0] / v[1] for c, v in d.items()}, dispatches=max(v[1] for v in d.values()))
           for k, d in acc.items()}
    json.dump(out, open(f"gpurun_out/sq_{tag}.json", "w"), indent=1, sort_keys=True)
    for k, d in sorted(out.items()):
        print(k, {c: round(v, 1) for c, v in d.items()})


if __name__ == "__main__":
    main()
