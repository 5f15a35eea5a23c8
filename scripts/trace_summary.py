"""Per-(kernel, grid size) durations from a rocprofv3 --kernel-trace CSV.

bench.py mixes launch shapes in one process (grouped calls in the timed region, single-batch calls in the
untimed edge-counting pass), so rocprofv3's own per-kernel averages blend them; splitting by grid size keeps
the timed-region launches apart.  usage: python scripts/trace_summary.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    acc = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
        a = acc[(name[:44], g)]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    for (name, g), (d, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{name:46s} grid={g:10d} calls={c:6d} avg={d / c / 1e3:9.2f} us total={d / 1e6:9.2f} ms")


if __name__ == "__main__":
    main()
