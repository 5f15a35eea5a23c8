"""Concurrency summary of a rocprofv3 kernel trace (*_kernel_trace.csv): how much of the traced span has 0, 1, 2, ...
kernels resident at once, the sum of kernel durations against the span, and per kernel the mean duration.
usage: python scripts/overlap.py <kernel_trace.csv> [skip_first_seconds]"""
import csv
import re
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(e for _, e, _ in rows)
# keep the last 40 % of the trace: the timed, steady part of the bench
t0 = rows[0][0] + int((t_end - rows[0][0]) * 0.6)
rows = [r for r in rows if r[0] >= t0]
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
level, last, hist = 0, ev[0][0], defaultdict(int)
for t, d in ev:
    hist[level] += t - last
    last = t
    level += d
span = rows[-1][1] - rows[0][0] if rows else 1
tot = sum(e - s for s, e, _ in rows)
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  sum of durations {tot / 1e6:.2f} ms  ratio {tot / span:.2f}")
for k in sorted(hist):
    print(f"  {k} kernels resident: {100.0 * hist[k] / span:5.1f} %")
per = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    n = re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:56]
    per[n][0] += e - s
    per[n][1] += 1
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:56s} n={c:6d} mean={t / c / 1e3:9.1f} us  share of span {100.0 * t / span:5.1f} %")
