#!/usr/bin/env python3
"""usage: records_timeline.py <kernel_trace.csv> — per call of gigl_records_encode (record_plan -> record_scan -> record_write
back to back on one stream): mean duration of each kernel and the idle gaps between them, from rocprofv3's kernel trace"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    k = next((n for n in ("record_plan", "record_scan", "record_write") if n in name), None)
    if k:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
calls, cur = [], []
for s, e, k in rows:
    if k == "record_plan" and cur:
        calls.append(cur)
        cur = []
    cur.append((s, e, k))
if cur:
    calls.append(cur)
calls = [c for c in calls if [k for _, _, k in c] == ["record_plan", "record_scan", "record_write"]][2:]
n = len(calls)
dur = {k: 0.0 for k in ("record_plan", "record_scan", "record_write")}
gap_ps = gap_sw = gap_next = 0.0
for i, c in enumerate(calls):
    for s, e, k in c:
        dur[k] += (e - s) / 1e3
    gap_ps += (c[1][0] - c[0][1]) / 1e3
    gap_sw += (c[2][0] - c[1][1]) / 1e3
    if i + 1 < n:
        gap_next += (calls[i + 1][0][0] - c[2][1]) / 1e3
print(f"{n} calls: plan {dur['record_plan']/n:.1f} us, gap {gap_ps/n:.1f}, scan {dur['record_scan']/n:.1f}, gap {gap_sw/n:.1f}, "
      f"write {dur['record_write']/n:.1f}, gap to the next call's plan {gap_next/max(n-1,1):.1f}; "
      f"call to call {(calls[-1][0][0]-calls[0][0][0])/1e3/max(n-1,1):.1f} us")
