"""Per-call kernel time of the grouped (G-batch) launches in a bench.py kernel trace: launches are keyed by
(kernel, grid size); per kernel the grids that belong to grouped calls are the largest ones (the single-batch
launches of the untimed counting pass use smaller grids).  Prints the MEDIAN duration per (kernel, grid) for grids
with at least `min_calls` launches.  usage: python scripts/percall.py <kernel_trace.csv> [min_calls]"""
import collections, csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
by = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:40]
    if name.startswith("at::") or "rocprim" in name or name.startswith("__amd"):
        continue
    by[(name, int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = collections.defaultdict(list)
for (name, g), d in by.items():
    if len(d) >= min_calls:
        names[name].append((g, d))
tot = 0.0
out = []
for name, lst in names.items():
    lst.sort(reverse=True)
    gmax = lst[0][0]
    # grouped-call grids: within 64x of the kernel's largest grid and not the most frequent small one
    keep = [(g, d) for g, d in lst if len(d) <= 2 * min(len(x[1]) for x in lst)]
    t = sum(statistics.median(d) for g, d in keep) / 1e3
    out.append((t, name, [(g, len(d), round(statistics.median(d) / 1e3, 1)) for g, d in keep]))
for t, name, detail in sorted(out, reverse=True):
    tot += t
    print(f"{name:40s} per-call={t:8.1f} us   {detail}")
print(f"sum per call {tot:.1f} us")
