"""Per-call kernel time of the grouped launches in a bench.py kernel trace: for each library kernel take the
N longest launches (N = number of grouped calls x launches per call) — the single-batch launches of the untimed
counting pass are the short ones.  usage: python scripts/percall.py <kernel_trace.csv> <n_calls>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_calls = int(sys.argv[2])
by = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:40]
    by[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, d in sorted(by.items(), key=lambda kv: -sum(sorted(kv[1], reverse=True)[:2 * n_calls])):
    if k.startswith("at::") or "rocprim" in k or k.startswith("__amd") or "uniq" in k or "csc_" in k or "coo_" in k or "table_" in k or "maxdeg" in k:
        continue
    per = 2 if k in ("expand_kernel", "linear_lds_kernel<2>", "insert_slots_kernel") else 1
    top = sorted(d, reverse=True)[:per * n_calls]
    t = sum(top) / n_calls / 1e3
    tot += t
    print(f"{k:40s} launches/call={per} per-call={t:8.1f} us")
print(f"sum per call {tot:.1f} us")
