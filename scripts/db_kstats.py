"""per-kernel totals from a rocprofv3 results .db (when only the sqlite output was kept)
usage: python scripts/db_kstats.py <results.db> [top_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
tot = db.execute(f"select sum(end-start)/1e3 from {kd}").fetchone()[0]
q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit {top}")
for r in db.execute(q):
    print(f"{r[0][:100]:100s} n={r[1]:6d} tot={r[2]:10.0f}us avg={r[3]:8.2f}us {100 * r[2] / tot:5.1f}%")
