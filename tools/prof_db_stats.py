"""Per-kernel totals of a rocprofv3 results database (when --stats left no CSV): python tools/prof_db_stats.py file.db [filter]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = collections.defaultdict(list)
for n, d in rows:
    agg[n.split("(")[0]].append(d / 1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tot = sum(sum(v) for v in agg.values())
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if flt in n:
        print(f"{n[:90]:90s} {len(v):6d} {sum(v) / 1e3:9.3f} ms  avg {sum(v) / len(v):8.1f} us  min {min(v):7.1f}  max {max(v):7.1f}")
print(f"TOTAL {tot / 1e3:.3f} ms over {len(rows)} dispatches")
