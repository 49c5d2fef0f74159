#!/usr/bin/env python3
"""Per-kernel time summary from a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats writes
<name>_results.db on ROCm 7.2).  Usage: tools/rocpd_stats.py results.db [top_n]"""
import re
import sqlite3
import sys


def main():
    db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    by_grid = len(sys.argv) > 3 and sys.argv[3] == "grid"   # split every kernel by its grid size
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, d.end - d.start, d.grid_size_x, d.workgroup_size_x from {kd} d "
                     f"join {ks} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, dur, gx, wx in rows:
        name = re.sub(r"\(.*", "", name)
        if by_grid:
            name = f"{name[:48]} g{gx // max(wx, 1)}"
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:72]:72s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.1f} {a[2] / 1e3:9.1f} {a[3] / 1e3:9.1f} "
              f"{100.0 * a[1] / total:6.2f}")
    print(f"{'TOTAL':72s} {len(rows):7d} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main()
