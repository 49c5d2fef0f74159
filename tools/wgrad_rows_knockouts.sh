#!/bin/bash
# per-shape kernel time of the row walker (tools/bench_wgrad_rows.py, rows arm) under the product library and knock-out builds
# (tools/diag_build.sh wr_X "-DMT_WR_EXP_X" wgrad_rows_kernel.hip): which phase of the kernel bounds it
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for v in product "$@"; do
  rm -rf $R/gpurun_out/wk
  if [ $v = product ]; then unset MT_LIB_PATH; else export MT_LIB_PATH=$R/_exp/libmt_wr_$v.so; fi
  MT_BENCH_ONLY=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wk -o e -- python3 $R/tools/bench_wgrad_rows.py 1 > $R/gpurun_out/wk.log 2>&1
  echo "== $v"
  python3 - $(find $R/gpurun_out/wk -name '*.db' | head -1) <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, d.end - d.start, d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
rows = [(n, d) for n, d, _ in rows if "wgrad_rows" in n]
per = 22
out = []
for i in range(0, len(rows), per):
    ch = sorted(d for _, d in rows[i:i + per])
    out.append(f"{ch[len(ch) // 2] / 1e3:6.1f}")
print(" ".join(out), " us (median per shape, bench order)")
PY
  rm -rf $R/gpurun_out/wk
done
