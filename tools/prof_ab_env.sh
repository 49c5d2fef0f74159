#!/bin/bash
# Same-box A/B of the bench step's KERNEL times under two environment settings (run ON THE GPU BOX from the repo root):
#   tools/prof_ab_env.sh "MT_X=0" "MT_X=1" [extra bench flags]   -> per-kernel totals of both arms side by side, largest differences first
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$1" "$2"; do
  rm -rf $R/gpurun_out/pab_$i
  # (a subshell per arm: the setting of one arm must not leak into the other)
  ( export $v; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pab_$i -o p -- python3 $R/bench.py --eager --no_cpu_baseline --no_extra --no_hbm_kernels --steps 6 --warmup 3 $3 > $R/gpurun_out/pab_$i.log 2>&1 )
  python3 $R/tools/rocpd_stats.py $(find $R/gpurun_out/pab_$i -name "*.db" | head -1) 400 > $R/gpurun_out/pab_$i.txt
  i=$((i+1))
done
python3 - $R/gpurun_out/pab_0.txt $R/gpurun_out/pab_1.txt <<'PY'
import sys
def rd(p):
    d = {}
    for ln in open(p).read().splitlines()[1:]:
        f = ln.split()
        if len(f) >= 7: d[f[0]] = (int(f[1]), float(f[2]))
        elif f and f[0] == "TOTAL": d["TOTAL"] = (int(f[1]), float(f[2]))
    return d
a, b = rd(sys.argv[1]), rd(sys.argv[2])
rows = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
print(f"{'kernel':60s} {'calls A':>8s} {'ms A':>9s} {'calls B':>8s} {'ms B':>9s} {'B-A ms/step':>12s}   (9 steps)")
for k in rows[:25]:
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    print(f"{k[:60]:60s} {ca:8d} {ta:9.3f} {cb:8d} {tb:9.3f} {(tb - ta) / 9:12.3f}")
PY
