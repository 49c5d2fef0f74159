#!/bin/bash
# L2 / fabric counters of one layer's forward kernel: run ON THE GPU BOX from the repo root:
#   bash tools/pmc_layer.sh TAG N Cin H Cout k stride [T]   -> gpurun_out/pmcl_TAG_*.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for pass in "a:FETCH_SIZE" "b:TCC_HIT_sum" "c:TCC_MISS_sum"; do
  n=${pass%%:*}; c=${pass#*:}
  echo "pass $n: $c"
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcl_${TAG}_$n -o $n --output-format csv -- python3 $R/tools/layer_fwd.py "$@" > $R/gpurun_out/pmcl_${TAG}_$n.log 2>&1 || { tail -5 $R/gpurun_out/pmcl_${TAG}_$n.log; }
done
cd $R && python3 - <<PY
import csv, glob, collections
for n in "abc":
    g = glob.glob("gpurun_out/pmcl_${TAG}_%s/**/*counter_collection.csv" % n, recursive=True)
    if not g: print("pass", n, "missing"); continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(g[0])):
        if "igemm" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(d.items()):
        print("${TAG}", k, c, "avg %.1f over %d launches" % (sum(v) / len(v), len(v)))
PY
