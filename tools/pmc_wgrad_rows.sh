#!/bin/bash
# SQ counters of the row-walker weight gradient over tools/bench_wgrad_rows.py (rows arm, one problem per launch): two separate --pmc
# passes (wave cycles / waits / issue / MFMA busy / LDS; instruction counts by class).  Run ON THE GPU BOX from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "a:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "b:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  n=${pass%%:*}; c=${pass#*:}
  rm -rf $R/gpurun_out/pmcwr_$n
  MT_BENCH_ONLY=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcwr_$n -o $n --output-format csv -- python3 $R/tools/bench_wgrad_rows.py 1 > $R/gpurun_out/pmcwr_$n.log 2>&1 || { tail -5 $R/gpurun_out/pmcwr_$n.log; }
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for n in "ab":
    g = glob.glob("gpurun_out/pmcwr_%s/**/*counter_collection.csv" % n, recursive=True)
    if not g: print("pass", n, "missing"); continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(g[0])):
        if "wgrad_rows" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(d.items()):
        print(k, c, "avg %.4g over %d launches" % (sum(v) / len(v), len(v)))
PY
rm -rf gpurun_out/pmcwr_a gpurun_out/pmcwr_b
