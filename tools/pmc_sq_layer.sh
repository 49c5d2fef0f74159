#!/bin/bash
# SQ counters (wave cycles, waits, issue, MFMA busy, LDS conflicts) of one layer's forward kernel: run ON THE GPU BOX from the repo
# root:   bash tools/pmc_sq_layer.sh TAG N Cin H Cout k stride [T]   -> prints per-kernel averages, files under gpurun_out/pmcs_TAG_*
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for pass in "a:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "b:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "c:FETCH_SIZE" "d:WRITE_SIZE"; do
  n=${pass%%:*}; c=${pass#*:}
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcs_${TAG}_$n -o $n --output-format csv -- python3 $R/tools/layer_fwd.py "$@" > $R/gpurun_out/pmcs_${TAG}_$n.log 2>&1 || { tail -5 $R/gpurun_out/pmcs_${TAG}_$n.log; }
done
cd $R && python3 - <<PY
import csv, glob, collections
for n in "abcd":
    g = glob.glob("gpurun_out/pmcs_${TAG}_%s/**/*counter_collection.csv" % n, recursive=True)
    if not g: print("pass", n, "missing"); continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(g[0])):
        if "igemm" in r["Kernel_Name"] or "wsreg" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(d.items()):
        print("${TAG}", k, c, "avg %.4g over %d launches" % (sum(v) / len(v), len(v)))
    t = glob.glob("gpurun_out/pmcs_${TAG}_%s/**/*kernel_trace.csv" % n, recursive=True)
    if t and n == "a":
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0])):
            if "igemm" in r["Kernel_Name"] or "wsreg" in r["Kernel_Name"]:
                d[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in d.items():
            print("${TAG}", k, "duration avg %.1f us over %d" % (sum(v) / len(v), len(v)))
PY
