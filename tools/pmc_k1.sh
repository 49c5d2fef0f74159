#!/bin/bash
# PMC passes for the dominant kernel (K1 forward through the C ABI): run ON THE GPU BOX from the repo root:
#   bash tools/pmc_k1.sh            -> gpurun_out/pmc_{f,w,s}/..., then tools/pmc_k1_json.py builds the summary.
# Counters are collected in their own runs with --kernel-trace only (no other trace domains).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE" "s:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  n=${pass%%:*}; c=${pass#*:}
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$n -o $n --output-format csv -- python3 $R/tools/bench_k1.py > $R/gpurun_out/pmc_$n.log 2>&1
done
cd $R && python3 tools/pmc_k1_json.py ${1:-profiles/round2_k1_fwd_pmc.json}
