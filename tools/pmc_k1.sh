#!/bin/bash
# PMC passes for the three GEMMs of the dominant convolution (K1 through the C ABI): run ON THE GPU BOX from the repo root:
#   MT_GIT_COMMIT=<hash> bash tools/pmc_k1.sh [round4]   -> gpurun_out/pmc_<mode>_{f,w,s}/..., then tools/pmc_k1_json.py
#   builds profiles/<round>_k1_<mode>_pmc.json (bench.py cites the fwd one).
# Counters are collected in their own runs with --kernel-trace only (no other trace domains).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${1:-round4}
cd /tmp && export TMPDIR=/tmp
for mode in fwd dgrad wgrad; do
  for pass in "f:FETCH_SIZE" "w:WRITE_SIZE" "s:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
    n=${pass%%:*}; c=${pass#*:}
    rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${mode}_$n -o $n --output-format csv -- python3 $R/tools/bench_k1.py $mode > $R/gpurun_out/pmc_${mode}_$n.log 2>&1
  done
  (cd $R && python3 tools/pmc_k1_json.py $mode profiles/${ROUND}_k1_${mode}_pmc.json)
done
