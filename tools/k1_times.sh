#!/bin/bash
# K1 forward / data gradient / weight gradient: kernel durations of the product library (rocprofv3 kernel trace of tools/bench_k1.py)
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for mode in fwd dgrad wgrad; do
  rm -rf $R/gpurun_out/k1_$mode
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/k1_$mode -o e -- python3 $R/tools/bench_k1.py $mode > /dev/null 2>&1
  echo "$mode: $(python3 $R/tools/rocpd_stats.py $(find $R/gpurun_out/k1_$mode -name '*.db' | head -1) 5 | grep -E 'pipe_patch|wgrad_pipe' | cut -c1-130)"
  rm -rf $R/gpurun_out/k1_$mode
done
