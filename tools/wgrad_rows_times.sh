#!/bin/bash
# kernel durations of the weight gradients of tools/bench_wgrad_rows.py with the row walker off / on (rocprofv3 kernel trace, by grid)
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for arm in 0 1; do
  rm -rf $R/gpurun_out/wr_$arm
  MT_BENCH_ONLY=$arm rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wr_$arm -o e -- python3 $R/tools/bench_wgrad_rows.py 1 > $R/gpurun_out/wr_prof_$arm.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $R/gpurun_out/wr_$arm -name '*.db' | head -1) 40 grid | grep -E "wgrad|unpack|kernel  " | cut -c1-140 > $R/gpurun_out/wr_kstats_$arm.txt
  rm -rf $R/gpurun_out/wr_$arm
done
