#!/bin/bash
# average duration of the kernels matching a pattern over a short eager bench run:  tools/kernel_time.sh <pattern> [MT_LIB_PATH]
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt
MT_LIB_PATH=${2:+$R/}$2 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o k -- python3 $R/bench.py --eager --no_cpu_baseline --no_extra --no_hbm_kernels --steps 4 --warmup 2 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(find $R/gpurun_out/kt -name '*.db' | head -1) 200 | grep -E "$1" | cut -c1-130
rm -rf $R/gpurun_out/kt
