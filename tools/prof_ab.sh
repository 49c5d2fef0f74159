#!/bin/bash
# kernel-time totals of the bench step with the persistent gather-GEMM off / on (run ON THE GPU BOX from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export MT_IGEMM_PERSIST=$v
  rm -rf $R/gpurun_out/prof_p$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_p$v -o p$v -- python3 $R/bench.py --no_cpu_baseline --no_extra --steps 6 --warmup 3 > $R/gpurun_out/prof_p$v.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $R/gpurun_out/prof_p$v -name "*.db" | head -1) 70 > $R/gpurun_out/prof_p$v.txt
  echo "persist=$v"; grep -E "igemm|TOTAL" $R/gpurun_out/prof_p$v.txt
done
