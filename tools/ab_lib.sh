#!/bin/bash
# Same-box A/B of one micro-benchmark under two builds of the library, per-kernel times from rocprofv3 (run ON THE GPU BOX from the
# repo root):  tools/ab_lib.sh "<kernel regex>" tools/bench_k1_wgrad.py [args]     arms: product library, then every _exp/libmt_*.so
# named in $AB_LIBS (space-separated names, e.g. AB_LIBS="wold")
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for arm in product $AB_LIBS; do
  if [ "$arm" = product ]; then unset MT_LIB_PATH MT_DIAG_LIB; else export MT_LIB_PATH=$R/_exp/libmt_$arm.so MT_DIAG_LIB=$R/_exp/libmt_$arm.so; fi
  rm -rf $R/gpurun_out/ab_$arm
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_$arm -o ab -- python3 $R/$1 "${@:2}" > $R/gpurun_out/ab_$arm.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find $R/gpurun_out/ab_$arm -name "*.db" | head -1) 70 > $R/gpurun_out/ab_$arm.txt
  echo "== $arm (rep $rep)"; grep -E "$PAT" $R/gpurun_out/ab_$arm.txt || true
done
done
