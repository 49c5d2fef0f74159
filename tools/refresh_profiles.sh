# Run ON THE GPU BOX from the repo root:  MT_GIT_COMMIT=<hash> bash tools/refresh_profiles.sh [round3] [v11]
# -> gpurun_out/{bench_default.json, <ver>_kernel_stats*.txt, layer_table*.txt}, profiles/<round>_k1_*_pmc.json
set -e
R=$(pwd)
ROUND=${1:-round4}
VER=${2:-v17}
echo "== bench default (driver-style: no flags)"
timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.json; echo
echo "== kernel stats (single-scale, the headline of rounds 1-2, and the multi-scale headline)"
cd /tmp && export TMPDIR=/tmp
for v in ss ms; do
  rm -rf $R/gpurun_out/prof_${VER}_$v
  flag=""; [ $v = ss ] && flag="--single_scale"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${VER}_$v -o ${VER} -- python3 $R/bench.py --eager --no_cpu_baseline --no_extra --no_hbm_kernels --steps 6 --warmup 3 $flag > $R/gpurun_out/prof_${VER}_$v.log 2>&1
  DB=$(find $R/gpurun_out/prof_${VER}_$v -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $DB 80 > $R/gpurun_out/${VER}_${v}_kernel_stats.txt
  python3 $R/tools/rocpd_stats.py $DB 120 grid > $R/gpurun_out/${VER}_${v}_kernel_stats_by_grid.txt
  rm -rf $R/gpurun_out/prof_${VER}_$v
done
cd $R
head -12 gpurun_out/${VER}_ss_kernel_stats.txt
echo "== layer tables"
timeout -k 10 300 python tools/layer_table.py --steps 4 > gpurun_out/layer_table.txt 2>/dev/null; tail -n 1 gpurun_out/layer_table.txt
timeout -k 10 300 python tools/layer_table.py --steps 4 --ms_dis > gpurun_out/layer_table_ms.txt 2>/dev/null; tail -n 1 gpurun_out/layer_table_ms.txt
echo "== PMC passes of the K1 trio"
bash tools/pmc_k1.sh $ROUND > gpurun_out/pmc_k1.log 2>&1 || tail -5 gpurun_out/pmc_k1.log
ls profiles/${ROUND}_k1_*_pmc.json && cp profiles/${ROUND}_k1_*_pmc.json gpurun_out/
rm -rf gpurun_out/pmc_*_f gpurun_out/pmc_*_w gpurun_out/pmc_*_s

echo "== fabric traffic per kernel of the shipped binary (FETCH_SIZE / WRITE_SIZE passes over 3 eager steps)"
cd /tmp && export TMPDIR=/tmp
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
  n=${pass%%:*}; c=${pass#*:}
  rm -rf $R/gpurun_out/pt_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pt_$n -o $n --output-format csv -- python3 $R/bench.py --eager --no_cpu_baseline --no_extra --no_hbm_kernels --steps 2 --warmup 1 > $R/gpurun_out/pt_$n.log 2>&1 || tail -3 $R/gpurun_out/pt_$n.log
done
cd $R && python3 tools/pmc_traffic.py gpurun_out/pt > gpurun_out/fabric_traffic_by_kernel.txt && head -5 gpurun_out/fabric_traffic_by_kernel.txt
rm -rf gpurun_out/pt_f gpurun_out/pt_w
