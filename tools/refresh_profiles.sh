set -e
R=$(pwd)
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== bench default"
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json; echo
echo "== kernel stats"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_v9
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v9 -o v9 -- python3 $R/bench.py --no_cpu_baseline --no_extra --steps 6 --warmup 3 > $R/gpurun_out/prof_v9.log 2>&1
cd $R
DB=$(find gpurun_out/prof_v9 -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 80 > gpurun_out/v9_kernel_stats.txt
python3 tools/rocpd_stats.py $DB 120 grid > gpurun_out/v9_kernel_stats_by_grid.txt
head -12 gpurun_out/v9_kernel_stats.txt
echo "== layer tables"
timeout -k 10 300 python tools/layer_table.py --steps 4 > gpurun_out/layer_table.txt 2>&1; tail -n 1 gpurun_out/layer_table.txt
timeout -k 10 300 python tools/layer_table.py --steps 4 --ms_dis > gpurun_out/layer_table_ms.txt 2>&1; tail -n 1 gpurun_out/layer_table_ms.txt
