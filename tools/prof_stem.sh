set -e
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stem
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stem -o st -- python3 $R/bench.py --no_cpu_baseline --no_extra --steps 6 --warmup 3 > $R/gpurun_out/prof_stem.log 2>&1
cd $R
python3 tools/rocpd_stats.py $(find gpurun_out/prof_stem -name "*.db" | head -1) 90 > gpurun_out/prof_stem.txt
grep -E "stem|TOTAL|igemm_kernelILb1ELi64|wgrad_kernelILb1ELb1|unpack_wave" gpurun_out/prof_stem.txt
