set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_b32_prof -o run -- python3 tools/batch_probe.py 240 5 32 > gpurun_out/r04_batch32_under_rocprof.txt 2>&1
db=$(find gpurun_out/r04_b32_prof -name "*.db" | head -1)
python3 tools/rocpd_export.py stats $db gpurun_out/r04_batch32_kernel_stats.csv
python3 tools/kernel_avgs.py $db > gpurun_out/r04_batch32_kernel_avgs_by_grid.txt
rm -rf gpurun_out/r04_b32_prof
cat gpurun_out/r04_batch32_under_rocprof.txt | tail -2
head -14 gpurun_out/r04_batch32_kernel_stats.csv | cut -c1-150
