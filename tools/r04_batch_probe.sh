set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/batch_probe.py 240 5 1,2,4,8,16,32 > gpurun_out/r04_batch_probe_240.txt 2>&1
cat gpurun_out/r04_batch_probe_240.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r04_bp_prof -o run -- python3 tools/batch_probe.py 240 5 16 > gpurun_out/r04_batch_probe_rocprof.txt 2>&1
db=$(find gpurun_out/r04_bp_prof -name "*.db" | head -1)
python3 tools/rocpd_export.py stats $db gpurun_out/r04_batch16_kernel_stats.csv
python3 tools/kernel_avgs.py $db > gpurun_out/r04_batch16_kernel_avgs_by_grid.txt
head -30 gpurun_out/r04_batch16_kernel_avgs_by_grid.txt
rm -rf gpurun_out/r04_bp_prof
