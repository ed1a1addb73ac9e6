set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/collection_trace.py 240 12 1,4,8,16,24 > gpurun_out/r04_collection_trace_240.txt 2>&1
cat gpurun_out/r04_collection_trace_240.txt
rocprofv3 --kernel-trace -d gpurun_out/r04_ct_prof -o run -- python3 tools/collection_trace.py 240 12 16 > gpurun_out/r04_collection_trace_240_under_rocprof.txt 2>&1
db=$(find gpurun_out/r04_ct_prof -name "*.db" | head -1)
python3 tools/trace_concurrency.py $db 0.4 > gpurun_out/r04_collection_concurrency_240x16.txt
cat gpurun_out/r04_collection_concurrency_240x16.txt
rm -rf gpurun_out/r04_ct_prof
