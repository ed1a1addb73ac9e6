cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_batch.py tests/test_gpu_sequence_u8.py -x -q -m gpu 2>&1 | tail -4
python3 tools/batch_probe.py 240 5 1,2,4,8,16,32 2>&1 | tee gpurun_out/r04_batch_probe_240.txt
python3 tools/batch_probe.py 480 5 1,4,8,16 2>&1 | tee gpurun_out/r04_batch_probe_480.txt
python3 tools/batch_probe.py 960 5 1,2,4,8 2>&1 | tee gpurun_out/r04_batch_probe_960.txt
python3 tools/collection_probe.py 240 100 1,16 2>&1 | tee gpurun_out/r04_collection_probe_240.txt
python3 tools/collection_probe.py 480 64 1,16 2>&1 | tee gpurun_out/r04_collection_probe_480.txt
