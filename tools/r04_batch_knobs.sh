cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_batch.py tests/test_gpu_tiles.py -x -q -m gpu -k "not bench_tiles" 2>&1 | tail -4
python3 tools/batch_probe.py 240 8 1,16,32 2>&1 | tee gpurun_out/r04_batch_probe_240_L8.txt
python3 tools/batch_probe.py 240 15 1,16,32 2>&1 | tee gpurun_out/r04_batch_probe_240_L15.txt
PAPOF_SOR_RESIDENT=3072 python3 tools/batch_probe.py 240 5 16,32 2>&1 | tail -2
