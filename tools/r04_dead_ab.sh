cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_small.py tests/test_gpu_batch.py tests/test_gpu_sequence_u8.py tests/test_gpu_strips.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
for dd in 0 1; do
  PAPOF_SOR_DEAD=$dd python bench.py --no-cpu-baseline --no-collection --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('DEAD=$dd', d['value'], d['ms_per_step'], d['roofline']['frac'], d['full_sha_equal'], [e['avg_launch_us'] for e in d['roofline']['by_level']])"
done; done
for dd in 0 1; do
PAPOF_SOR_DEAD=$dd python bench.py --schedule reference --no-cpu-baseline --no-collection 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('ref schedule DEAD=$dd', d['value'], d['ms_per_step'], d['roofline']['frac'], d['full_sha_equal'])"
done
