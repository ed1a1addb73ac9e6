cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_widths or one_kernel_and_two" 2>&1 | tail -3
for i in 1 2; do
for tx in 16 32; do
  PAPOF_FS_TX=$tx python bench.py --no-cpu-baseline --no-collection --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('TX=$tx', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
PAPOF_FS_TX=32 python tools/batch_probe.py 240 5 16,32 | tail -2
PAPOF_FS_TX=16 python tools/batch_probe.py 240 5 16,32 | tail -2
