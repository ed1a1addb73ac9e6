#!/bin/bash
# same-box A/B of the strips orchestration (PAPOF_STRIPS = 1 off | 2 | 3 | 4): bench.py headline, a few steps each
set -e
mkdir -p gpurun_out
for s in ${STRIPS_LIST:-1 2 1 2 3 4}; do
  echo "== PAPOF_STRIPS=$s" | tee -a gpurun_out/strips_ab.txt
  PAPOF_STRIPS=$s timeout -k 10 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'duv', d.get('max_abs_duv_vs_reference'), 'sor_ms', r.get('sor_ms_per_step'), 'launches', r.get('launches_per_step'), 'frac', r.get('frac'))
" | tee -a gpurun_out/strips_ab.txt
done
