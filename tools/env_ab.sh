#!/bin/bash
# same-box A/B of environment knobs: tools/env_ab.sh "VAR=a" "VAR=b" ... [ROUNDS=3]; each setting runs bench.py alternately
set -e
mkdir -p gpurun_out
for r in $(seq 1 ${ROUNDS:-3}); do
  for kv in "$@"; do
    env $kv timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-40s ms_per_step %.3f value %.2f duv %s sor_ms %.3f call_incl %s' % ('$kv', d['ms_per_step'], d['value'], d.get('max_abs_duv_vs_reference'), r['sor_ms_per_step'], d.get('value_call_inclusive')))
" | tee -a gpurun_out/env_ab.txt
  done
done
