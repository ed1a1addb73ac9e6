#!/bin/bash
# same-box A/B of two builds of libpapof.so: alternating bench.py runs of the headline (PAPOF_LIB selects the build)
# usage: tools/ab_bench.sh <other lib> [rounds]
set -e
other=$1; rounds=${2:-3}
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for lib in "" "$other"; do
    PAPOF_LIB=$lib timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-32s ms_per_step %.3f value %.2f duv %s sor_ms %.3f call_incl %s' % ('$lib' or 'this build', d['ms_per_step'], d['value'], d.get('max_abs_duv_vs_reference'), r['sor_ms_per_step'], d.get('value_call_inclusive')))
" | tee -a gpurun_out/ab_bench.txt
  done
done
