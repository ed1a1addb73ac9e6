#!/bin/bash
# same-box comparison of this build with variant builds (tools/ab_build.sh): tools/ab_libs.sh <rounds> <lib> [<lib> ...]
rounds=$1; shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $rounds); do
  for lib in "" "$@"; do
    PAPOF_LIB=$lib python3 bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-collection 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-36s ms_per_step %.4f duv %s' % ('$lib' or 'this build', d['ms_per_step'], d.get('max_abs_duv_vs_reference')))"
  done
done
