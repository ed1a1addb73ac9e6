#!/bin/bash
# In-pair re-check of the solver's knobs after the dead lanes left the bus (same box, alternating with the default)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for env in "X=1" "PAPOF_SOR_DEPTH=6" "PAPOF_SOR_DEPTH=10" "PAPOF_SOR_XCD=0" "PAPOF_SOR_XCD=2" "PAPOF_SOR_FUSE=1" "PAPOF_SOR_FUSE=2" "PAPOF_OVERLAP=0"; do
  env $env python3 bench.py --no-cpu-baseline --no-collection --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('%-22s' % '$env', d['ms_per_step'], d['roofline']['frac'], [e['avg_launch_us'] for e in d['roofline']['by_level']])"
done; done
