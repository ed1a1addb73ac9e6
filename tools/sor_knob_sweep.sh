#!/bin/bash
# isolated 30-sweep solves at the five 1080p level sizes under a list of knob settings (tools/sor_tool.py bench, exact order)
mkdir -p gpurun_out
for kv in "PAPOF_X=0" "PAPOF_SOR_XCD=2" "PAPOF_SOR_XCD=0" "PAPOF_SOR_FUSE=2" "PAPOF_SOR_FUSE=1" "PAPOF_SOR_DEPTH=6" "PAPOF_SOR_DEPTH=10" "PAPOF_SOR_FUSE=2 PAPOF_SOR_XCD=2"; do
  echo "== $kv" | tee -a gpurun_out/sor_knob_sweep.txt
  env $kv timeout -k 10 200 python tools/sor_tool.py bench 0 2>&1 | head -5 | tee -a gpurun_out/sor_knob_sweep.txt
done
