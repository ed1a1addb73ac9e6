#!/bin/bash
# The batch figures of round 4 (profiles/r04_batch_probe_after_own_fill.txt, r04_collection_probe_after_own_fill.txt): ms per pair
# of papof_flow_batch_u8 by batch size, and of flow_collection() unbatched / batched, on one box.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.."
out=gpurun_out
{
  python3 tools/batch_probe.py 240 5 1,2,4,8,16,32
  python3 tools/batch_probe.py 240 8 32
  python3 tools/batch_probe.py 240 15 16,32
  python3 tools/batch_probe.py 480 5 1,4,8,16
} > $out/r04_batch_probe.txt 2>&1
{
  python3 tools/collection_probe.py 240 100 1,16
  python3 tools/collection_probe.py 480 64 1,16
} > $out/r04_collection_probe.txt 2>&1
# what the knobs of the solver do to a batch (XCD affinity is what matters: sor.hip, sor_solve)
for env in "X=1" "PAPOF_SOR_DEPTH=10" "PAPOF_SOR_RESIDENT=3072" "PAPOF_SOR_DEAD=0" "PAPOF_SOR_TINY=0"; do
  echo "== $env"; env $env python3 tools/batch_probe.py 240 5 16,32 2>&1 | tail -2
done > $out/r04_batch_knobs.txt 2>&1
echo done
