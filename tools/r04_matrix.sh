#!/bin/bash
# Round-4 measurement matrix (one box): the headline; every resolution on the reference schedule WITH the collection figures
# (unbatched sequences in flight and, for small frames, batches: csrc/batch.hip); config-4 schedule without; the other sweep
# orders at the sizes BASELINE.json names; deep pyramids; the batch and collection probes.  Outputs under gpurun_out/<tag>_*.
cd "$(dirname "$0")/.."
tag=${1:-r04}
out=gpurun_out
python3 bench.py --steps 20 --warmup 5 > $out/${tag}_bench_headline.json 2> $out/${tag}_bench_headline.err
: > $out/${tag}_bench_all_resolutions_reference.jsonl
for res in 240 480 960 1920; do
  python3 bench.py --res $res --schedule reference --steps 10 --warmup 2 2>/dev/null >> $out/${tag}_bench_all_resolutions_reference.jsonl
  echo "done reference $res"
done
: > $out/${tag}_bench_all_resolutions_cfg4.jsonl
for res in 240 480 960; do
  python3 bench.py --res $res --schedule cfg4 --steps 10 --warmup 2 --no-collection 2>/dev/null >> $out/${tag}_bench_all_resolutions_cfg4.jsonl
  echo "done cfg4 $res"
done
python3 bench.py --mode redblack --no-collection 2>/dev/null > $out/${tag}_bench_redblack.json
python3 bench.py --mode jacobi --res 480 --schedule reference --no-collection 2>/dev/null > $out/${tag}_bench_jacobi_480.json
python3 bench.py --mode redblack --res 960 --schedule reference --no-collection 2>/dev/null > $out/${tag}_bench_redblack_960.json
: > $out/${tag}_bench_deep_pyramids.jsonl
for res in 240 1920; do for lv in 8 15; do
  python3 bench.py --res $res --levels $lv --schedule reference --steps 5 --warmup 2 --no-cpu-baseline --no-collection 2>/dev/null >> $out/${tag}_bench_deep_pyramids.jsonl
done; done
python3 tools/batch_probe.py 240 5 1,2,4,8,16,32 > $out/${tag}_batch_probe_240.txt 2>&1
python3 tools/batch_probe.py 240 15 1,16,32 > $out/${tag}_batch_probe_240_L15.txt 2>&1
python3 tools/batch_probe.py 480 5 1,4,8,16 > $out/${tag}_batch_probe_480.txt 2>&1
python3 tools/collection_probe.py 240 100 1,16 > $out/${tag}_collection_probe_240.txt 2>&1
echo matrix done
