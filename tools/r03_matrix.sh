#!/bin/bash
# Round-3 measurement matrix (one box): the headline, every resolution on both schedules with the CPU baseline, the other
# sweep orders at the sizes BASELINE.json names, deep pyramids.  Outputs under gpurun_out/<tag>_*.
cd "$(dirname "$0")/.."
tag=${1:-r03}
out=gpurun_out
python3 bench.py --steps 20 --warmup 5 > $out/${tag}_bench_headline.json 2> $out/${tag}_bench_headline.err
for sched in cfg4 reference; do
  : > $out/${tag}_bench_all_resolutions_${sched}.jsonl
  for res in 240 480 960 1920; do
    python3 bench.py --res $res --schedule $sched --steps 10 --warmup 2 --no-collection 2>/dev/null >> $out/${tag}_bench_all_resolutions_${sched}.jsonl
    echo "done $sched $res"
  done
done
python3 bench.py --mode redblack --no-collection 2>/dev/null > $out/${tag}_bench_redblack.json
python3 bench.py --mode redblack --schedule reference --no-collection 2>/dev/null > $out/${tag}_bench_redblack_reference_schedule.json
python3 bench.py --mode jacobi --res 480 --schedule reference --no-collection 2>/dev/null > $out/${tag}_bench_jacobi_480.json
python3 bench.py --mode redblack --res 960 --schedule reference --no-collection 2>/dev/null > $out/${tag}_bench_redblack_960.json
: > $out/${tag}_bench_deep_pyramids.jsonl
for res in 240 1920; do for lv in 8 15; do
  python3 bench.py --res $res --levels $lv --schedule reference --steps 5 --warmup 2 --no-cpu-baseline --no-collection 2>/dev/null >> $out/${tag}_bench_deep_pyramids.jsonl
done; done
echo matrix done
