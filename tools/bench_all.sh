#!/bin/bash
# bench.py at the four widths of the HoChiMinhTraffic sets, both schedules, WITH the CPU baseline leg (north_star:
# "throughput at 240/480/960/1920 ... next to the reference Code/Serial C++ path timed on the same box's host cores").
cd "$(dirname "$0")/.."
tag=${1:-r02}
for sched in cfg4 reference; do
  : > gpurun_out/${tag}_bench_all_resolutions_${sched}.jsonl
  for res in 240 480 960 1920; do
    python3 bench.py --res $res --schedule $sched --steps 10 --warmup 2 2>/dev/null >> gpurun_out/${tag}_bench_all_resolutions_${sched}.jsonl
    echo "done $sched $res"
  done
done
python3 bench.py --mode redblack 2>/dev/null > gpurun_out/${tag}_bench_redblack.json
python3 bench.py --mode redblack --schedule reference 2>/dev/null > gpurun_out/${tag}_bench_redblack_reference_schedule.json
python3 bench.py --mode jacobi --res 480 --schedule reference 2>/dev/null > gpurun_out/${tag}_bench_jacobi_480.json
python3 bench.py --mode redblack --res 960 --schedule reference 2>/dev/null > gpurun_out/${tag}_bench_redblack_960.json
