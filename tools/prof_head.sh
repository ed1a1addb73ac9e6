#!/bin/bash
# rocprofv3 evidence of the headline at HEAD, one box: the default bench line, the same workload under --kernel-trace --stats
# (exported to csv, per-grid averages, time line of one call), FETCH_SIZE / WRITE_SIZE passes in runs of their own (exported,
# and folded into profiles/pmc_traffic.json), the same for the red-black mode's dominant kernel.  Databases are deleted.
# usage: tools/prof_head.sh <tag>      (on the GPU box: cd /tmp && export TMPDIR=/tmp first; outputs gpurun_out/<tag>_*)
set -e
tag=$1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o run -- python3 bench.py --no-cpu-baseline --no-collection --steps 8 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.err
db=$(find $out/${tag}_prof -name "*.db" | head -1)
python3 tools/rocpd_export.py stats $db $out/${tag}_kernel_stats.csv
python3 tools/kernel_avgs.py $db > $out/${tag}_kernel_avgs_by_grid.txt
python3 tools/call_timeline.py $db 4 > $out/${tag}_timeline_one_call.txt || true
python3 tools/timeline_gaps.py $db 4 > $out/${tag}_timeline_gaps.txt || true
rm -rf $out/${tag}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/${tag}_pmc_$c -o run -- python3 bench.py --no-cpu-baseline --no-collection --steps 3 --warmup 1 > /dev/null 2> $out/${tag}_pmc.err
  python3 tools/rocpd_export.py pmc $(find $out/${tag}_pmc_$c -name "*.db" | head -1) $c $out/${tag}_pmc_$c.csv
done
python3 tools/pmc_traffic.py $(find $out/${tag}_pmc_FETCH_SIZE -name "*.db" | head -1) $(find $out/${tag}_pmc_WRITE_SIZE -name "*.db" | head -1) 1920_cfg4_exact k_sor_
rm -rf $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/${tag}_rbpmc_$c -o run -- python3 bench.py --mode redblack --no-cpu-baseline --no-collection --steps 3 --warmup 1 > /dev/null 2> $out/${tag}_pmc.err
done
python3 tools/pmc_traffic.py $(find $out/${tag}_rbpmc_FETCH_SIZE -name "*.db" | head -1) $(find $out/${tag}_rbpmc_WRITE_SIZE -name "*.db" | head -1) 1920_cfg4_redblack k_sor_blocked
rm -rf $out/${tag}_rbpmc_FETCH_SIZE $out/${tag}_rbpmc_WRITE_SIZE
cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
head -12 $out/${tag}_kernel_stats.csv
