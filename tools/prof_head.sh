#!/bin/bash
# rocprofv3 evidence of the headline at HEAD: bench line, the same command under --kernel-trace --stats (exported to csv),
# FETCH_SIZE / WRITE_SIZE passes (exported), databases deleted.   usage: tools/prof_head.sh <tag>
set -e
tag=$1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o run -- python3 bench.py --no-cpu-baseline --steps 5 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.err
python3 tools/rocpd_export.py stats $(find $out/${tag}_prof -name "*.db" | head -1) $out/${tag}_kernel_stats.csv
python3 tools/timeline_gaps.py $(find $out/${tag}_prof -name "*.db" | head -1) > $out/${tag}_timeline_one_call.txt || true
rm -rf $out/${tag}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/${tag}_pmc -o run -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $out/${tag}_pmc.err
  python3 tools/rocpd_export.py pmc $(find $out/${tag}_pmc -name "*.db" | head -1) $c $out/${tag}_pmc_$c.csv
  rm -rf $out/${tag}_pmc
done
head -12 $out/${tag}_kernel_stats.csv
