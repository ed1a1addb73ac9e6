#!/bin/bash
# rocprofv3 evidence for one bench.py command line: kernel-trace statistics + two PMC passes (FETCH_SIZE, WRITE_SIZE).
# usage: tools/prof_bench.sh <tag> <bench args...>      outputs under gpurun_out/<tag>_*
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out
python3 bench.py "$@" --no-cpu-baseline --no-collection > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o run -- python3 bench.py "$@" --no-cpu-baseline --no-collection --steps 5 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.err
rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o run -- python3 bench.py "$@" --no-cpu-baseline --no-collection --steps 3 --warmup 1 > /dev/null 2> $out/${tag}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o run -- python3 bench.py "$@" --no-cpu-baseline --no-collection --steps 3 --warmup 1 > /dev/null 2> $out/${tag}_pmc_write.err
find $out/${tag}_prof $out/${tag}_pmc_fetch $out/${tag}_pmc_write -name "*.csv" | head -20
