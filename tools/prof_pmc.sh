#!/bin/bash
# one rocprofv3 --pmc pass of the default bench command; usage: tools/prof_pmc.sh <tag> "<COUNTERS...>" [bench args]
tag=$1; ctrs=$2; shift 2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 --pmc $ctrs -d gpurun_out/${tag} -o run -- python3 bench.py --no-cpu-baseline --no-collection --steps 3 --warmup 1 "$@" > /dev/null 2> gpurun_out/${tag}.err
python3 - <<PY
import sqlite3
c=sqlite3.connect("gpurun_out/${tag}/run_results.db")
rows=c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration), grid_size from counters_collection group by kernel_name, counter_name, grid_size order by kernel_name, grid_size desc, counter_name").fetchall()
for r in rows:
    n=r[0].replace('papof::(anonymous namespace)::','').replace('void ','').split('(')[0]
    print("%-40s grid %-9d %-28s n=%-4d avg=%14.1f dur_us=%8.1f" % (n[:40], r[5], r[1], r[2], r[3], r[4]/1e3))
PY
rm -rf gpurun_out/${tag}   # the database is tens of MB; the table printed above is what is kept
