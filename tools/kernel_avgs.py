#!/usr/bin/env python3
"""Average duration per kernel and grid size from a rocprofv3 --kernel-trace database (run_results.db).
usage: kernel_avgs.py run_results.db [name-substring ...]"""
import sqlite3
import sys

db, pats = sys.argv[1], sys.argv[2:]
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(end-start), min(end-start) from kernels "
                 "group by name, grid_x, grid_y, grid_z order by name, grid_x*grid_y*grid_z desc").fetchall()
for name, gx, gy, gz, n, avg, mn in rows:
    short = name.replace("papof::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if pats and not any(p in short for p in pats):
        continue
    print("%-34s grid %6d x %5d x %d  n=%4d  avg %8.1f us  min %8.1f us" % (short[:34], gx, gy, gz, n, avg / 1e3, mn / 1e3))
