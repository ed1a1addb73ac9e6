#!/usr/bin/env python3
"""Export the parts of a rocprofv3 run (rocpd sqlite database, the default output of ROCm 7.2's rocprofv3) that get
committed under profiles/: per-kernel statistics of a --kernel-trace run, or per-kernel averages of one --pmc counter.

usage: rocpd_export.py stats <run_results.db> <out.csv>
       rocpd_export.py pmc   <run_results.db> <COUNTER> <out.csv>
"""
import csv
import sqlite3
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])


def pmc(db, counter, out):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value), min(value), max(value), avg(duration), grid_size, "
                     "workgroup_size, lds_block_size, vgpr_count from counters_collection where counter_name = ? "
                     "group by kernel_name, grid_size order by sum(value) desc", (counter,)).fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", counter + "_avg", counter + "_min", counter + "_max", "AvgDurationNs",
                    "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count"])
        for r in rows:
            w.writerow(list(r))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
