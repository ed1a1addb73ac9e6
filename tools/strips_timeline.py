#!/usr/bin/env python3
"""Time line of ONE device-resident call out of a rocprofv3 --kernel-trace database (bench.py --steps N): every kernel
with start / end relative to the call, its queue, name and grid -- to see which strip's kernels overlap what
(api.hip: smooth_flow_strips).  usage: strips_timeline.py run_results.db [tail_kernels]"""
import sqlite3, sys
db = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
rows = c.execute("select name, start, end, queue_id, %s from kernels order by start" % gx).fetchall()
short = lambda n: n.replace("papof::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
bic = [i for i, r in enumerate(rows) if "k_bicubic" in r[0]]
lo, hi = bic[-8] + 1, bic[-7] + 1
call = rows[lo:hi]
t0 = call[0][1]
qs = sorted({r[3] for r in call})
print("call %.3f ms, %d kernels, queues %s" % ((call[-1][2] - t0) / 1e6, len(call), qs))
if tail:
    call = call[-tail:]
for r in call:
    print("%9.1f %9.1f  %7.1f us  q%-2d %-40s grid %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3,
                                                       qs.index(r[3]), short(r[0])[:40], r[4]))
