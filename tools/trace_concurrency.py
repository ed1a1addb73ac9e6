#!/usr/bin/env python3
"""Device side of a many-calls-in-flight run (tools/collection_trace.py under rocprofv3 --kernel-trace): over the busiest
window of the trace -- the last `frac` of it, where the largest number of queues is active -- how much of the time ANY kernel
runs, how many run at once on average, launches per second, and per hardware queue the share of time it is busy and the
gaps between its kernels (a queue whose gaps dominate is waiting for its HOST thread to enqueue, or for dispatch).
usage: trace_concurrency.py run_results.db [frac=0.4]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
T0, T1 = rows[0][1], max(r[2] for r in rows)
w0 = T1 - (T1 - T0) * frac
win = [r for r in rows if r[1] >= w0]
t0, t1 = win[0][1], max(r[2] for r in win)
span = t1 - t0
ev = sorted([(r[1], 1) for r in win] + [(r[2], -1) for r in win])
busy = 0
depth = 0
area = 0
last = t0
for t, d in ev:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    depth += d
    last = t
byq = collections.defaultdict(list)
for r in win:
    byq[r[3]].append(r)
short = lambda n: n.replace("papof::(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
print("window %.1f ms, %d kernels on %d queues: %.0f launches/s" % (span / 1e6, len(win), len(byq), len(win) / (span / 1e9)))
print("some kernel running %.1f %% of the time; %.2f kernels in flight on average" % (100.0 * busy / span, area / span))
qb, qg, qmed = [], [], []
for q, ks in byq.items():
    b = sum(k[2] - k[1] for k in ks)
    gaps = sorted(max(0, ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1))
    qb.append(b / span)
    qg.append(sum(gaps) / span)
    qmed.append(gaps[len(gaps) // 2] / 1e3 if gaps else 0.0)
print("per queue: busy %.1f %% (min %.1f, max %.1f); gaps between its kernels %.1f %% of the time, median gap %.1f us"
      % (100 * sum(qb) / len(qb), 100 * min(qb), 100 * max(qb), 100 * sum(qg) / len(qg), sorted(qmed)[len(qmed) // 2]))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    a = agg[short(r[0])]
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
print("%-32s %7s %10s %10s" % ("kernel", "n", "avg us", "share"))
tot = sum(a[1] for a in agg.values())
for n, a in sorted(agg.items(), key=lambda x: -x[1][1])[:12]:
    print("%-32s %7d %10.1f %9.1f%%" % (n[:32], a[0], a[1] / a[0], 100 * a[1] / tot))
