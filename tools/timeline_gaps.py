#!/usr/bin/env python3
"""Where is the GPU idle inside one call?  Reads a rocprofv3 --kernel-trace database of `bench.py --steps N` and, for the
LAST timed step, lists the main-stream kernels in start order with the gap in front of each; prints the sums."""
import sqlite3, sys, collections
db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall() if "stream_id" in cols else \
       c.execute("select name, start, end, 0, queue_id from kernels order by start").fetchall()
short = lambda n: n.replace("papof::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
# find the bicubic kernels: each call ends with one; take the window between the last two device-resident calls
bic = [i for i, r in enumerate(rows) if "k_bicubic" in r[0]]
lo, hi = bic[-8] + 1, bic[-7] + 1   # a call well inside the timed region (later calls are the host-buffer legs)
call = rows[lo:hi]
t0, t1 = call[0][1], call[-1][2]
print("call window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(call)))
byq = collections.defaultdict(list)
for r in call:
    byq[r[4]].append(r)
for q, ks in byq.items():
    busy = sum(k[2] - k[1] for k in ks)
    gaps = [max(0, ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1)]
    print("queue %s: %d kernels, busy %.3f ms, gaps %.3f ms (%d gaps > 5 us, max %.1f us)" % (q, len(ks), busy / 1e6, sum(gaps) / 1e6, sum(g > 5000 for g in gaps), max(gaps or [0]) / 1e3))
main = max(byq.values(), key=len)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, k in enumerate(main):
    g = max(0, k[1] - main[i - 1][2]) if i else 0
    a = agg[short(k[0])]
    a[0] += 1; a[1] += (k[2] - k[1]) / 1e3; a[2] += g / 1e3
print("%-34s %5s %10s %12s" % ("kernel (main queue)", "n", "busy us", "gap before us"))
for n, a in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%-34s %5d %10.1f %12.1f" % (n[:34], a[0], a[1], a[2]))
