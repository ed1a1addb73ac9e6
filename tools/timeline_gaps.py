#!/usr/bin/env python3
"""Where is the GPU idle inside one call?  Reads a rocprofv3 --kernel-trace database of `bench.py` and, for ONE device-resident
call of the timed region (a call opens with two k_hwc_to_planar launches; default: the fifth), lists the main queue's kernels
by name with their busy time and the gaps in front of them, and prints the sums -- among them everything on the main queue
that is not a solver kernel (kernels + gaps), the figure VERDICT 4 of round 2 was written against.
usage: timeline_gaps.py run_results.db [call index]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ci = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
short = lambda n: n.replace("papof::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
starts = [i for i, r in enumerate(rows) if "k_hwc_to_planar" in r[0]]
call_starts = starts[0::2]
call = rows[call_starts[ci]:call_starts[ci + 1]]
t0, t1 = min(r[1] for r in call), max(r[2] for r in call)
print("call window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(call)))
byq = collections.defaultdict(list)
for r in call:
    byq[r[3]].append(r)
for q, ks in sorted(byq.items()):
    busy = sum(k[2] - k[1] for k in ks)
    gaps = [max(0, ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1)]
    print("queue %s: %d kernels, busy %.3f ms, gaps %.3f ms (%d gaps > 5 us, max %.1f us), first kernel at %.3f ms"
          % (q, len(ks), busy / 1e6, sum(gaps) / 1e6, sum(g > 5000 for g in gaps), max(gaps or [0]) / 1e3, (ks[0][1] - t0) / 1e6))
main = max(byq.values(), key=len)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, k in enumerate(main):
    g = max(0, k[1] - main[i - 1][2]) if i else 0
    a = agg[short(k[0])]
    a[0] += 1
    a[1] += (k[2] - k[1]) / 1e3
    a[2] += g / 1e3
print("%-34s %5s %10s %14s" % ("kernel (main queue)", "n", "busy us", "gap before us"))
for n, a in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%-34s %5d %10.1f %14.1f" % (n[:34], a[0], a[1], a[2]))
sor = sum(a[1] for n, a in agg.items() if n.startswith("k_sor"))
busy = sum(a[1] for a in agg.values())
gaps = sum(a[2] for a in agg.values())
print("main queue: solver kernels %.3f ms, everything else %.3f ms busy + %.3f ms of gaps = %.3f ms (behind the %.3f ms until its "
      "first kernel)" % (sor / 1e3, (busy - sor) / 1e3, gaps / 1e3, (busy - sor + gaps) / 1e3, (main[0][1] - t0) / 1e6))
