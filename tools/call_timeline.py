#!/usr/bin/env python3
"""Time line of ONE device-resident call from a rocprofv3 --kernel-trace database of `bench.py`: per queue busy time and span,
the main queue's kernels by name, the preparation queues' kernels in start order, and when the first solver kernel starts.
usage: call_timeline.py run_results.db [call index, default 4] [--prep]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ci = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
short = lambda n: n.replace("papof::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
starts = [i for i, r in enumerate(rows) if "k_hwc_to_planar" in r[0]]
call_starts = starts[0::2]  # two planarisations open a call
call = rows[call_starts[ci]:call_starts[ci + 1]]
t0, t1 = min(r[1] for r in call), max(r[2] for r in call)
print("call window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(call)))
byq = collections.defaultdict(list)
for r in call:
    byq[r[3]].append(r)
main = max(byq.values(), key=len)
for q, ks in sorted(byq.items()):
    busy = sum(k[2] - k[1] for k in ks)
    print("queue %s%s: %d kernels, busy %.3f ms, span %.3f .. %.3f ms" % (q, " (main)" if ks is main else "", len(ks), busy / 1e6,
                                                                         (ks[0][1] - t0) / 1e6, (ks[-1][2] - t0) / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
for k in main:
    a = agg[short(k[0])]
    a[0] += 1
    a[1] += (k[2] - k[1]) / 1e3
print("main queue:")
for n, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-36s %4d %9.1f us" % (n, cnt, us))
print("  gaps between its kernels %.1f us" % (sum(max(0, main[i + 1][1] - main[i][2]) for i in range(len(main) - 1)) / 1e3))
sor = [k for k in main if "k_sor" in k[0]]
print("first solver kernel: starts at %.1f us, lasts %.1f us; main queue's first kernel at %.1f us" %
      ((sor[0][1] - t0) / 1e3, (sor[0][2] - sor[0][1]) / 1e3, (main[0][1] - t0) / 1e3))
if "--prep" in sys.argv:
    for q, ks in sorted(byq.items()):
        if ks is main:
            continue
        print("queue %s:" % q)
        for k in ks:
            print("  %8.1f %8.1f  %s" % ((k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3, short(k[0])))
    print("main queue up to the first solver kernel:")
    for k in main:
        print("  %8.1f %8.1f  %s" % ((k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3, short(k[0])))
        if "k_sor" in k[0]:
            break
