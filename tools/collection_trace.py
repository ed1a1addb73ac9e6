#!/usr/bin/env python3
"""What limits a collection of small frames?  k sequences in flight (own handle, own host thread, one
stream each -- what flow_collection() does), 240x135 pairs on the reference schedule, host uint8 in / float64 out.  Per call the
library reports the host wall time it spent ENQUEUEING (the HIP runtime's launch path) and WAITING for the stream
(papof_last_host_times); the loop time per pair on top of their sum is Python / result handling.
usage: collection_trace.py [res] [pairs per sequence] [k,k,...]
Under rocprofv3 --kernel-trace the same run gives the device side (tools/trace_concurrency.py on the kernel trace)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import cases
from papteam_opticalflow_amd import FlowSequence, Papof, capi

res = sys.argv[1] if len(sys.argv) > 1 else "240"
per_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16]
a, b = cases.load_frame_u8(res, 1), cases.load_frame_u8(res, 2)
h, w, c = a.shape
pool = [Papof(0) for _ in range(max(ks))]
for g in pool:
    g.set_stream_overlap(False)
print("%dx%d, 5 levels, reference schedule, %d pairs per sequence; ms per call" % (w, h, per_seq))
print("%9s %10s %10s %10s %12s %12s" % ("in flight", "ms / pair", "enqueue", "wait", "python rest", "pairs / s"))
for k in ks:
    stats = [[] for _ in range(k)]

    def run(s, warm):
        seq = FlowSequence(5, handle=pool[s])
        out = (capi.result_array((h, w)), capi.result_array((h, w)), capi.result_array((h, w, c)))
        seq.push(a, out)
        for i in range(2 if warm else per_seq):
            t0 = time.perf_counter()
            seq.push(b if i % 2 == 0 else a, out)
            dt = time.perf_counter() - t0
            if not warm:
                e, wt = pool[s].last_host_times()
                stats[s].append((dt, e, wt))
        seq.reset()
    for warm in (True, False):
        th = [threading.Thread(target=run, args=(s, warm)) for s in range(k)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
    flat = np.array([x for s in stats for x in s])
    call, enq, wt = flat.mean(axis=0) * 1e3
    n = k * per_seq
    print("%9d %10.3f %10.3f %10.3f %12.3f %12.1f" % (k, wall / n * 1e3, enq, wt, call - enq - wt, n / wall), flush=True)
