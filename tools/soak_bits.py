#!/usr/bin/env python3
"""Soak: the same calls many times -- alone, and with a second handle hammering the chip from another thread -- must return the
same bits every time (a rare hand-off or hazard bug shows up as one differing call in hundreds)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import cases
from papteam_opticalflow_amd import Papof, default_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g, g2 = Papof(0), Papof(0)
stop = False


def hammer():
    a, b = cases.load_pair("960")
    while not stop:
        g2.coarse2fine_flow(a, b, 4)


bad = 0
for res, levels, kw in (("1920", 5, dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)), ("480", 5, {}), ("240", 15, {})):
    a, b = cases.load_pair(res)
    P = default_params(**kw)
    ref = g.coarse2fine_flow(a, b, levels, P)[:3]
    for phase in ("alone", "beside another handle"):
        th = None
        if phase != "alone":
            stop = False
            th = threading.Thread(target=hammer)
            th.start()
        t0 = time.perf_counter()
        for it in range(n):
            r = g.coarse2fine_flow(a, b, levels, P)[:3]
            if not all(np.array_equal(x, y) for x, y in zip(r, ref)):
                bad += 1
                print("MISMATCH", res, levels, phase, it, flush=True)
        if th:
            stop = True
            th.join()
        print("%s L%d %-22s %d calls, %.1f ms each, mismatches so far %d" % (res, levels, phase, n, (time.perf_counter() - t0) / n * 1e3, bad), flush=True)
# batches (csrc/batch.hip): 16 consecutive 240x135 pairs per launch chain, alone and beside the other handle
v = [np.ascontiguousarray(np.roll(cases.load_frame_u8("240", 1 + i % 2), (i // 2) * 3, axis=1)) for i in range(17)]
ref = [tuple(x.copy() for x in o) for o in g.flow_batch(v, 5)[0]]
for phase in ("alone", "beside another handle"):
    th = None
    if phase != "alone":
        stop = False
        th = threading.Thread(target=hammer)
        th.start()
    t0 = time.perf_counter()
    nb = max(1, n // 4)
    for it in range(nb):
        out = g.flow_batch(v, 5)[0]
        if not all(np.array_equal(x, y) for o, r in zip(out, ref) for x, y in zip(o, r)):
            bad += 1
            print("MISMATCH batch", phase, it, flush=True)
    if th:
        stop = True
        th.join()
    print("batch of 16 x 240x135 L5 %-22s %d batches, %.2f ms per pair, mismatches so far %d" % (phase, nb, (time.perf_counter() - t0) / nb / 16 * 1e3, bad), flush=True)
print("guard stats", g.lap_guard_stats())
sys.exit(1 if bad else 0)
