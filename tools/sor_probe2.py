#!/usr/bin/env python3
"""Increment of the exact-order solve time per additional sweep (fixed image)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from papteam_opticalflow_amd import Papof
g = Papof(0)
for h, w in ((62, 1920), (1080, 1920), (341, 607)):
    prev = None
    for k in (1, 2, 3, 4, 5, 6, 8, 12, 16, 30):
        ms = min(g.bench_sor(h, w, k, mode=0, reps=10) for _ in range(3))
        print("H=%4d W=%4d K=%2d : %8.4f ms%s" % (h, w, k, ms, "" if prev is None else "   +%.1f us/sweep" % ((ms - prev[1]) * 1e3 / (k - prev[0]))), flush=True)
        prev = (k, ms)
g.close()
