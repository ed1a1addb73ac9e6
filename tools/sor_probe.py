#!/usr/bin/env python3
"""Decompose the exact-order SOR solve time: single task (tau per step), sweep chain, band chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from papteam_opticalflow_amd import Papof
g = Papof(0)
cases = [(62, 1920, 1), (62, 1920, 2), (62, 1920, 4), (62, 1920, 8), (62, 1920, 30), (124, 1920, 1), (496, 1920, 1), (1080, 1920, 1),
         (1080, 1920, 2), (1080, 1920, 30), (62, 240, 1), (62, 240, 30), (62, 7680, 1)]
for h, w, k in cases:
    ms = g.bench_sor(h, w, k, mode=0, reps=10)
    nb = (h + 61) // 62
    print("H=%4d W=%4d K=%2d nb=%2d : %8.4f ms   per-step(W+63) %.3f us" % (h, w, k, nb, ms, ms * 1e3 / (w + 63)), flush=True)
g.close()
