#!/usr/bin/env python3
"""Per-task wait statistics of the grouped exact-order SOR kernel (PAPOF_SOR_DBG=1): where a wave spends its time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAPOF_SOR_DBG"] = "1"
from papteam_opticalflow_amd import Papof
g = Papof(0)
for h, w, k in [(62, 1920, 2), (62, 1920, 4), (62, 1920, 8)]:
    ms = g.bench_sor(h, w, k, mode=0, reps=3)
    print("H=%d W=%d K=%d: %.4f ms" % (h, w, k, ms), flush=True)
g.close()
