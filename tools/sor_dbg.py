#!/usr/bin/env python3
"""Per-task statistics of the exact-order SOR kernels (PAPOF_SOR_DBG=1): the time line of the first tasks of a solve
(k_sor_exact) or the wait statistics of the grouped kernel (PAPOF_SOR_GROUP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAPOF_SOR_DBG"] = "1"
from papteam_opticalflow_amd import Papof
g = Papof(0)
cases = [(341, 607, 42), (135, 240, 30)] if len(sys.argv) < 4 else [tuple(int(x) for x in sys.argv[1:4])]
for h, w, k in cases:
    ms = g.bench_sor(h, w, k, mode=0, reps=3)
    print("H=%d W=%d K=%d: %.4f ms" % (h, w, k, ms), flush=True)
g.close()
