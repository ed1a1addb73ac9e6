#!/usr/bin/env python3
"""Per-step cost of one task of the exact-order SOR kernels: one band, very wide image (PAPOF_SOR_FUSE / _DEPTH from env)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from papteam_opticalflow_amd import Papof
g = Papof(0)
for h, w, k in [(58, 7680, 1), (58, 7680, 2), (58, 7680, 4)]:
    ms = g.bench_sor(h, w, k, mode=0, reps=10)
    print("fuse=%s depth=%s H=%d W=%d K=%d: %.4f ms  %.4f us per step" % (
        os.environ.get("PAPOF_SOR_FUSE", "1"), os.environ.get("PAPOF_SOR_DEPTH", "auto"), h, w, k, ms, ms * 1e3 / (w + 63)), flush=True)
g.close()
