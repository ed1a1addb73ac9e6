#!/usr/bin/env python3
"""Phase split of one 1920x1080 call (phase_timing=1: the stream is synchronised at every phase boundary, so the sum is
larger than the un-instrumented call; the reference's timing keys, include/papof.h)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import cases
from papteam_opticalflow_amd import Papof, default_params
g = Papof(0)
a, b = cases.load_pair("1920")
keys = ["allocation", "construction", "phase1 generate", "phase2 derivatives", "phase3 psi", "phase4 system", "phase5 SOR",
        "phase6 update", "postprocessing", "total"]
for timing in (1, 0):
    P = default_params(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0, phase_timing=timing)
    for rep in range(3):
        out = g.coarse2fine_flow(a, b, 5, P)
    t = out[3]
    print("phase_timing=%d: " % timing + "  ".join("%s %.3f" % (k, t[i] * 1e3) for i, k in enumerate(keys)), flush=True)
g.close()
