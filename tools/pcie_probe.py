#!/usr/bin/env python3
"""PCIe-inclusive time of the drop-in entry points (host numpy buffers in and out) for a few host-thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import cases
from papteam_opticalflow_amd import Papof, default_params
a8, b8 = cases.load_frame_u8("1920", 1), cases.load_frame_u8("1920", 2)
a, b = a8.astype(np.float64) / 255.0, b8.astype(np.float64) / 255.0
P = default_params(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)
for th in (int(x) for x in (sys.argv[1:] or ["4", "8", "16"])):
    os.environ["PAPOF_HOST_THREADS"] = str(th)
    g = Papof(0)
    for name, fn in (("f64", lambda: g.coarse2fine_flow(a, b, 5, P)), ("u8", lambda: g.coarse2fine_flow_u8(a8, b8, 5, P))):
        fn()
        t0 = time.perf_counter()
        for _ in range(4):
            fn()
        print("host threads %2d  %-3s : %.2f ms per pair" % (th, name, (time.perf_counter() - t0) / 4 * 1e3), flush=True)
    g.close()
