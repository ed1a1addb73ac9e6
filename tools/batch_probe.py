#!/usr/bin/env python3
"""ms per pair of papof_flow_batch_u8 as a function of the batch size (consecutive pairs of a video, reference schedule unless
given), next to the single call: total (host uint8 in, float64 out) and the solver kernels' share.
usage: batch_probe.py [res] [levels] [B,B,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import cases
from papteam_opticalflow_amd import Papof

res = sys.argv[1] if len(sys.argv) > 1 else "240"
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Bs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16, 32]
a, b = cases.load_frame_u8(res, 1), cases.load_frame_u8(res, 2)
g = Papof(0)
h, w, _ = a.shape
print("%dx%d, %d levels, reference schedule; ms per pair" % (w, h, levels))
print("%6s %10s %10s %10s" % ("batch", "total", "solver", "pairs / s"))
for B in Bs:
    frames = [np.ascontiguousarray(np.roll(a if i % 2 == 0 else b, (i // 2) * 3, axis=1)) for i in range(B + 1)]
    out, _ = g.flow_batch(frames, levels)
    reps = max(2, 24 // B)
    t0 = time.perf_counter()
    sor = 0.0
    for _ in range(reps):
        out, t = g.flow_batch(frames, levels, out=out)
        sor += t[6]
    dt = time.perf_counter() - t0
    print("%6d %10.3f %10.3f %10.1f" % (B, dt / reps / B * 1e3, sor / reps / B * 1e3, reps * B / dt), flush=True)
