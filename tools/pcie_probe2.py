#!/usr/bin/env python3
"""Where does the host side of the drop-in call go?  fresh vs reused output arrays, float64 vs uint8 inputs."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import cases
from papteam_opticalflow_amd import Papof, default_params, capi
a8, b8 = cases.load_frame_u8("1920", 1), cases.load_frame_u8("1920", 2)
a, b = a8.astype(np.float64) / 255.0, b8.astype(np.float64) / 255.0
h, w, c = a.shape
P = default_params(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)
g = Papof(0)
_D = ctypes.POINTER(ctypes.c_double)
def p(x): return x.ctypes.data_as(_D)
def pb(x): return x.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
vx, vy, wi, t = np.zeros((h, w)), np.zeros((h, w)), np.zeros((h, w, c)), np.zeros(10)
def timeit(name, fn, n=4):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print("%-44s %.2f ms" % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)
timeit("alloc np.zeros x3 + touch", lambda: [x.fill(0) for x in (np.zeros((h, w)), np.zeros((h, w)), np.zeros((h, w, c)))])
timeit("f64 in, wrapper (pinned recycled outputs)", lambda: g.coarse2fine_flow(a, b, 5, P))
os.environ["PAPOF_PINNED_OUT"] = "0"
timeit("f64 in, wrapper (fresh np.zeros outputs)", lambda: g.coarse2fine_flow(a, b, 5, P))
os.environ["PAPOF_PINNED_OUT"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "papteam_opticalflow_amd", "dropin"))
import pyflow
timeit("pyflow.coarse2fine_flow (the drop-in)", lambda: pyflow.coarse2fine_flow(a, b, 5, n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0))
timeit("f64 in, reused outputs (C ABI)", lambda: g.L.papof_flow(g.h, p(a), p(b), h, w, c, 5, ctypes.byref(P), p(vx), p(vy), p(wi), p(t)))
timeit("u8 in, reused outputs (C ABI)", lambda: g.L.papof_flow_u8(g.h, pb(a8), pb(b8), h, w, c, 5, ctypes.byref(P), p(vx), p(vy), p(wi), p(t)))
d1, d2 = g.dev_alloc(a.nbytes), g.dev_alloc(a.nbytes)
dx, dy, dw = g.dev_alloc(h * w * 8), g.dev_alloc(h * w * 8), g.dev_alloc(a.nbytes)
g.dev_upload(d1, a); g.dev_upload(d2, b)
timeit("device resident", lambda: g.flow_device(d1, d2, h, w, c, 5, P, dx, dy, dw))
timeit("download only (3 arrays, pageable)", lambda: (g.dev_download(vx, dx), g.dev_download(vy, dy), g.dev_download(wi, dw)))
timeit("upload only (2 f64 frames, pageable)", lambda: (g.dev_upload(d1, a), g.dev_upload(d2, b)))
g.close()
