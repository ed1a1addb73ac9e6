#!/usr/bin/env python3
"""The exact-order band split's STAGED protocol (what runs over RCCL) on ONE device, all ranks as threads through the RCCL
transport bound to the stand-in of tests/fake_rccl: ms per 1920x1080 config-4 pair and the solver kernels' share of rank 0 for
1 / 2 / 3 / 5 / 10 ranges of sweeps per solve (PAPOF_BANDS_CHUNKS).  A rehearsal, not a multi-GPU measurement: the ranks share one
chip (the latency-bound solver kernels leave most of it idle, the other stages do not), but it shows whether ranges of sweeps let
the ranks overlap at all.  usage: bands_chunks_probe.py [nranks=8] [res=1920]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ["PAPOF_RCCL_LIB"] = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
import numpy as np
import cases
from papteam_opticalflow_amd import Papof, capi, default_params

nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res = sys.argv[2] if len(sys.argv) > 2 else "1920"
a, b = cases.load_pair(res)
P = default_params(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0, sor_mode=0)
one = Papof(0)
one.coarse2fine_flow(a, b, 5, P)
t0 = time.perf_counter()
for _ in range(3):
    want = one.coarse2fine_flow(a, b, 5, P)
print("%s pair, config-4 schedule: one GPU %.2f ms per call (host buffers)" % (res, (time.perf_counter() - t0) / 3 * 1e3))
one.close()
for chunks in (1, 2, 3, 5, 10):
    os.environ["PAPOF_BANDS_CHUNKS"] = str(chunks)
    grp = capi.RcclTileGroup(nranks, nranks, 1, 0)
    grp.coarse2fine_flow(a, b, 5, P)
    t0 = time.perf_counter()
    vx, vy, wi, t = grp.coarse2fine_flow(a, b, 5, P)
    dt = time.perf_counter() - t0
    n_ex, _ = grp.ranks[0].stats()
    ok = np.array_equal(vx, want[0]) and np.array_equal(vy, want[1])
    print("%d ranks on one device, %2d ranges of sweeps per solve: device call of rank 0 %.2f ms (its solver kernels %.2f ms), %d "
          "exchanges, bits %s" % (nranks, chunks, t[9] * 1e3, t[6] * 1e3, n_ex, "= one GPU" if ok else "DIFFER"), flush=True)
    grp.close()
