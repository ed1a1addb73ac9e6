#!/usr/bin/env python3
"""Stress: do the solver kernels' results depend on what else runs on the chip?  For a list of knob settings (one set of
three handles each) and a list of (shape, sweeps) stage solves, three threads solve concurrently and every result is
compared with the solo solve of the same handle type.  Prints one line per configuration; exit code 1 on any mismatch.
(round 2: the two-sweeps-per-wave kernel's identity pair failed exactly this way)"""
import os, sys, threading, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from papteam_opticalflow_amd import Papof

def planes(h, w, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0.5, 50.0, (h, w)), rng.uniform(-0.02, 0.02, (h, w)), rng.uniform(0, 0.05, (h, w)),
            rng.uniform(0, 0.05, (h, w)), rng.uniform(-0.01, 0.01, (h, w)), rng.uniform(-0.01, 0.01, (h, w)))

KNOBS = [{}, {"PAPOF_SOR_FUSE": "2"}, {"PAPOF_SOR_FUSE": "1"}, {"PAPOF_SOR_FUSE": "2", "PAPOF_SOR_DEPTH": "10"},
         {"PAPOF_SOR_FUSE": "1", "PAPOF_SOR_DEPTH": "10"}, {"PAPOF_SOR_FUSE": "1", "PAPOF_SOR_DEPTH": "4"},
         {"PAPOF_SOR_GROUP": "2"}, {"PAPOF_SOR_GROUP": "4"}, {"PAPOF_SOR_XCD": "0"}, {"PAPOF_SOR_XCD": "2"},
         {"PAPOF_SOR_RESIDENT": "48"}]
CASES = [(1080, 1920, 9, 0), (1080, 1920, 12, 0), (607, 1080, 11, 0), (341, 607, 13, 0), (200, 300, 7, 0), (1080, 1920, 9, 1),
         (1080, 1920, 7, 2)]
ALL = ["PAPOF_SOR_FUSE", "PAPOF_SOR_DEPTH", "PAPOF_SOR_GROUP", "PAPOF_SOR_XCD", "PAPOF_SOR_RESIDENT"]
fails = 0
for kn in KNOBS:
    for k in ALL:
        os.environ.pop(k, None)
    os.environ.update(kn)
    hs = [Papof(0) for _ in range(3)]
    for (h, w, n_sor, mode) in CASES:
        if mode and kn:
            continue
        P = planes(h, w, h + w + n_sor)
        om = 1.0 if mode == 2 else 1.8
        want = hs[0].sor(*P, n_sor, mode=mode, omega=om)
        bad = [0, 0, 0]
        def work(i):
            for _ in range(6):
                got = hs[i].sor(*P, n_sor, mode=mode, omega=om)
                bad[i] += not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]))
        th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        [t.start() for t in th]; [t.join() for t in th]
        fails += sum(bad)
        print("%-52s %4dx%-4d sweeps %2d mode %d: wrong %s of 6" % (kn, w, h, n_sor, mode, bad), flush=True)
    [g.close() for g in hs]
sys.exit(1 if fails else 0)
