#!/usr/bin/env python3
"""SOR micro-benchmark: synthetic coefficient planes resident in HBM, `reps` back-to-back solves
timed with HIP events on the library's stream (papof_bench_sor).  Prints ms per solve and the algorithmic GB/s
(80 B per cell-update)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from papteam_opticalflow_amd import Papof  # noqa: E402

SIZES = [(1080, 1920, 30), (810, 1440, 33), (607, 1080, 36), (455, 810, 39), (341, 607, 42), (540, 960, 30),
         (270, 480, 30), (135, 240, 30)]


def main():
    modes = [int(m) for m in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
    g = Papof(0)
    for h, w, k in SIZES:
        for mode in modes:
            ms = g.bench_sor(h, w, k, mode=mode, reps=5)
            gbs = h * w * k * 80 / 1e9 / (ms * 1e-3)
            print("%4dx%-4d sweeps %2d mode %d : %8.3f ms/solve  %8.1f GB/s algorithmic  (%.1f%% of 8 TB/s)"
                  % (w, h, k, mode, ms, gbs, gbs / 80.0), flush=True)
    g.close()


if __name__ == "__main__":
    main()
