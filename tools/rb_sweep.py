#!/usr/bin/env python3
"""Blocked red-black / Jacobi solver: region shape x depth sweep (PAPOF_RB_SHAPE, PAPOF_RB_DEPTH) on the pyramid levels
of a 1080p pair.  ms per solve; '-' where the depth does not fit the shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from papteam_opticalflow_amd import Papof  # noqa: E402
from papteam_opticalflow_amd.capi import PapofError  # noqa: E402

SIZES = [(1080, 1920, 30), (810, 1440, 30), (607, 1080, 30), (455, 810, 30), (341, 607, 30), (270, 480, 30),
         (135, 240, 30)]
SHAPES = {1: "8x6", 2: "8x4", 3: "4x6", 4: "16x3", 5: "12x4"}


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    depths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 6, 8, 10, 12, 16]
    print("mode %d; rows: shape (waves x rows per lane), depth; columns: %s" % (mode, " ".join("%dx%d" % (w, h) for h, w, _ in SIZES)))
    for shape in sorted(SHAPES):
        for depth in depths:
            os.environ["PAPOF_RB_SHAPE"] = str(shape)
            os.environ["PAPOF_RB_DEPTH"] = str(depth)
            g = Papof(0)
            cells = []
            for h, w, k in SIZES:
                try:
                    cells.append("%7.3f" % g.bench_sor(h, w, k, mode=mode, reps=20))
                except PapofError:
                    cells.append("      -")
            g.close()
            print("shape %-4s depth %2d : %s" % (SHAPES[shape], depth, " ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
