#!/usr/bin/env python3
"""Measurements of the SOR solver alone (papof_bench_sor: synthetic coefficient planes resident in HBM, back-to-back solves timed
with HIP events on the library's stream).  One script for what used to be sor_bench.py, sor_probe.py,
sor_probe2.py, sor_tau.py and sor_dbg.py:

  sor_tool.py bench [modes]      every pyramid-level size of the bench workloads: ms per solve, algorithmic GB/s (80 B per update)
  sor_tool.py decompose          exact order: one task (tau per step), sweep chain, band chain
  sor_tool.py per-sweep          exact order: increment of the solve time per additional sweep
  sor_tool.py tau                exact order: per-step cost of ONE task (one band, very wide plane; PAPOF_SOR_FUSE / _DEPTH from env)
  sor_tool.py timeline [H W K]   PAPOF_SOR_DBG=1: per-task time line of the first tasks of a solve / wait statistics of k_sor_group
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LEVELS = [(1080, 1920, 30), (810, 1440, 33), (607, 1080, 36), (455, 810, 39), (341, 607, 42), (540, 960, 30),
          (270, 480, 30), (135, 240, 30)]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "bench"
    if what == "timeline":
        os.environ["PAPOF_SOR_DBG"] = "1"  # read when the handle is created
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    if what == "bench":
        modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
        for h, w, k in LEVELS:
            for mode in modes:
                ms = g.bench_sor(h, w, k, mode=mode, reps=5)
                gbs = h * w * k * 80 / 1e9 / (ms * 1e-3)
                print("%4dx%-4d sweeps %2d mode %d : %8.3f ms/solve  %8.1f GB/s algorithmic  (%.1f%% of 8 TB/s)"
                      % (w, h, k, mode, ms, gbs, gbs / 80.0), flush=True)
    elif what == "decompose":
        for h, w, k in [(62, 1920, 1), (62, 1920, 2), (62, 1920, 4), (62, 1920, 8), (62, 1920, 30), (124, 1920, 1), (496, 1920, 1),
                        (1080, 1920, 1), (1080, 1920, 2), (1080, 1920, 30), (62, 240, 1), (62, 240, 30), (62, 7680, 1)]:
            ms = g.bench_sor(h, w, k, mode=0, reps=10)
            print("H=%4d W=%4d K=%2d nb=%2d : %8.4f ms   per-step(W+63) %.3f us" % (h, w, k, (h + 61) // 62, ms, ms * 1e3 / (w + 63)),
                  flush=True)
    elif what == "per-sweep":
        for h, w in ((62, 1920), (1080, 1920), (341, 607)):
            prev = None
            for k in (1, 2, 3, 4, 5, 6, 8, 12, 16, 30):
                ms = min(g.bench_sor(h, w, k, mode=0, reps=10) for _ in range(3))
                print("H=%4d W=%4d K=%2d : %8.4f ms%s" % (h, w, k, ms, "" if prev is None else
                                                         "   +%.1f us/sweep" % ((ms - prev[1]) * 1e3 / (k - prev[0]))), flush=True)
                prev = (k, ms)
    elif what == "tau":
        for h, w, k in [(58, 7680, 1), (58, 7680, 2), (58, 7680, 4)]:
            ms = g.bench_sor(h, w, k, mode=0, reps=10)
            print("fuse=%s depth=%s H=%d W=%d K=%d: %.4f ms  %.4f us per step" % (
                os.environ.get("PAPOF_SOR_FUSE", "1"), os.environ.get("PAPOF_SOR_DEPTH", "auto"), h, w, k, ms, ms * 1e3 / (w + 63)),
                flush=True)
    elif what == "timeline":
        cases = [(341, 607, 42), (135, 240, 30)] if len(sys.argv) < 5 else [tuple(int(x) for x in sys.argv[2:5])]
        for h, w, k in cases:
            print("H=%d W=%d K=%d: %.4f ms" % (h, w, k, g.bench_sor(h, w, k, mode=0, reps=3)), flush=True)
    else:
        sys.exit(__doc__)
    g.close()


if __name__ == "__main__":
    main()
