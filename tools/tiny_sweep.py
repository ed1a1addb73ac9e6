#!/usr/bin/env python3
"""A/B of k_sor_tiny's tile width C on the level sizes of deep pyramids (PAPOF_TINY_C is read once per process, so one
process per C): ms per solve through papof_bench_sor."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [(60, 108, 60), (45, 81, 63), (33, 60, 66), (25, 45, 69), (19, 34, 72), (56, 101, 39), (42, 75, 42), (31, 56, 45),
         (23, 42, 48), (13, 24, 54), (4, 7, 66)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    out = []
    for h, w, k in SIZES:
        try:
            out.append("%.1f" % (g.bench_sor(h, w, k, 0, reps=20) * 1e3))
        except Exception as e:
            out.append("err")
    print(" ".join(out))
    sys.exit(0)
print("C      " + " ".join("%dx%dx%d" % (w, h, k) for h, w, k in SIZES))
for c in [0, 1, 2, 3, 4, 5, 6, 8, 10, 14]:
    env = dict(os.environ, PAPOF_TINY_C=str(c))
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
    print("%-6s %s" % ("auto" if c == 0 else c, r.stdout.strip() or r.stderr[-300:]))
