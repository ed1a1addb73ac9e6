import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from papteam_opticalflow_amd import Papof
g = Papof(0)
for (H, W, n_sor, split, delay) in ((1080, 1920, 7, 9, 0), (1080, 1920, 9, 9, 100), (1080, 1920, 11, 5, 300), (1080, 1920, 3, 9, 0), (1080, 1920, 33, 9, 100), (810, 1440, 9, 7, 0), (1080, 1920, 8, 9, 0)):
    mm, nb = g.test_sor_strips(H, W, n_sor, split, 8, delay)
    print("%dx%d n_sor %d bands %d split %d delay %d us: mismatching cells %d (8 reps)" % (W, H, n_sor, nb, split, delay, mm), flush=True)
