#!/usr/bin/env python3
"""Does libpapof.so coexist with PyTorch's HIP runtime in one process, in either initialisation order?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
import numpy as np
import cases
a, b = cases.load_pair("240")


def use_torch():
    import torch
    torch.cuda.set_device(0)
    x = torch.ones(1024, device="cuda")
    torch.cuda.synchronize()
    print("torch ok:", float(x.sum()), torch.version.hip, flush=True)


def use_papof():
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    vx, vy, w, t = g.coarse2fine_flow(a, b, 3)
    print("papof ok:", float(np.abs(vx).max()), flush=True)
    g.close()


steps = [use_torch, use_papof] if order == "torch_first" else [use_papof, use_torch]
for f in steps:
    f()
os.system("grep -E 'amdhip|rccl|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
