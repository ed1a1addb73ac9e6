#!/usr/bin/env python3
"""What plain streaming kernels reach on this chip (PyTorch's own fill / copy / reduction kernels, HIP events): the practical
ceilings next to which the 8 TB/s peak of the roofline and this repository's kernels should be read."""
import torch

dev = torch.device("cuda", 0)
n = 1 << 27  # 128 Mi doubles = 1 GiB
a = torch.empty(n, dtype=torch.float64, device=dev)
b = torch.empty(n, dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 8 / 1e9
for name, fn, traffic in (("fill (write only)", lambda: a.zero_(), gb), ("copy (read + write)", lambda: b.copy_(a), 2 * gb),
                          ("sum (read only)", lambda: a.sum(), gb), ("a += b (2 reads + 1 write)", lambda: a.add_(b), 3 * gb)):
    t = timed(fn)
    print("%-28s %7.3f ms  %6.2f TB/s" % (name, t * 1e3, traffic / t / 1e3))
for mb in (32, 128):  # the sizes of this path's planes: does a smaller buffer (Infinity Cache resident) change the picture?
    m = mb * (1 << 20) // 8
    t = timed(lambda: a[:m].zero_(), 50)
    print("fill of %3d MB               %7.3f us  %6.2f TB/s" % (mb, t * 1e6, m * 8 / 1e12 / t))
    t = timed(lambda: b[:m].copy_(a[:m]), 50)
    print("copy of %3d MB               %7.3f us  %6.2f TB/s" % (mb, t * 1e6, 2 * m * 8 / 1e12 / t))
