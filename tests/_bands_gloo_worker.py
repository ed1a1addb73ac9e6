"""Worker of tests/test_bands_gloo.py: one rank of the EXACT-ORDER band split (csrc/tiles.hip: bands_flow), on the CPU.

Rehearses its N>1 protocol -- the STAGED form the RCCL transport runs -- with real inter-process messages
(torch.distributed, gloo).  The split (which bands a rank runs, which rows it finally owns) comes from the PRODUCT
library's host function papof_bands_plan; the solve follows the kernel's dataflow: at sweep k rank g owns the rows
62 B0 - k .. 62 B1 - k - 1 (the bands climb one row per sweep), needs from the rank above the row just above its range with its
sweep-k value -- one row per sweep, all K of them in one message once the rank above has finished -- and hands the same to
the rank below.  The row that enters a range from above at sweep k carries its sweep-(k-1) value: it is the row received for
sweep k - 1.  Rank 0 gathers the finally owned rows and compares them bit for bit with the CPU oracle's lexicographic solve
(src/OpticalFlow.cpp:458-505) of the whole plane."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from papteam_opticalflow_amd import capi  # noqa: E402  (host-side geometry only)

BR = 62  # PAPOF_BAND_ROWS


def sweep_rows(P, du, dv, r0, r1, alpha, omega, h, w):
    """in-place lexicographic update of rows r0 .. r1-1 (src/OpticalFlow.cpp:458-505; a1 / a2 hoisted as in the kernel)"""
    phi, xy, a1, a2, b1, b2 = P
    om1, nalpha = 1 - omega, -alpha
    for i in range(r0, r1):
        for j in range(w):
            s1 = s2 = 0.0
            pc = phi[i, j]
            if j > 0:
                s1 += phi[i, j - 1] * du[i, j - 1]
                s2 += phi[i, j - 1] * dv[i, j - 1]
            if j < w - 1:
                s1 += pc * du[i, j + 1]
                s2 += pc * dv[i, j + 1]
            if i > 0:
                s1 += phi[i - 1, j] * du[i - 1, j]
                s2 += phi[i - 1, j] * dv[i - 1, j]
            if i < h - 1:
                s1 += pc * du[i + 1, j]
                s2 += pc * dv[i + 1, j]
            s1 *= nalpha
            s2 *= nalpha
            s1 += xy[i, j] * dv[i, j]
            nu = om1 * du[i, j] + a1[i, j] * (b1[i, j] - s1)
            s2 += xy[i, j] * nu
            nv = om1 * dv[i, j] + a2[i, j] * (b2[i, j] - s2)
            du[i, j] = nu
            dv[i, j] = nv


def main():
    h, w, K, seed = (int(x) for x in sys.argv[1:5])
    dist.init_process_group("gloo")
    rank, n = dist.get_rank(), dist.get_world_size()
    alpha, omega = 0.012, 1.8
    rng = np.random.default_rng(seed)
    phi = rng.uniform(0.5, 50.0, (h, w))
    xy = rng.uniform(-0.02, 0.02, (h, w))
    x2 = rng.uniform(0.0, 0.05, (h, w))
    y2 = rng.uniform(0.0, 0.05, (h, w))
    b1 = rng.uniform(-0.01, 0.01, (h, w))
    b2 = rng.uniform(-0.01, 0.01, (h, w))
    # the hoisted diagonals, as the assembly kernel computes them (kernels.hip: sor_diagonals)
    coeff = np.zeros((h, w))
    coeff[:, 1:] += phi[:, :-1]
    coeff[:, :-1] += phi[:, :-1]
    coeff[1:, :] += phi[:-1, :]
    coeff[:-1, :] += phi[:-1, :]
    coeff *= alpha
    a1 = omega / (x2 + alpha * 0.05 + coeff)
    a2 = omega / (y2 + alpha * 0.05 + coeff)
    P = (phi, xy, a1, a2, b1, b2)

    plan = capi.bands_plan(h, w, K, n, rank)
    B0, B1 = plan["B0"], plan["B1"]
    mine = B1 > B0
    nb = capi.bands_plan(h, w, K, 1, 0)["B1"]
    m = min(n, nb)  # ranks with bands: 0 .. m - 1
    du, dv = np.zeros((h, w)), np.zeros((h, w))
    msgs = 0
    # Ranges of sweeps (PAPOF_BANDS_CHUNKS of bands_flow; sys.argv[5], default 1 = the ranks take turns): per range, receive
    # the range's rows from the rank above, run the own rows for those sweeps, send the own last row of each sweep to the rank
    # below -- rank g + 1 works on range c while rank g works on range c + 1 (blocking sends and receives between processes).
    chunks = max(1, min(int(sys.argv[5]) if len(sys.argv) > 5 else 1, K))
    for c in range(chunks):
        k0, k1 = K * c // chunks, K * (c + 1) // chunks
        if k1 <= k0 or not mine:
            continue
        inbox = None
        if B0 > 0:
            t = torch.zeros(k1 - k0, 2, w, dtype=torch.float64)
            dist.recv(t, src=rank - 1)
            inbox = t.numpy()
            msgs += 1
        outbox = np.zeros((k1 - k0, 2, w))
        for k in range(k0, k1):
            up = BR * B0 - k - 1  # the row above the range: its sweep-k value comes from the rank above
            if inbox is not None and 0 <= up < h:
                du[up], dv[up] = inbox[k - k0, 0], inbox[k - k0, 1]
            r0, r1 = max(0, BR * B0 - k), min(h, BR * B1 - k)
            sweep_rows(P, du, dv, r0, r1, alpha, omega, h, w)
            last = BR * B1 - k - 1  # what the rank below needs of this sweep
            if 0 <= last < h:
                outbox[k - k0, 0], outbox[k - k0, 1] = du[last], dv[last]
        if rank + 1 < m:
            dist.send(torch.from_numpy(outbox), dst=rank + 1)
            msgs += 1
    # gather the finally owned rows on rank 0
    y0, y1 = plan["final_rows"]
    mine_rows = torch.zeros(2, h, w, dtype=torch.float64)
    mine_rows[0, y0:y1] = torch.from_numpy(du[y0:y1])
    mine_rows[1, y0:y1] = torch.from_numpy(dv[y0:y1])
    dist.all_reduce(mine_rows)  # the ranges are a partition: a sum of disjoint supports
    cover = torch.zeros(h, dtype=torch.float64)
    cover[y0:y1] += 1
    dist.all_reduce(cover)
    tot = torch.tensor([float(msgs)], dtype=torch.float64)
    dist.all_reduce(tot)
    if rank == 0:
        from _libs import OracleLib
        eu, ev = OracleLib().sor(phi, xy, x2, y2, b1, b2, K, alpha=alpha, omega=omega, mode=0)
        ok = bool((cover.numpy() == 1).all() and np.array_equal(mine_rows[0].numpy(), eu) and
                  np.array_equal(mine_rows[1].numpy(), ev))
        print("BANDS_GLOO ok=%d messages=%d ranks_with_bands=%d" % (int(ok), int(tot.item()), m), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
