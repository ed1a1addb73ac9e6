"""Worker of tests/test_tiles_gloo.py: one rank of a tile group, on the CPU.

Rehearses the N>1 protocol of the tiled solver (csrc/tiles.hip) with real inter-process messages (torch.distributed,
gloo): the tile rectangles and the owner -> needer message plan come from the PRODUCT library's host functions
(papof_tiles_rect / papof_tiles_halo_message -- no GPU needed); the solve follows tiles.hip launch for launch -- a
period of S half-sweeps between two (du, dv) exchanges is one or more launches of the temporally blocked solver (the
numpy model of k_sor_blocked, tests/sim_sor_blocked.py), each of depth g delivering the tile grown by what is left of
the period and reading that rectangle grown by g, alternating between two pairs of planes; an exchange fills the
ghost ring of the pair the next launch reads.  Rank 0 gathers the tiles and compares the result bit for bit with the
CPU oracle's red-black solve of the whole plane.  (`halfsweep` below is the round-1 schedule, one half-sweep on the
tile grown by S-1-m at a time: kept as an independent restatement and run as a cross-check.)
"""
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


def grow(r, d, w, h):
    x0, y0, x1, y1 = r
    if x1 <= x0 or y1 <= y0:
        return r
    return max(0, x0 - d), max(0, y0 - d), min(w, x1 + d), min(h, y1 + d)


def halfsweep(P, du, dv, region, colour, alpha, omega, h, w):
    """cells (i, j) of `region` with (i + j) % 2 == colour, updated in place from the other colour's values"""
    phi, xy, a1, a2, b1, b2 = P
    x0, y0, x1, y1 = region
    ii, jj = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
    sel = ((ii + jj) & 1) == colour
    i, j = ii[sel], jj[sel]
    z = np.zeros(i.shape)
    s1, s2 = z.copy(), z.copy()
    m = j > 0
    wgt = np.where(m, phi[i, np.maximum(j - 1, 0)], 0.0)
    s1 = np.where(m, s1 + wgt * du[i, np.maximum(j - 1, 0)], s1)
    s2 = np.where(m, s2 + wgt * dv[i, np.maximum(j - 1, 0)], s2)
    m = j < w - 1
    s1 = np.where(m, s1 + phi[i, j] * du[i, np.minimum(j + 1, w - 1)], s1)
    s2 = np.where(m, s2 + phi[i, j] * dv[i, np.minimum(j + 1, w - 1)], s2)
    m = i > 0
    wgt = np.where(m, phi[np.maximum(i - 1, 0), j], 0.0)
    s1 = np.where(m, s1 + wgt * du[np.maximum(i - 1, 0), j], s1)
    s2 = np.where(m, s2 + wgt * dv[np.maximum(i - 1, 0), j], s2)
    m = i < h - 1
    s1 = np.where(m, s1 + phi[i, j] * du[np.minimum(i + 1, h - 1), j], s1)
    s2 = np.where(m, s2 + phi[i, j] * dv[np.minimum(i + 1, h - 1), j], s2)
    s1 = s1 * -alpha
    s2 = s2 * -alpha
    s1 = s1 + xy[i, j] * dv[i, j]
    nu = (1 - omega) * du[i, j] + a1[i, j] * (b1[i, j] - s1)
    s2 = s2 + xy[i, j] * nu
    nv = (1 - omega) * dv[i, j] + a2[i, j] * (b2[i, j] - s2)
    du[i, j] = nu
    dv[i, j] = nv


def exchange(planes, w, h, rows, cols, halo, rank, n):
    """every rank receives its tile grown by `halo` from the owners; plan = papof_tiles_halo_message on both ends"""
    ops, recvs = [], []
    for r in range(n):
        if r == rank:
            continue
        x0, y0, x1, y1 = capi.tiles_halo_message(w, h, rows, cols, halo, rank, r)  # what I send to r
        if x1 > x0 and y1 > y0:
            buf = torch.from_numpy(np.ascontiguousarray(np.stack([p[y0:y1, x0:x1] for p in planes])))
            ops.append(dist.P2POp(dist.isend, buf, r))
        x0, y0, x1, y1 = capi.tiles_halo_message(w, h, rows, cols, halo, r, rank)  # what r sends to me
        if x1 > x0 and y1 > y0:
            buf = torch.empty((len(planes), y1 - y0, x1 - x0), dtype=torch.float64)
            ops.append(dist.P2POp(dist.irecv, buf, r))
            recvs.append(((x0, y0, x1, y1), buf))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for (x0, y0, x1, y1), buf in recvs:
        for p, b in zip(planes, buf.numpy()):
            p[y0:y1, x0:x1] = b
    return len(recvs)


def main():
    h, w, n_sor, halo, rows, cols = (int(x) for x in sys.argv[1:7])
    dist.init_process_group(backend="gloo")
    rank, n = dist.get_rank(), dist.get_world_size()
    assert n == rows * cols
    alpha, omega = 0.012, 1.8
    rng = np.random.default_rng(7)  # every rank builds the same operand planes (the GPU path assembles them redundantly)
    phi = rng.uniform(0.5, 50.0, (h, w))
    xy = rng.uniform(-0.02, 0.02, (h, w))
    x2 = rng.uniform(0, 0.05, (h, w))
    y2 = rng.uniform(0, 0.05, (h, w))
    b1 = rng.uniform(-0.01, 0.01, (h, w))
    b2 = rng.uniform(-0.01, 0.01, (h, w))
    import sim_sor_wave as sim
    a1, a2 = sim.sor_coefficients(phi, x2, y2, alpha, omega)
    T = capi.tiles_rect(w, h, rows, cols, rank)
    P = (phi, xy, a1, a2, b1, b2)
    n_half, n_ex = 2 * n_sor, 0
    import sim_sor_blocked as blk
    gmax = int(sys.argv[7]) if len(sys.argv) > 7 else 10  # depth cap of one launch (the region allows 15 at 32 rows)
    su = sv = None
    hs = 0
    while hs < n_half:  # tiles.hip, round 2
        s = min(halo, n_half - hs)
        done = 0
        while done < s:
            g = min(gmax, s - done)
            su, sv = blk._launch(P, su, sv, h, w, g, hs + done, 1, 32, alpha, omega, out=grow(T, s - done - g, w, h))
            done += g
        hs += s
        if hs < n_half:
            exchange((su, sv), w, h, rows, cols, halo, rank, n)
            n_ex += 1
    du, dv = su, sv
    # cross-check against the independent half-sweep restatement above where no messages are involved (one rank)
    if n == 1:
        cu, cv = np.zeros((h, w)), np.zeros((h, w))
        for k in range(n_half):
            halfsweep(P, cu, cv, (0, 0, w, h), k & 1, alpha, omega, h, w)
        assert np.array_equal(cu, du) and np.array_equal(cv, dv)
    # gather the tiles on rank 0 (need = whole plane there: a halo as large as the plane)
    mine = np.ascontiguousarray(np.stack([du[T[1]:T[3], T[0]:T[2]], dv[T[1]:T[3], T[0]:T[2]]]))
    if rank == 0:
        for r in range(1, n):
            x0, y0, x1, y1 = capi.tiles_rect(w, h, rows, cols, r)
            buf = torch.empty((2, y1 - y0, x1 - x0), dtype=torch.float64)
            if buf.numel():
                dist.recv(buf, r)
                du[y0:y1, x0:x1], dv[y0:y1, x0:x1] = buf[0].numpy(), buf[1].numpy()
        from _libs import OracleLib
        eu, ev = OracleLib().sor(phi, xy, x2, y2, b1, b2, n_sor, alpha=alpha, omega=omega, mode=1)
        ok = bool(np.array_equal(du, eu) and np.array_equal(dv, ev))
        print("TILES_GLOO ok=%d exchanges=%d maxabs=%.3e" % (ok, n_ex, float(np.abs(du - eu).max())), flush=True)
    elif mine.size:
        dist.send(torch.from_numpy(mine), 0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
