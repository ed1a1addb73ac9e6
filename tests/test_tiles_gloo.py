"""N>1 path of the tiled solver rehearsed on the CPU (world_size 2 and 4, torch.distributed `gloo`): message plan from
the product library's host functions, ghost-zone schedule as in csrc/tiles.hip, result bit-identical to the oracle's
red-black solve.  Also the host-side geometry: tiles cover the plane exactly once, and what `src` sends `dst` is
exactly the part of dst's grown tile that src owns."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("h,w,n_sor,halo,rows,cols,gmax", [(45, 64, 7, 4, 1, 2, 10), (50, 37, 5, 3, 2, 1, 10),
                                                            (61, 70, 6, 5, 2, 2, 10), (40, 40, 4, 1, 1, 2, 10),
                                                            (70, 300, 9, 7, 2, 2, 5), (33, 21, 3, 6, 1, 1, 4)])
def test_ghost_zone_halo_exchange_over_gloo(h, w, n_sor, halo, rows, cols, gmax):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(rows * cols),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_tiles_gloo_worker.py")] + [str(x) for x in (h, w, n_sor, halo, rows, cols, gmax)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    m = re.search(r"TILES_GLOO ok=(\d) exchanges=(\d+)", out.stdout)
    assert m, out.stdout + out.stderr[-2000:]
    assert m.group(1) == "1", out.stdout
    assert int(m.group(2)) == (2 * n_sor - 1) // halo  # one exchange per `halo` half-sweeps, none after the last


@pytest.mark.parametrize("w,h,rows,cols", [(1920, 1080, 2, 4), (607, 341, 2, 4), (75, 42, 2, 4), (5, 3, 2, 4),
                                            (101, 56, 3, 3), (240, 135, 1, 2)])
def test_tiles_partition_and_message_plan(w, h, rows, cols):
    from papteam_opticalflow_amd import capi
    n = rows * cols
    cover = np.zeros((h, w), dtype=np.int32)
    rects = [capi.tiles_rect(w, h, rows, cols, r) for r in range(n)]
    for x0, y0, x1, y1 in rects:
        assert 0 <= x0 <= x1 <= w and 0 <= y0 <= y1 <= h
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()  # disjoint cover
    for halo in (1, 4, 13):
        for dst in range(n):
            x0, y0, x1, y1 = rects[dst]
            want = np.zeros((h, w), dtype=bool)
            if x1 > x0 and y1 > y0:
                want[max(0, y0 - halo):min(h, y1 + halo), max(0, x0 - halo):min(w, x1 + halo)] = True
                want[y0:y1, x0:x1] = False
            got = np.zeros((h, w), dtype=np.int32)
            for src in range(n):
                a0, b0, a1, b1 = capi.tiles_halo_message(w, h, rows, cols, halo, src, dst)
                if a1 > a0 and b1 > b0:
                    assert src != dst
                    sx0, sy0, sx1, sy1 = rects[src]
                    assert sx0 <= a0 and a1 <= sx1 and sy0 <= b0 and b1 <= sy1  # only what the sender owns
                    got[b0:b1, a0:a1] += 1
            assert ((got == 1) == want).all() and got.max() <= 1  # the halo ring, each cell from exactly one owner


def test_default_grids():
    from papteam_opticalflow_amd import capi
    assert capi.tiles_grid(8) == (2, 4) and capi.tiles_grid(4) == (2, 2) and capi.tiles_grid(2) == (1, 2)
    assert capi.tiles_grid(1) == (1, 1) and capi.tiles_grid(6) == (2, 3) and capi.tiles_grid(7) == (1, 7)
