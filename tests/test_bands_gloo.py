"""N>1 path of the EXACT-ORDER band split rehearsed on the CPU (world_size 2, 3 and 4, torch.distributed `gloo`): the split
from the product library's host function (papof_bands_plan), the staged protocol of csrc/tiles.hip's bands_flow with real
inter-process messages, result bit-identical to the oracle's lexicographic solve.  Also the host-side geometry: the
finally owned rows partition the plane, and the coefficient rows cover what a rank's tasks touch."""
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


@pytest.mark.parametrize("h,w,n_sor,world,chunks", [(150, 23, 7, 2, 1), (200, 17, 5, 3, 1), (260, 12, 9, 4, 1), (70, 31, 4, 4, 1),
                                                     (130, 9, 70, 2, 1), (200, 17, 6, 3, 3), (150, 23, 7, 2, 64)])
def test_exact_order_band_split_over_gloo(h, w, n_sor, world, chunks):
    """chunks > 1: the staged protocol in ranges of sweeps (PAPOF_BANDS_CHUNKS of bands_flow) -- the rows of a range travel as soon
    as the range is done, the ranks work on different ranges at the same time; same bits, `chunks` messages per cut."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_bands_gloo_worker.py")] + [str(x) for x in (h, w, n_sor, 11 * h + w, chunks)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    m = re.search(r"BANDS_GLOO ok=(\d) messages=(\d+) ranks_with_bands=(\d+)", out.stdout)
    assert m, out.stdout + out.stderr[-2000:]
    assert m.group(1) == "1", out.stdout
    assert int(m.group(2)) == 2 * (int(m.group(3)) - 1) * min(chunks, n_sor)  # one message per cut and range (counted at both ends)


@pytest.mark.parametrize("h,w,n_sor,n", [(1080, 1920, 30, 8), (1080, 1920, 30, 2), (341, 607, 42, 8), (135, 240, 30, 8),
                                          (42, 75, 42, 8), (2, 4, 72, 3), (540, 960, 128, 5)])
def test_band_split_geometry(h, w, n_sor, n):
    from papteam_opticalflow_amd import capi
    BR = 62
    plans = [capi.bands_plan(h, w, n_sor, n, r) for r in range(n)]
    nb = capi.bands_plan(h, w, n_sor, 1, 0)["B1"]
    assert nb == (h + n_sor - 1 + BR - 1) // BR
    cover = np.zeros(h, dtype=int)
    prev_b1 = 0
    for p in plans:
        assert p["B0"] == prev_b1  # consecutive ranges of bands
        prev_b1 = p["B1"]
        y0, y1 = p["final_rows"]
        cover[y0:y1] += 1
        if p["B1"] > p["B0"]:
            c0, c1 = p["coef_rows"]
            rows = [BR * b - k - 1 + lane for b in range(p["B0"], p["B1"]) for k in range(n_sor) for lane in range(0, 63)]
            rows = [r for r in rows if 0 <= r < h]
            assert rows and min(rows) >= c0 and max(rows) < c1 and c0 <= y0 and y1 <= max(c1, y1)
            assert c0 == max(0, min(h, BR * p["B0"] - n_sor)) and c1 == min(h, BR * p["B1"])
        else:
            assert y1 == y0
    assert prev_b1 == nb and (cover == 1).all()  # every band runs somewhere; the final rows are a partition
