"""The tiled multi-GPU path (csrc/tiles.hip; SURVEY.md §8e, BASELINE.json configs[4]) on ONE GPU: all ranks of the tile
group live in this process (LOCAL transport: device copies in place of the RCCL sends), each driven by its own thread --
the orchestration, the regions every kernel runs on, the message plans and the ghost-zone schedule are the code the
RCCL transport runs.  A red-black half-sweep reads only the other colour's previous values, so the tiled solve must be
BIT-IDENTICAL to the one-GPU red-black solve, which in turn is checked against the oracle run in the same mode."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    yield g
    g.close()


def _params(**kw):
    from papteam_opticalflow_amd import default_params
    kw.setdefault("sor_mode", 1)
    return default_params(**kw)


def _run_tiles(nranks, rows, cols, halo, a, b, levels, P, group=None):
    from papteam_opticalflow_amd.capi import LocalTileGroup
    grp = (group or LocalTileGroup)(nranks, rows, cols, halo)
    try:
        out = grp.coarse2fine_flow(a, b, levels, P)
        stats = grp.ranks[0].stats()
    finally:
        grp.close()
    return out, stats


@pytest.mark.parametrize("res,levels,rows,cols,halo", [
    ("240", 3, 1, 2, 4), ("240", 5, 2, 2, 10), ("240", 4, 2, 4, 6), ("240", 3, 3, 1, 1), ("480", 5, 2, 4, 10),
    ("240", 2, 1, 3, 7),
])
def test_tiled_equals_single_gpu_redblack(gpu, tile_group, res, levels, rows, cols, halo):
    a, b = cases.load_pair(res)
    P = _params()
    want = gpu.coarse2fine_flow(a, b, levels, P)[:3]
    (vx, vy, wi, t), (n_ex, n_bytes) = _run_tiles(rows * cols, rows, cols, halo, a, b, levels, P, tile_group)
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), want):
        assert np.array_equal(g, w), "%s differs: max-abs %.3e" % (name, np.abs(g - w).max())
    assert t[9] > 0 and t[6] > 0 and n_ex > 0 and n_bytes > 0
    print("tiles %dx%d halo %d on %s L%d: %d exchanges, %.2f MB moved by rank 0" % (rows, cols, halo, res, levels, n_ex,
                                                                                     n_bytes / 1e6))


def test_tiled_ragged_sizes_and_schedule(gpu, oracle):
    a, b = cases.load_pair("240")
    a = np.ascontiguousarray(a[:101, :173])
    b = np.ascontiguousarray(b[:101, :173])
    kw = dict(n_outer=2, n_outer_per_level=1, n_sor=7, n_sor_per_level=2, alpha=0.02, omega=1.5)
    P = _params(**kw)
    (vx, vy, wi, _), _ = _run_tiles(6, 2, 3, 5, a, b, 3, P)
    want = gpu.coarse2fine_flow(a, b, 3, P)[:3]
    for g, w in zip((vx, vy, wi), want):
        assert np.array_equal(g, w)
    p = oracle.default_params()
    for k, v in dict(kw, sor_mode=1).items():
        setattr(p, k, v)
    ow = oracle.coarse2fine_flow(a, b, 3, p)[:3]
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), ow):
        assert np.abs(g - w).max() <= 1e-9, name


def test_tiled_2x4_matches_the_oracle_itself(gpu, oracle, tile_group):
    """BASELINE.json configs[4] geometry (2 x 4 tiles, LOCAL transport) compared with the ORACLE run in the same
    (red-black) mode -- not only with the one-GPU result: 480x270 pair, config-4 schedule (3 outer / 30 sweeps), 5 levels."""
    a, b = cases.load_pair("480")
    kw = dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)
    (vx, vy, wi, _), _ = _run_tiles(8, 2, 4, 10, a, b, 5, _params(**kw), tile_group)
    p = oracle.default_params()
    for k, v in dict(kw, sor_mode=1).items():
        setattr(p, k, v)
    want = oracle.coarse2fine_flow(a, b, 5, p)[:3]
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), want):
        assert np.array_equal(g, w), "%s: max-abs %.3e" % (name, np.abs(g - w).max())


def test_tiled_config4_schedule_full_hd_tile_grid(gpu):
    """BASELINE.json configs[4] geometry (2 x 4 tiles) with the config-4 schedule at 960x540 (full HD is the bench)."""
    a, b = cases.load_pair("960")
    P = _params(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)
    (vx, vy, wi, _), (n_ex, _) = _run_tiles(8, 2, 4, 10, a, b, 5, P)
    want = gpu.coarse2fine_flow(a, b, 5, P)[:3]
    for g, w in zip((vx, vy, wi), want):
        assert np.array_equal(g, w)
    assert n_ex == 15 * (5 + 1) + 4 + 2  # per solve: 60 half-sweeps / 10 - 1 exchanges + (u, v); level changes; gather


def test_config5_1920x1080_2x4_tiles_at_size(gpu, oracle, tile_group):
    """BASELINE.json configs[4] AT ITS SIZE, as far as one GPU allows: the 1920x1080 pair, 5 levels, config-4 schedule
    (3 outer / 30 SOR), sharded as 2 x 4 tiles with ghost zones 10 half-sweeps deep -- the eight ranks are threads of this
    process on one device (LOCAL transport: device copies where the multi-GPU run has RCCL sends; orchestration, regions,
    message plans and the ghost-zone schedule are the same code).  Must be bit-identical to the one-GPU red-black call AND
    to the oracle run in the same mode (~7 s of one host core), with the exchange count DESIGN.md 7 states: per solve
    60 / 10 - 1 = 5 (du, dv) exchanges + 1 of (u, v) = 6, x 15 solves, + 4 level changes + 2 for the final gather = 96."""
    a, b = cases.load_pair("1920")
    kw = dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)
    P = _params(**kw)
    (vx, vy, wi, t), (n_ex, n_bytes) = _run_tiles(8, 2, 4, 10, a, b, 5, P, tile_group)
    want = gpu.coarse2fine_flow(a, b, 5, P)[:3]
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), want):
        assert np.array_equal(g, w), "%s differs from the one-GPU red-black call: max-abs %.3e" % (name, np.abs(g - w).max())
    assert n_ex == 15 * (5 + 1) + 4 + 2 == 96
    p = oracle.default_params()
    for k, v in dict(kw, sor_mode=1).items():
        setattr(p, k, v)
    ow = oracle.coarse2fine_flow(a, b, 5, p)[:3]
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), ow):
        assert np.array_equal(g, w), "%s differs from the same-mode oracle: max-abs %.3e" % (name, np.abs(g - w).max())
    print("config 5 at size: 2x4 tiles of 480x540, %d exchanges, %.1f MB moved by rank 0, %.1f ms on one device" %
          (n_ex, n_bytes / 1e6, t[9] * 1e3))


def test_tiled_rejects_jacobi_and_inner_iterations(gpu):
    """(sor_mode = 0, the reference's own order, is no longer rejected: it runs as the exact-order split into ranges of
    solver bands, tests/test_gpu_bands.py)"""
    from papteam_opticalflow_amd import PapofError
    a, b = cases.load_pair("240")
    for kw in (dict(sor_mode=2), dict(n_inner=2)):
        with pytest.raises(PapofError):
            _run_tiles(2, 1, 2, 4, a, b, 2, _params(**kw))


def test_rccl_transport_loads_and_runs_a_group_of_one(gpu):
    """The RCCL transport itself (dlopen of librccl, ncclGetUniqueId, ncclCommInitRank) on the one GPU this box has:
    a group of one rank has no peers, so this covers the library plumbing, not the sends (those need >= 2 GPUs and are
    exercised by bench.py --gpus N on the multi-GPU node)."""
    from papteam_opticalflow_amd import capi
    a, b = cases.load_pair("240")
    P = _params()
    uid = capi.tiles_unique_id()
    assert len(uid) == capi.TILES_ID_BYTES and any(uid)
    tr = capi.TileRank.create(gpu, uid, 0, 1)
    h, w, c = a.shape
    d1, d2 = gpu.dev_alloc(a.nbytes), gpu.dev_alloc(b.nbytes)
    dx, dy, dw = gpu.dev_alloc(h * w * 8), gpu.dev_alloc(h * w * 8), gpu.dev_alloc(a.nbytes)
    gpu.dev_upload(d1, a)
    gpu.dev_upload(d2, b)
    tr.flow_device(d1, d2, h, w, c, 3, P, dx, dy, dw)
    vx, wi = np.zeros((h, w)), np.zeros((h, w, c))
    gpu.dev_download(vx, dx)
    gpu.dev_download(wi, dw)
    tr.close()
    for p in (d1, d2, dx, dy, dw):
        gpu.dev_free(p)
    want = gpu.coarse2fine_flow(a, b, 3, P)
    assert np.array_equal(vx, want[0]) and np.array_equal(wi, want[2])


def test_bench_tiles_reporting_path_single_rank():
    """bench.py's secondary tiled measurement (what `--gpus N` adds to the JSON line), forced on the one GPU of this box:
    PyTorch process group of one, RCCL id + communicator, tile group of one, live bit-identity check, and still
    exactly ONE line on stdout (RCCL prints its banner to stdout when a communicator is created)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--res", "240", "--steps", "2", "--warmup", "1",
                          "--force-tiles", "--no-cpu-baseline", "--no-concurrent"], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    t = r["tiles"]
    assert "error" not in t, t
    assert t["grid"] == "1x1" and t["bit_identical_to_one_gpu_redblack"] is True and t["value"] > 0
    assert t["n_ranks_seen"] == 1 and t["ranks_seen"] == [0] and t["tile_grid_in_use"] == "1x1"
    assert len(t["sor_ms_per_pair_by_rank"]) == 1 and t["sor_ms_per_pair_by_rank"][0] > 0
    assert r["scaling"] == "weak" and t["scaling"] == "strong" and r["value"] > 0
    ex = t["exact_order"]  # the same pair in the reference's sweep order, split over the ranks (bands_flow)
    assert "error" not in ex, ex
    assert ex["bit_identical_to_one_gpu_exact"] is True and ex["value"] > 0 and ex["n_ranks_seen"] == 1


@pytest.mark.parametrize("seed", range(3))
def test_tiles_random_grids_halos_and_shapes(gpu, tile_group, seed):
    """Random frame sizes, tile grids, ghost-zone depths, pyramid depths and schedules: whatever the tiled path accepts must be
    bit-identical to the one-GPU red-black call (a grid it refuses is refused with PAPOF_EINVAL, not computed wrongly)."""
    from papteam_opticalflow_amd import PapofError
    rng = np.random.default_rng(4000 + seed)
    a0, b0 = cases.load_pair("480")
    done = 0
    for case in range(6):
        h, w = int(rng.integers(48, 271)), int(rng.integers(64, 481))
        y0, x0 = int(rng.integers(0, 270 - h + 1)), int(rng.integers(0, 480 - w + 1))
        a, b = np.ascontiguousarray(a0[y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(b0[y0:y0 + h, x0:x0 + w])
        rows, cols, halo = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 11))
        levels = int(rng.integers(1, 4))
        kw = dict(n_outer=int(rng.integers(1, 3)), n_outer_per_level=int(rng.integers(0, 2)), n_sor=int(rng.integers(1, 25)),
                  n_sor_per_level=int(rng.integers(0, 3)))
        print("seed %d case %d: %dx%d L%d grid %dx%d halo %d %s" % (seed, case, h, w, levels, rows, cols, halo, kw), flush=True)
        P = _params(**kw)
        try:
            (vx, vy, wi, _), _ = _run_tiles(rows * cols, rows, cols, halo, a, b, levels, P, tile_group)
        except PapofError as e:
            assert e.code == -1, e  # PAPOF_EINVAL: a grid / halo the path does not take
            continue
        want = gpu.coarse2fine_flow(a, b, levels, P)
        assert np.array_equal(vx, want[0]) and np.array_equal(vy, want[1]) and np.array_equal(wi, want[2]), (seed, case)
        done += 1
    assert done >= 2
