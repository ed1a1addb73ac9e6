"""The EXACT-ORDER split of one frame pair over several ranks (csrc/tiles.hip: bands_flow; sor.hip: the SPLIT form of
k_sor_exact) on ONE GPU: all ranks are threads of this process, each with its own handle, arena, counters and streams
(LOCAL transport), and the solver kernel of a rank writes the one cell per step that crosses a cut -- and its progress --
straight into the planes and counters of the rank below.  Unlike the red-black tiles this keeps the reference's own sweep
order (src/OpticalFlow.cpp:458-505), so the result must be the REFERENCE's bits: compared here with the goldens the
untouched reference produced (1920x1080 config-4 schedule, 960x540 reference schedule), with the one-GPU exact call, and
with the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gpu():
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    yield g
    g.close()


def _params(**kw):
    from papteam_opticalflow_amd import default_params
    kw.setdefault("sor_mode", 0)
    return default_params(**kw)


def _run(nranks, a, b, levels, P, group=None):
    from papteam_opticalflow_amd.capi import LocalTileGroup
    grp = (group or LocalTileGroup)(nranks, nranks, 1, 0)
    try:
        out = grp.coarse2fine_flow(a, b, levels, P)
        stats = grp.ranks[0].stats()
    finally:
        grp.close()
    return out, stats


@pytest.mark.parametrize("nranks", [2, 8])
@pytest.mark.parametrize("case,res,levels,kw", [
    ("cfg4_1920_L5", "1920", 5, dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)),
    ("e2e_960_L5", "960", 5, {}),
])
def test_band_split_returns_the_reference_bits(case, res, levels, kw, nranks, tile_group):
    """whole calls with 2 and 8 ranks, array_equal to the reference's goldens: the SHA-256 of the
    full float64 arrays the untouched reference produced (golden.json), not only the strided subsample."""
    a, b = cases.load_pair(res)
    (vx, vy, wi, t), (n_ex, n_bytes) = _run(nranks, a, b, levels, _params(**kw), tile_group)
    staged = tile_group.__name__ == "RcclTileGroup"  # no peer addressing: one cut message per solve and cut on top
    man = json.load(open(os.path.join(GOLD, "golden.json")))["cases"][case]
    gold = np.load(os.path.join(GOLD, "golden.npz"))
    for name, got in (("vx", vx), ("vy", vy), ("warpI2", wi)):
        assert np.array_equal(cases.subsample(got), gold["%s|%s" % (case, name)]), \
            "%s/%s: max-abs %.3e vs the reference's subsample" % (case, name, np.abs(cases.subsample(got) - gold["%s|%s" % (case, name)]).max())
        assert cases.sha(got) == man[name]["sha"], "%s/%s: full-array SHA-256 differs from the reference's" % (case, name)
    n_solves = sum((kw.get("n_outer", 7) + k * kw.get("n_outer_per_level", 1)) for k in range(levels))
    if not staged:
        assert n_ex == n_solves + (levels - 1) + 2  # one (u, v) exchange per outer iteration, level changes, the final gather
    else:
        assert n_ex > n_solves + (levels - 1) + 2
    assert t[9] > 0 and t[6] > 0
    print("exact-order band split, %d ranks, %s: %d exchanges, %.1f MB moved by rank 0, SOR %.2f ms of %.2f ms on one device" %
          (nranks, case, n_ex, n_bytes / 1e6, t[6] * 1e3, t[9] * 1e3))


@pytest.mark.parametrize("nranks,res,levels,kw", [
    (3, "240", 5, {}),                                                       # coarse levels have fewer bands than ranks
    (8, "240", 4, dict(n_outer=2, n_outer_per_level=1, n_sor=9, n_sor_per_level=4)),   # 135 rows: at most 3 bands for 8 ranks
    (5, "480", 3, dict(n_outer=2, n_outer_per_level=0, n_sor=70, n_sor_per_level=5)),  # more sweeps than rows per band
    (2, "480", 5, dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0, alpha=0.02, omega=1.5)),
    (1, "240", 3, {}),
])
def test_band_split_equals_one_gpu_exact_call(gpu, tile_group, request, nranks, res, levels, kw):
    request.node._standin_may_be_idle = nranks == 1
    a, b = cases.load_pair(res)
    P = _params(**kw)
    want = gpu.coarse2fine_flow(a, b, levels, P)[:3]
    (vx, vy, wi, _), _ = _run(nranks, a, b, levels, P, tile_group)
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), want):
        assert np.array_equal(g, w), "%d ranks %s L%d %s %s: max-abs %.3e" % (nranks, res, levels, kw, name, np.abs(g - w).max())


def test_band_split_ragged_frame_matches_oracle(oracle, tile_group):
    a, b = cases.load_pair("480")
    a = np.ascontiguousarray(a[:203, :311])
    b = np.ascontiguousarray(b[:203, :311])
    kw = dict(n_outer=2, n_outer_per_level=1, n_sor=11, n_sor_per_level=2)
    (vx, vy, wi, _), _ = _run(4, a, b, 3, _params(**kw), tile_group)
    p = oracle.default_params()
    for k, v in kw.items():
        setattr(p, k, v)
    ow = oracle.coarse2fine_flow(a, b, 3, p)[:3]
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), ow):
        assert np.array_equal(g, w), "%s: max-abs %.3e" % (name, np.abs(g - w).max())


@pytest.mark.parametrize("nranks,res,levels,kw", [
    (2, "960", 5, dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)),
    (8, "480", 5, {}),
    (3, "240", 4, dict(n_outer=2, n_outer_per_level=1, n_sor=9, n_sor_per_level=4)),
])
def test_staged_protocol_of_the_rccl_transport_gives_the_same_bits(gpu, monkeypatch, nranks, res, levels, kw):
    """Transports that cannot address peer memory from a kernel (RCCL) run the split STAGED: a rank's kernel writes the cut
    cells into an outbox of its own, they travel as one message per solve and cut, and the rank below launches its bands
    after unpacking them (tiles.hip: bands_flow).  PAPOF_BANDS_STAGED=1 makes the LOCAL transport behave that way, so the
    code the RCCL transport runs -- everything but ncclSend / ncclRecv themselves -- is checked here bit for bit."""
    monkeypatch.setenv("PAPOF_BANDS_STAGED", "1")
    a, b = cases.load_pair(res)
    P = _params(**kw)
    want = gpu.coarse2fine_flow(a, b, levels, P)[:3]
    (vx, vy, wi, _), (n_ex, _) = _run(nranks, a, b, levels, P)
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), want):
        assert np.array_equal(g, w), "staged, %d ranks %s L%d %s: max-abs %.3e" % (nranks, res, levels, name, np.abs(g - w).max())
    assert n_ex > 0


@pytest.mark.parametrize("chunks,nranks,res,levels,kw", [
    (3, 2, "960", 5, dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)),
    (4, 8, "480", 5, {}),
    (7, 3, "240", 4, dict(n_outer=2, n_outer_per_level=1, n_sor=9, n_sor_per_level=4)),
    (64, 5, "240", 2, dict(n_outer=2, n_outer_per_level=0, n_sor=5, n_sor_per_level=3)),   # more chunks than sweeps: one sweep per launch
])
def test_staged_protocol_in_ranges_of_sweeps_gives_the_same_bits(gpu, monkeypatch, tile_group, chunks, nranks, res, levels, kw):
    """the staged protocol issues every rank's solve as launches over ranges of sweeps
    (PAPOF_BANDS_CHUNKS) and sends the cut cells of a range as soon as its launch has ended, so that rank g + 1 runs sweeps
    [k0, k1) while rank g runs [k1, k2).  Same bits as the one-GPU exact call for any number of ranges, on the LOCAL transport
    (forced staged) and on the RCCL transport bound to the stand-in, where nothing but stream order holds the ranges apart."""
    monkeypatch.setenv("PAPOF_BANDS_STAGED", "1")
    monkeypatch.setenv("PAPOF_BANDS_CHUNKS", str(chunks))
    a, b = cases.load_pair(res)
    P = _params(**kw)
    want = gpu.coarse2fine_flow(a, b, levels, P)[:3]
    (vx, vy, wi, _), (n_ex, _) = _run(nranks, a, b, levels, P, tile_group)
    for name, g, w in zip(("vx", "vy", "warpI2"), (vx, vy, wi), want):
        assert np.array_equal(g, w), "%d ranges, %d ranks %s L%d %s: max-abs %.3e" % (chunks, nranks, res, levels, name, np.abs(g - w).max())


def test_chunked_staged_split_returns_the_reference_bits_at_1080p(monkeypatch, tile_group):
    """... and at size: 8 ranks, 3 ranges of sweeps per solve, the reference's golden of the 1920x1080 config-4 pair (full-array SHA)."""
    monkeypatch.setenv("PAPOF_BANDS_STAGED", "1")
    monkeypatch.setenv("PAPOF_BANDS_CHUNKS", "3")
    a, b = cases.load_pair("1920")
    kw = dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)
    (vx, vy, wi, _), _ = _run(8, a, b, 5, _params(**kw), tile_group)
    man = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]["cfg4_1920_L5"]
    for name, got in (("vx", vx), ("vy", vy), ("warpI2", wi)):
        assert cases.sha(got) == man[name]["sha"], "cfg4_1920_L5/%s: full-array SHA-256 differs from the reference's" % name


def test_exact_split_over_the_rccl_transport_with_one_rank(gpu):
    """The RCCL transport itself with sor_mode = 0 (a group of one: no cut, no sends -- the plumbing and the dispatch)."""
    from papteam_opticalflow_amd import capi
    a, b = cases.load_pair("240")
    P = _params()
    tr = capi.TileRank.create(gpu, capi.tiles_unique_id(), 0, 1)
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
    for p_ in (d1, d2, dx, dy, dw):
        gpu.dev_free(p_)
    want = gpu.coarse2fine_flow(a, b, 3, P)
    assert np.array_equal(vx, want[0]) and np.array_equal(wi, want[2])


def test_band_split_rejects_what_it_does_not_cover(gpu):
    from papteam_opticalflow_amd import PapofError
    a, b = cases.load_pair("240")
    for kw in (dict(n_inner=2), dict(interpolation=1), dict(noise_model=1), dict(n_sor=129)):
        with pytest.raises(PapofError):
            _run(2, a, b, 2, _params(**kw))


@pytest.mark.parametrize("nranks", [2, 3])
def test_band_split_proves_the_laplacian_noise_guard_or_refuses(gpu, tile_group, nranks):
    """The band split runs without the noise estimate, so it must PROVE that the reference's `LapPara < 1E-20` guard
    (src/OpticalFlow.cpp:399-400) cannot have tripped -- every rank checks every pixel of its rows behind every update, the
    flags of all ranks are gathered -- or refuse the pair.  Ordinary frames: proven (the other tests of this file).  Duplicate
    frames: no valid sample on any rank, proven, and the bits of the one-GPU call.  A pair scaled to 1e-21, on which the guard
    does trip: every rank returns an error that names the one-GPU call."""
    from papteam_opticalflow_amd import PapofError
    a, b = cases.load_pair("240")
    P = _params()
    (vx, vy, wi, _), _ = _run(nranks, a, a, 3, P, tile_group)
    want = gpu.coarse2fine_flow(a, a, 3, P)
    assert np.array_equal(vx, want[0]) and np.array_equal(vy, want[1]) and np.array_equal(wi, want[2])
    assert not vx.any() and not vy.any()
    with pytest.raises(PapofError) as e:
        _run(nranks, np.ascontiguousarray(a * 1e-21), np.ascontiguousarray(b * 1e-21), 3, P, tile_group)
    assert "guard" in str(e.value) and "one GPU" in str(e.value)


def test_a_peer_that_never_publishes_ends_in_a_timeout_not_a_hang(monkeypatch):
    """The bounded waits across the cut: rank 0 never launches its solver kernels (PAPOF_BANDS_SILENT_RANK), so the first
    band of rank 1 never sees progress of the band above it.  Its tasks must give up after their bounded spin, raise their
    rank's abort word -- every other task of that rank ends when it sees it -- and the call must come back with
    PAPOF_ETIMEOUT (-5) within seconds instead of hanging or returning numbers as if nothing had happened."""
    import time
    from papteam_opticalflow_amd import PapofError
    a, b = cases.load_pair("480")
    monkeypatch.setenv("PAPOF_BANDS_SILENT_RANK", "0")
    t0 = time.time()
    with pytest.raises(PapofError) as err:
        _run(2, a, b, 1, _params(n_outer=1, n_outer_per_level=0, n_sor=5, n_sor_per_level=0))
    assert err.value.code == -5, err.value
    assert time.time() - t0 < 120
    monkeypatch.delenv("PAPOF_BANDS_SILENT_RANK")
    (vx, vy, wi, _), _ = _run(2, a, b, 1, _params(n_outer=1, n_outer_per_level=0, n_sor=5, n_sor_per_level=0))  # and works afterwards
    assert np.isfinite(vx).all()


def test_a_silent_peer_on_the_rccl_transport_ends_in_a_timeout_not_a_hang(monkeypatch, tile_group, request):
    """The same fault on the transport the multi-GPU node runs (staged protocol: the cut cells of a solve travel as one message
    behind the producer's kernel).  Rank 0 neither launches its solver kernels nor sends its cut cells; the receive of rank 1 is a
    kernel on its stream that nothing will ever complete.  The call's final wait polls the stream against PAPOF_TILES_TIMEOUT_S,
    aborts the communicator (ncclCommAbort: the library's kernels in flight exit) and returns PAPOF_ETIMEOUT on every rank --
    rank 0 runs into the same deadline in its next exchange with the rank that gave up.  A new group works afterwards."""
    import time
    from papteam_opticalflow_amd import PapofError
    if tile_group.__name__ != "RcclTileGroup":
        pytest.skip("the LOCAL transport's variant is test_a_peer_that_never_publishes_ends_in_a_timeout_not_a_hang")
    request.node._standin_errors_expected = True  # the missing send shifts the later messages of that pair: the stand-in objects
    a, b = cases.load_pair("480")
    P = _params(n_outer=1, n_outer_per_level=0, n_sor=5, n_sor_per_level=0)
    monkeypatch.setenv("PAPOF_BANDS_SILENT_RANK", "0")
    monkeypatch.setenv("PAPOF_TILES_TIMEOUT_S", "4")
    t0 = time.time()
    with pytest.raises(PapofError) as err:
        _run(2, a, b, 1, P, tile_group)
    assert err.value.code == -5, err.value
    assert time.time() - t0 < 60
    monkeypatch.delenv("PAPOF_BANDS_SILENT_RANK")
    monkeypatch.delenv("PAPOF_TILES_TIMEOUT_S")
    (vx, vy, wi, _), _ = _run(2, a, b, 1, P, tile_group)
    want, _ = _run(2, a, b, 1, P)
    assert np.array_equal(vx, want[0]) and np.array_equal(vy, want[1])


@pytest.mark.parametrize("seed", range(3))
def test_band_split_random_shapes_ranks_and_schedules(gpu, tile_group, seed):
    """Random frame sizes (one to five solver bands), rank counts (more ranks than bands included), pyramid depths and schedules:
    the split returns the bits of the one-GPU exact call."""
    rng = np.random.default_rng(3000 + seed)
    a0, b0 = cases.load_pair("480")
    for case in range(4):
        h, w = int(rng.integers(40, 271)), int(rng.integers(40, 301))
        y0, x0 = int(rng.integers(0, 270 - h + 1)), int(rng.integers(0, 480 - w + 1))
        a, b = np.ascontiguousarray(a0[y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(b0[y0:y0 + h, x0:x0 + w])
        levels, nranks = int(rng.integers(1, 5)), int(rng.integers(2, 7))
        kw = dict(n_outer=int(rng.integers(1, 4)), n_outer_per_level=int(rng.integers(0, 2)), n_sor=int(rng.integers(2, 50)),
                  n_sor_per_level=int(rng.integers(0, 3)))
        print("seed %d case %d: %dx%d L%d ranks %d %s" % (seed, case, h, w, levels, nranks, kw), flush=True)
        P = _params(**kw)
        (vx, vy, wi, _), _ = _run(nranks, a, b, levels, P, tile_group)
        want = gpu.coarse2fine_flow(a, b, levels, P)
        assert np.array_equal(vx, want[0]) and np.array_equal(vy, want[1]) and np.array_equal(wi, want[2]), (seed, case)
