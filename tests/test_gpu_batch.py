"""B frame pairs of one shape in ONE launch chain (csrc/batch.hip; include/papof.h: papof_flow_batch*) -- the reference's
benchmark walks collections of small frames (Code/Serial/TestSuite.py:69-81, :91).  Every pair of a batch must come back with
the BITS of the single call on that pair -- which the other GPU tests pin to the oracle and to the reference's goldens -- for
consecutive pairs of a video and for independent pairs, uint8 and float64 frames, gray frames, ragged shapes, deep pyramids
(few-pixel levels: the guard's re-run path), other schedules; what the batched chain does not cover runs as single calls
through the same entry point."""
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


def _video(res, n, crop=None):
    """n frames that all differ: the two decoded frames of the reference's collection, shifted copies of them"""
    a, b = cases.load_frame_u8(res, 1), cases.load_frame_u8(res, 2)
    out = []
    for i in range(n):
        f = np.roll(a if i % 2 == 0 else b, (i // 2) * 3, axis=1)
        if crop:
            f = f[:crop[0], :crop[1]]
        out.append(np.ascontiguousarray(f))
    return out


def _same(got, want, what):
    for name, g, w in zip(("vx", "vy", "warpI2"), got, want):
        assert np.array_equal(g, w), "%s %s: max-abs %.3e" % (what, name, np.abs(g - w).max())


@pytest.mark.parametrize("res,n_pairs,levels,kw", [
    ("240", 5, 5, {}),                                                       # the reference's default collection shape
    ("240", 16, 3, dict(n_outer=2, n_outer_per_level=0, n_sor=9, n_sor_per_level=2)),
    ("480", 3, 5, dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)),   # config-4 schedule
    ("240", 2, 8, {}),                                                       # the reference's 8-level height
])
def test_batch_of_consecutive_pairs_equals_the_single_calls(gpu, res, n_pairs, levels, kw):
    from papteam_opticalflow_amd import default_params
    frames = _video(res, n_pairs + 1)
    P = default_params(**kw) if kw else None
    out, t = gpu.flow_batch(frames, levels, P, sequence=True)
    assert len(out) == n_pairs and t[9] > 0 and t[6] > 0
    for i in range(n_pairs):
        want = gpu.coarse2fine_flow_u8(frames[i], frames[i + 1], levels, P)[:3]
        _same(out[i], want, "pair %d of %d, %s L%d" % (i, n_pairs, res, levels))


def test_batch_of_independent_pairs_float64_and_gray(gpu):
    frames = [f.astype(np.float64) / 255.0 for f in _video("240", 6, crop=(101, 173))]
    out, _ = gpu.flow_batch(frames, 3, None, sequence=False)
    assert len(out) == 3
    for i in range(3):
        _same(out[i], gpu.coarse2fine_flow(frames[2 * i], frames[2 * i + 1], 3)[:3], "independent pair %d" % i)
    gray = [np.ascontiguousarray(f[..., 1:2]) for f in _video("240", 4)]
    out, _ = gpu.flow_batch(gray, 4, None, sequence=True)
    for i in range(3):
        _same(out[i], gpu.coarse2fine_flow_u8(gray[i], gray[i + 1], 4)[:3], "gray pair %d" % i)


def test_batch_matches_the_oracle_directly(gpu, oracle):
    frames = _video("240", 4, crop=(90, 120))
    out, _ = gpu.flow_batch(frames, 3, None, sequence=True)
    for i in range(3):
        a, b = frames[i].astype(np.float64) / 255.0, frames[i + 1].astype(np.float64) / 255.0
        _same(out[i], oracle.coarse2fine_flow(a, b, 3)[:3], "pair %d vs the oracle" % i)


def test_batch_reruns_a_pair_whose_guard_it_cannot_prove(gpu):
    """Duplicate frames inside a video: |Im1 - warpIm2| is exactly 0 everywhere, no witness exists, the pair goes through the
    single call (which has the guard's exact pass) -- zero flow, and the neighbours' results untouched."""
    v = _video("240", 4)
    frames = [v[0], v[1], v[1], v[2]]
    before = gpu.lap_guard_stats()["reruns"]
    out, _ = gpu.flow_batch(frames, 3, None, sequence=True)
    assert gpu.lap_guard_stats()["reruns"] > before
    assert not out[1][0].any() and not out[1][1].any()
    for i in (0, 2):
        _same(out[i], gpu.coarse2fine_flow_u8(frames[i], frames[i + 1], 3)[:3], "neighbour %d of a duplicate pair" % i)


def test_deep_pyramid_batch_and_what_the_chain_does_not_cover(gpu):
    """15 levels (the reference's deepest height: levels of a few pixels) and a non-default branch (red-black order): same
    entry point, same bits as the single calls."""
    from papteam_opticalflow_amd import default_params
    frames = _video("240", 3)
    before = gpu.lap_guard_stats()["reruns"]
    out, _ = gpu.flow_batch(frames, 15, None, sequence=True)
    assert gpu.lap_guard_stats()["reruns"] == before, "few-pixel levels: the exhaustive check must prove the guard open, no re-run"
    for i in range(2):
        _same(out[i], gpu.coarse2fine_flow_u8(frames[i], frames[i + 1], 15)[:3], "15 levels, pair %d" % i)
    P = default_params(sor_mode=1)
    out, _ = gpu.flow_batch(frames, 3, P, sequence=True)
    for i in range(2):
        _same(out[i], gpu.coarse2fine_flow_u8(frames[i], frames[i + 1], 3, P)[:3], "red-black, pair %d" % i)
    with pytest.raises(ValueError):
        gpu.flow_batch(frames[:1], 3)


def test_batch_is_faster_per_pair_than_single_calls(gpu):
    import time
    frames = _video("240", 17)
    gpu.flow_batch(frames, 5)  # arena, counters
    t0 = time.perf_counter()
    gpu.flow_batch(frames, 5)
    per_pair_batch = (time.perf_counter() - t0) / 16
    gpu.coarse2fine_flow_u8(frames[0], frames[1], 5)
    t0 = time.perf_counter()
    for i in range(4):
        gpu.coarse2fine_flow_u8(frames[i], frames[i + 1], 5)
    per_pair_single = (time.perf_counter() - t0) / 4
    print("240x135, reference schedule: %.2f ms per pair in a batch of 16, %.2f ms per single call" %
          (per_pair_batch * 1e3, per_pair_single * 1e3))
    assert per_pair_batch < 0.5 * per_pair_single


def test_flow_collection_batched_equals_unbatched(gpu):
    """flow_collection(batch=B): chains of B consecutive pairs per launch chain, two chains in flight, a ragged last chain --
    pair for pair the bits of the unbatched collection and of the single call."""
    from papteam_opticalflow_amd import flow_collection
    frames = _video("240", 12)
    want = flow_collection(frames, 3, batch=0, in_flight=2)
    got = flow_collection(frames, 3, batch=4, in_flight=2)
    assert len(got) == len(want) == 11
    for i, (g, w) in enumerate(zip(got, want)):
        assert sorted(g[0]) == sorted(w[0])  # the ten timing keys, string-valued
        _same(g[1:], w[1:], "collection pair %d" % i)
    seen = {}
    flow_collection(frames, 3, batch=5, on_pair=lambda i, t, vx, vy, wi: seen.__setitem__(i, (vx.copy(), vy.copy(), wi.copy())))
    assert sorted(seen) == list(range(11))
    for i in range(11):
        _same(seen[i], want[i][1:], "on_pair %d" % i)


def test_a_collection_larger_than_one_launch_holds_goes_through_in_sub_batches(gpu, monkeypatch):
    """More pairs than one launch of the solver may hold go through in sub-batches (their size follows the band count and the
    arena; PAPOF_BATCH_MAX caps it -- 3 here: 7 pairs as 3 + 2 + 2, never a batch of one): same bits, pair for pair.  A frame too
    tall for the batched chain (>= 16 solver bands) runs as single calls through the same entry point."""
    frames = _video("240", 8)
    monkeypatch.setenv("PAPOF_BATCH_MAX", "3")
    out, _ = gpu.flow_batch(frames, 3, None, sequence=True)
    monkeypatch.delenv("PAPOF_BATCH_MAX")
    assert len(out) == 7
    for i in range(7):
        _same(out[i], gpu.coarse2fine_flow_u8(frames[i], frames[i + 1], 3)[:3], "sub-batched pair %d" % i)
    tall = [np.ascontiguousarray(np.tile(f, (8, 1, 1))[:1000, :64]) for f in frames[:3]]  # 1000 rows: 17 bands
    out, _ = gpu.flow_batch(tall, 2, None, sequence=True)
    for i in range(2):
        _same(out[i], gpu.coarse2fine_flow_u8(tall[i], tall[i + 1], 2)[:3], "tall frame, pair %d" % i)
