"""Parity of the callers' side of the path, through the C ABI, on the GPU:

* uint8 frames: the device-side `/ 255.` must give the bits of the caller's `im.astype(float) / 255.`
  (Code/Serial/OpticalFlowCalculation.py:65-70), i.e. results identical to the fp64 entry point and to the oracle;
* sequence mode: pushing frames n, n+1, n+2 (TestSuite.py:69-81 walks overlapping pairs) must return, pair by pair,
  exactly what independent coarse2fine_flow calls return.
"""
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


def _frames(res, n=3):
    return [cases.load_frame_u8(res, i + 1) for i in range(n)]


def _f64(u8):
    return u8.astype(np.float64) / 255.0


@pytest.mark.parametrize("res,levels", [("240", 3), ("480", 5)])
def test_u8_entry_point_is_bit_identical_to_f64_and_oracle(gpu, oracle, res, levels):
    a8, b8 = _frames(res, 2)
    got = gpu.coarse2fine_flow_u8(a8, b8, levels)
    ref = gpu.coarse2fine_flow(_f64(a8), _f64(b8), levels)
    want = oracle.coarse2fine_flow(_f64(a8), _f64(b8), levels)
    for name, g, r, w in zip(("vx", "vy", "warpI2"), got, ref, want):
        assert np.array_equal(g, r), name + ": uint8 path differs from the fp64 path"
        assert np.abs(g - w).max() <= 1e-9, name
        print("u8 %s L%d %-7s %s" % (res, levels, name, "(bit-exact vs oracle)" if np.array_equal(g, w) else ""))


def test_u8_every_sample_value(gpu):
    """All 256 sample values go through the device-side division: warp of identical frames returns the frame."""
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    im = np.ascontiguousarray(np.stack([ramp, ramp.T, ramp[::-1]], axis=2))
    vx, vy, wi, _ = gpu.coarse2fine_flow_u8(im, im, 1)
    assert not vx.any() and not vy.any()
    assert np.array_equal(wi, _f64(im))


@pytest.mark.parametrize("res,levels,as_u8", [("240", 5, False), ("240", 3, True), ("480", 5, True)])
def test_sequence_equals_independent_pairs(gpu, res, levels, as_u8):
    frames = _frames(res, 3)
    pairs = [gpu.coarse2fine_flow(_f64(frames[i]), _f64(frames[i + 1]), levels)[:3] for i in range(2)]
    gpu.seq_reset()
    feed = frames if as_u8 else [_f64(f) for f in frames]
    assert gpu.seq_push(feed[0], levels) is None  # primes the sequence
    for i in (1, 2):
        out = gpu.seq_push(feed[i], levels)
        assert out is not None
        for name, g, w in zip(("vx", "vy", "warpI2"), out[:3], pairs[i - 1]):
            assert np.array_equal(g, w), "pair %d %s" % (i, name)
        assert out[3][9] > 0
    # a frame of another shape starts a new sequence instead of failing; so does an intervening ordinary call
    small = np.ascontiguousarray(feed[0][:64, :80])
    assert gpu.seq_push(small, 2) is None
    assert gpu.seq_push(np.ascontiguousarray(feed[1][:64, :80]), 2) is not None
    gpu.coarse2fine_flow(_f64(frames[0])[:32, :32].copy(), _f64(frames[1])[:32, :32].copy(), 1)
    assert gpu.seq_push(np.ascontiguousarray(feed[2][:64, :80]), 2) is None


def test_sequence_other_levels_or_params_restart_or_continue(gpu):
    from papteam_opticalflow_amd import default_params
    f = [_f64(x) for x in _frames("240", 3)]
    gpu.seq_reset()
    assert gpu.seq_push(f[0], 3) is None
    assert gpu.seq_push(f[1], 4) is None          # another pyramid plan: new sequence primed with f[1]
    p = default_params(n_outer=2, n_outer_per_level=0, n_sor=5, n_sor_per_level=0)
    out = gpu.seq_push(f[2], 4, p)                # solver parameters may change between pushes
    want = gpu.coarse2fine_flow(f[1], f[2], 4, p)
    assert out is not None
    for g, w in zip(out[:3], want[:3]):
        assert np.array_equal(g, w)


def test_package_level_sequence_and_u8_api():
    from papteam_opticalflow_amd import FlowSequence, coarse2fine_flow, coarse2fine_flow_u8
    frames = _frames("240", 3)
    seq = FlowSequence(2)
    assert seq.push(frames[0]) is None
    t, vx, vy, w = seq.push(frames[1])
    t2, vx2, vy2, w2 = coarse2fine_flow_u8(frames[0], frames[1], 2)
    t3, vx3, vy3, w3 = coarse2fine_flow(_f64(frames[0]), _f64(frames[1]), 2)
    assert list(t) == list(t2) == list(t3) and all(isinstance(x, str) for x in t.values())
    assert np.array_equal(vx, vx2) and np.array_equal(vx, vx3) and np.array_equal(w, w3) and np.array_equal(vy, vy2)
    seq.close()


def test_collection_in_flight_equals_pairwise(gpu):
    """flow_collection: several sequences in flight on one GPU (threads, one handle each) give, pair by pair, the bits
    of the pairwise call."""
    from papteam_opticalflow_amd import flow_collection
    frames = _frames("240", 3)
    frames = [frames[0], frames[1], frames[2], frames[1], frames[0], frames[2]]  # 5 pairs out of the 3 fixtures
    got = flow_collection(frames, 3, in_flight=3)
    assert len(got) == 5 and all(g is not None for g in got)
    for i, (t, vx, vy, w) in enumerate(got):
        want = gpu.coarse2fine_flow(_f64(frames[i]), _f64(frames[i + 1]), 3)
        assert np.array_equal(vx, want[0]) and np.array_equal(vy, want[1]) and np.array_equal(w, want[2]), i
        assert isinstance(t["Total C++ Execution"], str)
    kept = {}
    assert flow_collection(frames, 3, in_flight=2, on_pair=lambda i, t_, vx_, vy_, w_: kept.__setitem__(
        i, (vx_.copy(), vy_.copy(), w_.copy()))) is None  # callback form: arrays are reused, so copy what is kept
    assert sorted(kept) == [0, 1, 2, 3, 4]
    for i in range(5):
        assert all(np.array_equal(x, y) for x, y in zip(kept[i], got[i][1:]))
    assert flow_collection(frames[:1], 3) == []
    one = flow_collection(frames[:2], 3, in_flight=8)
    assert len(one) == 1 and np.array_equal(one[0][1], got[0][1])


def test_collection_default_in_flight_and_one_stream_per_handle(gpu):
    """flow_collection() without `in_flight` picks the number of sequences by frame size (16 for the small frames of the
    reference's own test matrix) and runs every handle on ONE stream (papof_set_stream_overlap(h, 0): with several handles in
    flight the extra streams only share hardware queues).  Same bits as the pairwise call; the switch itself gives the same
    bits on a single handle too."""
    from papteam_opticalflow_amd import Papof, collection_in_flight, flow_collection
    assert collection_in_flight(135, 240) == 16 and collection_in_flight(270, 480) == 16
    assert collection_in_flight(540, 960) == 8 and collection_in_flight(1080, 1920) == 4
    frames = _frames("240", 3)
    video = [frames[i % 3] for i in range(20)]  # 19 pairs: 16 segments of one or two pairs
    got = flow_collection(video, 3)
    assert len(got) == 19
    want = {}
    for i, (t, vx, vy, w) in enumerate(got):
        key = (i % 3, (i + 1) % 3)
        if key not in want:
            want[key] = gpu.coarse2fine_flow(_f64(video[i]), _f64(video[i + 1]), 3)
        assert np.array_equal(vx, want[key][0]) and np.array_equal(vy, want[key][1]) and np.array_equal(w, want[key][2]), i
    g = Papof(0)
    try:
        g.set_stream_overlap(False)
        one = g.coarse2fine_flow(_f64(video[0]), _f64(video[1]), 3)
        g.set_stream_overlap(True)
        two = g.coarse2fine_flow(_f64(video[0]), _f64(video[1]), 3)
    finally:
        g.close()
    for x, y, z in zip(one[:3], two[:3], want[(0, 1)][:3]):
        assert np.array_equal(x, z) and np.array_equal(y, z)


def test_graph_mode_replays_the_same_bits(gpu, oracle):
    """papof_set_graph_mode: the second call with the same arguments is captured into a hipGraph (both streams, every
    kernel / memset / copy), later calls replay it with one launch.  Eager, capturing and replaying calls must all
    return the oracle's bits; sequences (two graphs, one per pyramid-slot parity) and other shapes in between too."""
    from papteam_opticalflow_amd import Papof, default_params
    frames = _frames("240", 3)
    f = [_f64(x) for x in frames]
    want01 = oracle.coarse2fine_flow(f[0], f[1], 3)[:3]
    g = Papof(0)
    g.set_graph_mode(True)
    try:
        for call in range(4):  # eager, capture, replay, replay
            got = g.coarse2fine_flow(f[0], f[1], 3)
            for x, y in zip(got[:3], want01):
                assert np.array_equal(x, y), call
            assert got[3][9] > 0
        P = default_params(n_outer=2, n_outer_per_level=0, n_sor=5, n_sor_per_level=0, sor_mode=1)
        for call in range(3):  # other parameters and sweep order: their own graph
            got = g.coarse2fine_flow_u8(frames[1], frames[2], 4, P)
            want = gpu.coarse2fine_flow(f[1], f[2], 4, P)
            assert all(np.array_equal(x, y) for x, y in zip(got[:3], want[:3])), call
        pair = [gpu.coarse2fine_flow(f[i], f[i + 1], 3)[:3] for i in range(2)]
        g.seq_reset()
        feed = [frames[0], frames[1], frames[2]] * 4
        expect = [None, pair[0], pair[1], gpu.coarse2fine_flow(f[2], f[0], 3)[:3]]
        for i, fr in enumerate(feed):
            out = g.seq_push(fr, 3)
            if i == 0:
                assert out is None
                continue
            w_ = expect[1 + (i - 1) % 3]
            assert all(np.array_equal(x, y) for x, y in zip(out[:3], w_)), i
        got = g.coarse2fine_flow(f[0], f[1], 3)  # the pair graph from the beginning is still valid
        assert all(np.array_equal(x, y) for x, y in zip(got[:3], want01))
    finally:
        g.close()


def test_pinned_result_arrays_are_recycled_and_hold_the_same_results(monkeypatch):
    """The Python bindings take (vx, vy, warpI2) from a pool of page-locked memory (papof_host_alloc) and get the memory
    back when the caller drops the arrays: same results as with plain np.zeros arrays, ordinary writable numpy arrays,
    and the second call reuses the first call's memory once it has been dropped."""
    import gc
    import sys
    import os
    from papteam_opticalflow_amd import Papof, capi
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "papteam_opticalflow_amd", "dropin"))
    import pyflow
    rng = np.random.default_rng(5)
    a = np.ascontiguousarray(rng.uniform(0, 1, (300, 500, 3)))  # 1.2 MB per flow plane: above the pool's 1-MiB floor
    b = np.ascontiguousarray(np.clip(np.roll(a, (1, 2), (0, 1)) + rng.normal(0, 0.01, a.shape), 0, 1))
    g = Papof(0)
    try:
        monkeypatch.setenv("PAPOF_PINNED_OUT", "0")
        plain = g.coarse2fine_flow(a, b, 3)[:3]
        monkeypatch.setenv("PAPOF_PINNED_OUT", "1")
        vx, vy, wi, _ = g.coarse2fine_flow(a, b, 3)
        assert all(np.array_equal(x, y) for x, y in zip((vx, vy, wi), plain))
        assert vx.flags.writeable and vx.dtype == np.float64 and vx.flags.c_contiguous
        vx += 1.0  # an ordinary array
        addr = vx.ctypes.data
        gc.collect()  # whatever earlier tests left behind goes now, not between the two readings below
        pool = capi.pinned_pool()
        live = pool.live_bytes
        assert live >= vx.nbytes + vy.nbytes + wi.nbytes
        view = vx[10:20]  # a view keeps the block alive
        del vx
        gc.collect()
        assert pool.live_bytes == live and view[0, 0] == plain[0][10, 0] + 1.0
        del view, vy, wi
        gc.collect()
        from papteam_opticalflow_amd.pinned_pool import size_class
        assert pool.live_bytes == live - 2 * size_class(300 * 500 * 8) - size_class(300 * 500 * 8 * 3)
        assert pool.pinned_bytes <= pool.budget  # the budget counts the idle blocks too
        again = g.coarse2fine_flow(a, b, 3)
        assert addr in (again[0].ctypes.data, again[1].ctypes.data)  # recycled
        assert np.array_equal(again[0], plain[0])
        t, u, v, w2 = pyflow.coarse2fine_flow(a, b, 3)  # the Cython drop-in has its own pool of the same kind
        assert np.array_equal(u, plain[0]) and np.array_equal(w2, plain[2]) and u.flags.writeable
        del u, v, w2
        gc.collect()
        t, u, v, w2 = pyflow.coarse2fine_flow(a, b, 3)
        assert np.array_equal(v, plain[1])
    finally:
        g.close()
