"""Pins the CPU oracle (oracle/papof_oracle.c) against the golden vectors produced by the untouched
reference (tests/golden/make_golden.py).  Bar: bit-for-bit (SHA-256 of the float64 bytes)."""
import json
import os

import numpy as np
import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "golden.json")))
_NPZ = None


def golden_sub(case, name):
    global _NPZ
    if _NPZ is None:
        _NPZ = np.load(os.path.join(GOLD, "golden.npz"))
    return _NPZ["%s|%s" % (case, name)]


def test_frames_decode_as_when_goldens_were_made():
    for key, meta in MANIFEST["frames"].items():
        res, idx = key.split("/")
        px = cases.load_frame_u8(res, int(idx))
        assert list(px.shape) == meta["shape"]
        assert cases.sha(px.astype(np.float64)) == meta["sha"], "JPEG decoder differs from the golden run: " + key


def _names():
    slow = os.environ.get("PAPOF_SLOW") == "1"
    return [n for n in cases.CASES if slow or n not in cases.SLOW]


@pytest.mark.parametrize("case", _names())
def test_oracle_matches_reference_bit_for_bit(oracle, case):
    out = cases.CASES[case](oracle)
    gold = MANIFEST["cases"][case]
    assert set(out) == set(gold)
    for name, a in out.items():
        assert list(a.shape) == gold[name]["shape"], (case, name)
        if cases.sha(a) != gold[name]["sha"]:
            d = np.abs(cases.subsample(a) - golden_sub(case, name)).max()
            pytest.fail("%s/%s differs from the reference: max-abs on subsample %.3e" % (case, name, d))


def test_laplacian_known_answer(oracle):
    """operator matrix printed by the reference's OpticalFlow::testLaplacian(3)
    (src/OpticalFlow.cpp:693-723) for unit weights; asymmetric last column / last row."""
    expect = np.array([[2, -1, 0, -1, 0, 0, 0, 0, 0], [-1, 3, -1, 0, -1, 0, 0, 0, 0], [0, 0, 1, 0, 0, -1, 0, 0, 0],
                       [-1, 0, 0, 3, -1, 0, -1, 0, 0], [0, -1, 0, -1, 4, -1, 0, -1, 0], [0, 0, -1, 0, 0, 2, 0, 0, -1],
                       [0, 0, 0, 0, 0, 0, 1, -1, 0], [0, 0, 0, 0, 0, 0, -1, 2, -1], [0, 0, 0, 0, 0, 0, 0, 0, 0]],
                      dtype=np.float64)
    m = np.zeros((9, 9))
    for i in range(9):
        u = np.zeros(9)
        u[i] = 1
        m[:, i] = oracle.laplacian(u.reshape(3, 3), np.ones((3, 3))).ravel()
    assert np.array_equal(m, expect)


def test_pyramid_dims_known_answer(oracle):
    """level dims measured on the reference with ConstructPyramidLevels(ratio .75, 5 levels)."""
    table = {(1080, 1920): [(1920, 1080), (1440, 810), (1080, 607), (810, 455), (607, 341)],
             (270, 480): [(480, 270), (360, 202), (270, 151), (202, 113), (151, 85)],
             (135, 240): [(240, 135), (180, 101), (135, 75), (101, 56), (75, 42)]}
    import ctypes
    for (h, w), dims in table.items():
        d = np.zeros(10, dtype=np.int32)
        oracle.L.orc_pyramid(None, h, w, 3, 0.75, 5, d.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), None)
        assert [(int(d[2 * i]), int(d[2 * i + 1])) for i in range(5)] == dims


def test_sor_modes_are_distinct_and_deterministic(oracle):
    """Red-black / Jacobi are throughput modes: they must NOT be mistaken for reference parity."""
    rng = np.random.default_rng(2)
    h, w = 33, 47
    phi = rng.uniform(0.5, 50.0, (h, w))
    imdxy = rng.uniform(-0.02, 0.02, (h, w))
    imdx2 = rng.uniform(0, 0.05, (h, w))
    imdy2 = rng.uniform(0, 0.05, (h, w))
    r1 = rng.uniform(-0.01, 0.01, (h, w))
    r2 = rng.uniform(-0.01, 0.01, (h, w))
    ex = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, 30, mode=0)
    rb = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, 30, mode=1)
    ja = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, 30, omega=1.0, mode=2)
    ex2 = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, 30, mode=0)
    assert np.array_equal(ex[0], ex2[0]) and np.array_equal(ex[1], ex2[1])
    assert np.abs(ex[0] - rb[0]).max() > 1e-6
    assert np.all(np.isfinite(ja[0])) and np.all(np.isfinite(rb[0]))
    # all three converge to the same fixed point when run long enough
    ex_l = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, 3000, mode=0)
    rb_l = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, 3000, mode=1)
    assert np.abs(ex_l[0] - rb_l[0]).max() < 1e-9


def test_recomposed_level_loop_equals_the_reference_entry_point():
    """The cfg4_*, ratio* and inner2 goldens come from `ref_coarse2fine_flow_sched` (oracle/ref_driver.cpp), a
    re-composition of the reference's level loop out of its public statics -- the only way to run the untouched
    reference with another schedule.  This pins that re-composition to the reference's own entry point: with the
    reference's hard-coded schedule (outer 7 + k, inner 1, sweeps 30 + 3k, src/OpticalFlow.cpp:747-751, :823) it must
    return `OpticalFlow::Coarse2FineFlow`'s bits.  Container only: needs the compiled reference (oracle/_ref)."""
    from _libs import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/libpapof_ref.so not built (the reference exists in the build container only)")
    ref = RefLib()
    a, b = cases.load_pair("240")
    for levels in (1, 3, 5):
        vx, vy, wi, _ = ref.coarse2fine_flow(a, b, levels)
        sx, sy, sw = ref.coarse2fine_flow_sched(a, b, levels, 0.012, 0.75, 7, 1, 1, 30, 3)
        assert np.array_equal(vx, sx) and np.array_equal(vy, sy) and np.array_equal(wi, sw), levels
