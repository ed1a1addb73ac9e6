"""Parity tests proper (-m gpu): every HIP stage and the whole call, through the C ABI (libpapof.so), against the
CPU oracle on identical inputs.  The oracle itself is pinned bit-for-bit to the untouched reference
(tests/test_oracle_golden.py).

Tolerances.  BASELINE.json's bar is max-abs 1e-4 on (u, v).  The kernels are written to reproduce the reference's
fp64 operation order without FMA contraction, so the tests demand far more: TOL = 1e-12 for single stages and
1e-9 for whole solves (chaotic amplification of a last-bit difference through 45+ outer iterations stays orders
of magnitude below that); bit-for-bit equality is reported when it holds.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

TOL_STAGE = 1e-12
TOL_SOLVE = 1e-9
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    yield g
    g.close()


def _cmp(name, got, want, tol):
    assert got.shape == want.shape, name
    assert np.all(np.isfinite(got)), name + ": non-finite values"
    d = float(np.abs(got - want).max()) if got.size else 0.0
    print("%-28s max-abs %.3e %s" % (name, d, "(bit-exact)" if np.array_equal(got, want) else ""))
    assert d <= tol, "%s: max-abs %.3e > %.1e" % (name, d, tol)


STAGES = ["stage_pyramid", "stage_gaussian", "stage_resize", "stage_im2feature", "stage_warp", "stage_getdxs",
          "stage_laplacian", "stage_smoothflow", "stage_flow16"]  # stage_smoothflow includes nInner = 2 (src/OpticalFlow.cpp:300-304)


@pytest.mark.parametrize("case", STAGES)
def test_stage_matches_oracle(gpu, oracle, case):
    got = cases.CASES[case](gpu)
    want = cases.CASES[case](oracle)
    assert set(got) == set(want)
    for k in want:
        _cmp(case + "/" + k, got[k], want[k], TOL_STAGE)


# The Gaussian-mixture noise model (src/OpticalFlow.cpp:359-367, :539-591) is the one branch that cannot be bit-compatible:
# it evaluates exp() (the device library's, <= 1 ulp, not glibc's bits) and sums over all pixels (the reference adds
# them sequentially, a GPU reduction adds them as a tree).  Policy: the mixture parameters after an EM
# estimate agree to 1e-12 relative and the flow after one SmoothFlowSOR call of 3 outer iterations to 1e-9 (measured
# 1e-13 / 3e-13).  Whole calls: see test_gaussian_mixture_end_to_end_... -- the mixture model makes the iteration chaotic.
TOL_GM_PARA = 1e-12
TOL_GM_SOLVE = 1e-9


def test_min_width_pyramid_matches_oracle(gpu, oracle):
    """GaussianPyramid::ConstructPyramid(image, ratio, minWidth) (src/GaussianPyramid.cpp:47-77): the level count formula,
    then the same levels -- bit for bit against the oracle (itself pinned to the reference's golden vectors)."""
    got = cases.CASES["stage_pyramid_minwidth"](gpu)
    want = cases.CASES["stage_pyramid_minwidth"](oracle)
    assert set(got) == set(want) and got["n"][0] == 7 and got["r05_n"][0] == 2
    for k in want:
        _cmp("stage_pyramid_minwidth/" + k, got[k], want[k], TOL_STAGE)
    a, _ = cases.load_pair("240")
    import ctypes
    from papteam_opticalflow_amd import capi
    oracle.L.orc_pyramid_levels_for_min_width.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int]
    for width, ratio, mw in ((240, 0.75, 30), (1920, 0.75, 16), (240, 0.5, 16), (240, 0.9, 100), (240, 0.2, 30),
                             (240, 0.75, 239), (960, 0.98, 20), (50, 0.4, 3)):
        n = ctypes.c_int(0)
        assert capi.load().papof_pyramid_levels_for_min_width(width, ratio, mw, ctypes.byref(n)) == 0
        assert n.value == oracle.L.orc_pyramid_levels_for_min_width(width, ratio, mw), (width, ratio, mw)


def test_branch_stages_match_oracle(gpu, oracle):
    """In-loop bicubic warping (bit-exact) and the Gaussian-mixture model (tolerance, see above) at the level of one
    SmoothFlowSOR call."""
    got = cases.CASES["stage_branches"](gpu)
    want = cases.CASES["stage_branches"](oracle)
    assert set(got) == set(want)
    for k in ("bicubic_noclamp", "bc_warp", "bc_u", "bc_v"):
        _cmp("stage_branches/" + k, got[k], want[k], TOL_STAGE)
    for k in ("gm_est", "gm_para"):
        rel = float(np.abs(got[k] / want[k] - 1).max())
        print("%-28s max relative %.3e" % ("stage_branches/" + k, rel))
        assert rel <= TOL_GM_PARA, (k, rel)
    for k in ("gm_warp", "gm_u", "gm_v"):
        _cmp("stage_branches/" + k, got[k], want[k], TOL_GM_SOLVE)


def test_bicubic_interpolation_end_to_end_matches_reference_golden(gpu):
    """interpolation = Bicubic, whole call, against what the untouched reference produced with its static set that way
    (oracle/ref_driver.cpp::ref_coarse2fine_flow_opts): bit-compatible like the default path."""
    gold = np.load(os.path.join(GOLD, "golden.npz"))
    got = cases.CASES["bicubic_240_L3"](gpu)
    for k, a in got.items():
        _cmp("bicubic_240_L3/" + k + " vs reference", cases.subsample(a), gold["bicubic_240_L3|" + k], TOL_SOLVE)


@pytest.mark.parametrize("res,levels,interpolation", [("240", 3, 0), ("480", 4, 1)])
def test_gaussian_mixture_end_to_end_within_the_algorithms_own_conditioning(gpu, oracle, res, levels, interpolation):
    """noise model = Gaussian mixture, whole calls.  With this model the reference's iteration is CHAOTIC: a 1e-13 relative
    perturbation grows ~7x per outer iteration in the reference's own arithmetic (the CPU oracle, pinned bit for bit to
    the reference, run on inputs scaled by 1 + 1e-13, ends 8e-10 / 6e-4 / 10 pixels away from itself after 1 / 2 / 3
    levels of the reference schedule at 240x135; with the default Laplacian model: 1e-13).  Device exp() and parallel
    sums are such a perturbation, so no implementation can meet an absolute bound here; what is asserted:
      * with 2 outer iterations per level (6-8 in all: amplification <= ~7^8) the GPU matches the oracle to 1e-7
        (measured 2e-11 at 240x135 / 3 levels, 2e-9 at 480x270 / 4 levels);
      * at the reference schedule the GPU is no further from the oracle than 10x the oracle's own self-divergence
        under the 1e-13 scaling (both printed), and the golden distance is reported."""
    from papteam_opticalflow_amd import default_params
    a, b = cases.load_pair(res)
    short = dict(n_outer=2, n_outer_per_level=0)
    got = gpu.coarse2fine_flow(a, b, levels, default_params(noise_model=1, interpolation=interpolation, **short))[:3]
    p = oracle.default_params()
    p.noise_model, p.interpolation, p.n_outer, p.n_outer_per_level = 1, interpolation, 2, 0
    want = oracle.coarse2fine_flow(a, b, levels, p)[:3]
    for name, g, w_ in zip(("vx", "vy", "warpI2"), got, want):
        _cmp("gmixture %s L%d, 2 outer iterations %s" % (res, levels, name), g, w_, 1e-7)
    got = gpu.coarse2fine_flow(a, b, levels, default_params(noise_model=1, interpolation=interpolation))[:2]
    p = oracle.default_params()
    p.noise_model, p.interpolation = 1, interpolation
    want = oracle.coarse2fine_flow(a, b, levels, p)[:2]
    pert = oracle.coarse2fine_flow(a * (1 + 1e-13), b * (1 + 1e-13), levels, p)[:2]
    d_gpu = max(np.abs(g - w_).max() for g, w_ in zip(got, want))
    d_self = max(np.abs(q - w_).max() for q, w_ in zip(pert, want))
    print("gmixture %s L%d reference schedule: GPU vs oracle %.3e, oracle vs itself on (1 + 1e-13)-scaled inputs %.3e"
          % (res, levels, d_gpu, d_self))
    assert np.all(np.isfinite(got[0])) and np.all(np.isfinite(got[1]))
    assert d_gpu <= 10 * d_self + TOL_SOLVE, (d_gpu, d_self)


def test_laplacian_known_answer_on_gpu(gpu):
    """The reference's testLaplacian(3) matrix reproduced by the HIP kernel."""
    expect = np.array([[2, -1, 0, -1, 0, 0, 0, 0, 0], [-1, 3, -1, 0, -1, 0, 0, 0, 0], [0, 0, 1, 0, 0, -1, 0, 0, 0],
                       [-1, 0, 0, 3, -1, 0, -1, 0, 0], [0, -1, 0, -1, 4, -1, 0, -1, 0], [0, 0, -1, 0, 0, 2, 0, 0, -1],
                       [0, 0, 0, 0, 0, 0, 1, -1, 0], [0, 0, 0, 0, 0, 0, -1, 2, -1], [0, 0, 0, 0, 0, 0, 0, 0, 0]],
                      dtype=np.float64)
    m = np.zeros((9, 9))
    for i in range(9):
        u = np.zeros(9)
        u[i] = 1
        m[:, i] = gpu.laplacian(u.reshape(3, 3), np.ones((3, 3))).ravel()
    assert np.array_equal(m, expect)


@pytest.mark.parametrize("shape", [(135, 240), (68, 120), (3, 4), (1, 9), (9, 1)])
def test_linear_system_matches_oracle(gpu, oracle, shape):
    a, b = cases.load_pair("240")
    h, w = shape
    f1 = np.ascontiguousarray(cases.features5(a)[:h, :w])
    f2 = np.ascontiguousarray(cases.features5(b)[:h, :w])
    rng = np.random.default_rng(5)
    u = cases.smooth_field(rng, 135, 240, 1.5)[:h, :w].copy()
    v = cases.smooth_field(rng, 135, 240, 1.5)[:h, :w].copy()
    got = gpu.linear_system(f1, f2, u, v)
    dx, dy, dt = oracle.getDxs(f1, f2)
    want = oracle.linear_system(dx, dy, dt, u, v)
    for name, g, w_ in zip(("phi", "imdxy", "imdx2", "imdy2", "rhs1", "rhs2"), got, want):
        _cmp("linear_system%s/%s" % (shape, name), g, w_, TOL_STAGE)


def _sor_planes(h, w, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0.5, 50.0, (h, w)), rng.uniform(-0.02, 0.02, (h, w)), rng.uniform(0, 0.05, (h, w)),
            rng.uniform(0, 0.05, (h, w)), rng.uniform(-0.01, 0.01, (h, w)), rng.uniform(-0.01, 0.01, (h, w)))


@pytest.mark.parametrize("h,w,n_sor", [(70, 50, 4), (130, 37, 3), (64, 20, 3), (1, 5, 3), (5, 1, 3), (129, 3, 2),
                                        (42, 75, 5), (341, 607, 42), (540, 960, 30), (200, 1000, 7), (1000, 17, 9)])
def test_sor_exact_order_matches_oracle(gpu, oracle, h, w, n_sor):
    """The hyperplane-scheduled kernel must reproduce the reference's in-place lexicographic sweeps
    (src/OpticalFlow.cpp:458-505); any stale cross-workgroup read would show up here."""
    planes = _sor_planes(h, w, h * 7 + w)
    du, dv = gpu.sor(*planes, n_sor, mode=0)
    eu, ev = oracle.sor(*planes, n_sor, mode=0)
    _cmp("sor_exact %dx%d k=%d du" % (h, w, n_sor), du, eu, TOL_STAGE)
    _cmp("sor_exact %dx%d k=%d dv" % (h, w, n_sor), dv, ev, TOL_STAGE)


def test_sor_exact_is_deterministic_at_full_size(gpu):
    planes = _sor_planes(1080, 1920, 99)
    a = gpu.sor(*planes, 30, mode=0)
    b = gpu.sor(*planes, 30, mode=0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.all(np.isfinite(a[0]))


@pytest.mark.parametrize("mode,omega", [(1, 1.8), (2, 1.0)])
@pytest.mark.parametrize("h,w,n_sor", [(70, 51, 5), (135, 240, 30), (1, 7, 3), (6, 1, 4)])
def test_sor_throughput_modes_match_oracle_in_same_mode(gpu, oracle, mode, omega, h, w, n_sor):
    """Red-black / Jacobi are checked against the oracle run in the SAME mode; they are not reference parity
     and the test says so by also printing their distance to the exact order."""
    planes = _sor_planes(h, w, 3 * h + w)
    du, dv = gpu.sor(*planes, n_sor, omega=omega, mode=mode)
    eu, ev = oracle.sor(*planes, n_sor, omega=omega, mode=mode)
    _cmp("sor mode%d %dx%d du" % (mode, h, w), du, eu, TOL_STAGE)
    _cmp("sor mode%d %dx%d dv" % (mode, h, w), dv, ev, TOL_STAGE)
    xu, _ = oracle.sor(*planes, n_sor, omega=omega, mode=0)
    print("   distance to the exact (reference) order: %.3e" % np.abs(du - xu).max())


@pytest.mark.parametrize("mode,omega", [(1, 1.8), (2, 1.0)])
def test_inner_iterations_in_throughput_modes(gpu, oracle, mode, omega):
    a, b = cases.load_pair("240")
    f1 = np.ascontiguousarray(cases.features5(a)[::2, ::2])
    f2 = np.ascontiguousarray(cases.features5(b)[::2, ::2])
    h, w, _ = f1.shape
    z = np.zeros((h, w))
    got = gpu.smoothflow_sor(f1, f2, f2, z, z, 0.012, 2, 3, 7, omega=omega, mode=mode)
    want = oracle.smoothflow_sor(f1, f2, f2, z, z, 0.012, 2, 3, 7, omega=omega, mode=mode)
    for name, g, w_ in zip(("warp", "u", "v"), got, want):
        _cmp("inner3 mode%d/%s" % (mode, name), g, w_, TOL_SOLVE)


@pytest.mark.parametrize("mode,omega", [(0, 1.8), (1, 1.8), (2, 1.0)])
def test_smoothflow_level_matches_oracle(gpu, oracle, mode, omega):
    a, b = cases.load_pair("240")
    f1 = np.ascontiguousarray(cases.features5(a)[::2, ::2])
    f2 = np.ascontiguousarray(cases.features5(b)[::2, ::2])
    h, w, _ = f1.shape
    z = np.zeros((h, w))
    got = gpu.smoothflow_sor(f1, f2, f2, z, z, 0.012, 3, 1, 10, omega=omega, mode=mode)
    want = oracle.smoothflow_sor(f1, f2, f2, z, z, 0.012, 3, 1, 10, omega=omega, mode=mode)
    for name, g, w_ in zip(("warp", "u", "v"), got, want):
        _cmp("smoothflow mode%d/%s" % (mode, name), g, w_, TOL_SOLVE)


E2E = [("240", 1), ("240", 2), ("240", 5), ("240", 15), ("480", 5), ("960", 5)]


@pytest.mark.parametrize("res,levels", E2E)
def test_end_to_end_matches_oracle(gpu, oracle, res, levels):
    a, b = cases.load_pair(res)
    vx, vy, wi, t = gpu.coarse2fine_flow(a, b, levels)
    ox, oy, ow, _ = oracle.coarse2fine_flow(a, b, levels)
    _cmp("e2e %s L%d vx" % (res, levels), vx, ox, TOL_SOLVE)
    _cmp("e2e %s L%d vy" % (res, levels), vy, oy, TOL_SOLVE)
    _cmp("e2e %s L%d warpI2" % (res, levels), wi, ow, TOL_SOLVE)
    assert np.all(np.asarray(t) > 0), t  # all ten reference timers are measured on the default call


@pytest.mark.parametrize("case", ["e2e_1920_L5", "cfg4_1920_L5", "e2e_960_L5", "cfg4_480_L5", "gray_240_L3",
                                  "ratio05_240_L3", "ratio09_240_L4", "inner2_240_L3"])
def test_end_to_end_matches_reference_golden(gpu, case):
    """Directly against the values the untouched reference produced (strided subsample in golden.npz),
    including the full-size 1920x1080 configurations of BASELINE.json that the oracle would need a minute for."""
    gold = np.load(os.path.join(GOLD, "golden.npz"))
    man = json.load(open(os.path.join(GOLD, "golden.json")))["cases"][case]
    got = cases.CASES[case](gpu)
    for k, a in got.items():
        _cmp(case + "/" + k + " vs reference", cases.subsample(a), gold["%s|%s" % (case, k)], TOL_SOLVE)
        # ... and EVERY value: the SHA-256 of the full float64 array the untouched reference produced (the subsample is 1 % of a
        # 1080p plane and cannot see a defect confined to one band seam or column)
        assert cases.sha(a) == man[k]["sha"], "%s/%s: full-array SHA-256 differs from the reference's" % (case, k)


@pytest.mark.parametrize("h,w,levels,kw", [
    (37, 53, 3, {}),                                            # ragged, smaller than one band
    (100, 7, 2, {}),                                            # tall and thin
    (7, 100, 2, {}),                                            # one-band strip
    (63, 64, 2, dict(n_sor=1, n_sor_per_level=0)),              # single sweep, band boundary at row 62
    (125, 66, 3, dict(n_sor=61, n_sor_per_level=2)),            # more sweeps than rows per band (bands climb > 62 rows)
    (90, 120, 4, dict(alpha=0.05, omega=1.3, n_outer=2, n_outer_per_level=2, n_sor=7, n_sor_per_level=5)),
    (90, 120, 3, dict(ratio=0.5)),                              # other pyramid ratio (different Gaussian half-widths)
    (90, 120, 3, dict(ratio=0.9)),
    (90, 120, 3, dict(ratio=0.2)),                              # out of range -> clamped to 0.75 like the reference
])
def test_ragged_sizes_and_parameters_match_oracle(gpu, oracle, h, w, levels, kw):
    from papteam_opticalflow_amd import default_params
    a, b = cases.load_pair("240")
    a = np.ascontiguousarray(a[:h, :w])
    b = np.ascontiguousarray(b[:h, :w])
    got = gpu.coarse2fine_flow(a, b, levels, default_params(**kw))[:3]
    p = oracle.default_params()
    for k, v in kw.items():
        setattr(p, k, v)
    want = oracle.coarse2fine_flow(a, b, levels, p)[:3]
    for name, g, w_ in zip(("vx", "vy", "warpI2"), got, want):
        _cmp("ragged %dx%d L%d %s %s" % (h, w, levels, kw, name), g, w_, TOL_SOLVE)


def test_config4_schedule_and_modes(gpu, oracle):
    a, b = cases.load_pair("240")
    for mode, omega in ((0, 1.8), (1, 1.8), (2, 1.0)):
        got = gpu.coarse2fine_flow_sched(a, b, 5, 0.012, 0.75, 3, 0, 1, 30, 0, mode=mode, omega=omega)
        want = oracle.coarse2fine_flow_sched(a, b, 5, 0.012, 0.75, 3, 0, 1, 30, 0, mode=mode, omega=omega)
        for name, g, w_ in zip(("vx", "vy", "warpI2"), got, want):
            _cmp("cfg4 mode%d %s" % (mode, name), g, w_, TOL_SOLVE)


def test_config2_480x270_jacobi_matches_oracle(gpu, oracle):
    """BASELINE.json configs[1]: 480x270 pair, Jacobi inner solver replacing SOR (correctness gate), at ITS size: the
    whole call (5 levels, reference schedule outer 7+k / sweeps 30+3k) against the oracle run in the same mode.
    omega = 1: plain Jacobi (an over-relaxed Jacobi iteration diverges)."""
    from papteam_opticalflow_amd import default_params
    a, b = cases.load_pair("480")
    got = gpu.coarse2fine_flow(a, b, 5, default_params(sor_mode=2, omega=1.0))[:3]
    p = oracle.default_params()
    p.sor_mode, p.omega = 2, 1.0
    want = oracle.coarse2fine_flow(a, b, 5, p)[:3]
    for name, g, w_ in zip(("vx", "vy", "warpI2"), got, want):
        _cmp("config2 480x270 jacobi %s" % name, g, w_, TOL_SOLVE)
    gold = np.load(os.path.join(GOLD, "golden.npz"))
    if "e2e_480_L5|vx" in gold.files:
        print("   distance of Jacobi (omega 1) to the reference order (subsample): %.3e" %
              np.abs(cases.subsample(got[0]) - gold["e2e_480_L5|vx"]).max())


def test_config3_960x540_redblack_matches_oracle(gpu, oracle):
    """BASELINE.json configs[2]: 960x540 pair, red-black SOR (LDS-tiled, temporally blocked kernel), full outer / inner
    fixed-point schedule of the reference (outer 7+k, inner 1, sweeps 30+3k, 5 levels), against the oracle in the same
    mode -- not against itself.  The distance to the reference's lexicographic order is printed, never asserted
    (red-black is a throughput mode, not reference parity)."""
    from papteam_opticalflow_amd import default_params
    a, b = cases.load_pair("960")
    got = gpu.coarse2fine_flow(a, b, 5, default_params(sor_mode=1))[:3]
    p = oracle.default_params()
    p.sor_mode = 1
    want = oracle.coarse2fine_flow(a, b, 5, p)[:3]
    for name, g, w_ in zip(("vx", "vy", "warpI2"), got, want):
        _cmp("config3 960x540 red-black %s" % name, g, w_, TOL_SOLVE)
    gold = np.load(os.path.join(GOLD, "golden.npz"))
    d = max(np.abs(cases.subsample(got[0]) - gold["e2e_960_L5|vx"]).max(),
            np.abs(cases.subsample(got[1]) - gold["e2e_960_L5|vy"]).max())
    print("   distance of red-black to the reference order (subsample): %.3e" % d)


def test_config4_1920x1080_redblack_matches_oracle(gpu, oracle):
    """BASELINE.json configs[3] at its full size with the kernel north_star describes (red-black SOR, LDS-staged halos):
    1920x1080 pair, 5 levels, 3 outer / 30 SOR, end to end against the oracle run in the same mode (~7 s of one host
    core).  The exact-order run of the same configuration is compared with the reference's golden in
    test_end_to_end_matches_reference_golden[cfg4_1920_L5]."""
    from papteam_opticalflow_amd import default_params
    a, b = cases.load_pair("1920")
    kw = dict(n_outer=3, n_outer_per_level=0, n_sor=30, n_sor_per_level=0)
    got = gpu.coarse2fine_flow(a, b, 5, default_params(sor_mode=1, **kw))[:3]
    p = oracle.default_params()
    for k, v in dict(kw, sor_mode=1).items():
        setattr(p, k, v)
    want = oracle.coarse2fine_flow(a, b, 5, p)[:3]
    for name, g, w_ in zip(("vx", "vy", "warpI2"), got, want):
        _cmp("config4 1920x1080 red-black %s" % name, g, w_, TOL_SOLVE)


@pytest.mark.parametrize("mode,omega", [(1, 1.8), (2, 1.0)])
@pytest.mark.parametrize("h,w,n_sor", [(1080, 1920, 30), (540, 960, 33), (341, 607, 42), (270, 480, 7), (129, 3, 2),
                                        (3, 300, 5), (49, 127, 11), (48, 128, 10), (97, 257, 1)])
def test_sor_blocked_kernels_match_oracle_at_size(gpu, oracle, mode, omega, h, w, n_sor):
    """The LDS-tiled, temporally blocked red-black / Jacobi kernel (sor.hip k_sor_blocked) at the sizes of every pyramid
    level it runs on, up to 1080p x 30 sweeps, bit for bit against the oracle in the same mode."""
    planes = _sor_planes(h, w, 5 * h + w)
    du, dv = gpu.sor(*planes, n_sor, omega=omega, mode=mode)
    eu, ev = oracle.sor(*planes, n_sor, omega=omega, mode=mode)
    assert np.array_equal(du, eu) and np.array_equal(dv, ev), (mode, h, w, n_sor, np.abs(du - eu).max())


@pytest.mark.parametrize("knobs", [{}, {"PAPOF_RB_SHAPE": "1"}, {"PAPOF_RB_SHAPE": "3", "PAPOF_RB_DEPTH": "5"},
                                   {"PAPOF_RB_SHAPE": "4", "PAPOF_RB_DEPTH": "7"}, {"PAPOF_RB_SHAPE": "5", "PAPOF_RB_DEPTH": "14"},
                                   {"PAPOF_RB_NAIVE": "1"}])
def test_sor_blocked_random_shapes_and_every_region_shape(oracle, knobs, monkeypatch):
    """Seeded random plane sizes (odd and even widths, planes smaller and larger than a region, 1 .. 12 sweeps) through
    every region shape / depth the blocked solver has (odd depths force the shifted, even-aligned regions), and through
    the one-launch-per-half-sweep kernels kept as a cross-check: all must give the oracle's bits in the same mode."""
    from papteam_opticalflow_amd import Papof
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g = Papof(0)
    try:
        rng = np.random.default_rng(2026)
        for _ in range(14):
            h, w, n_sor = int(rng.integers(1, 220)), int(rng.integers(1, 420)), int(rng.integers(1, 13))
            planes = _sor_planes(h, w, 17 * h + w)
            for mode, omega in ((1, 1.8), (2, 1.0)):
                du, dv = g.sor(*planes, n_sor, omega=omega, mode=mode)
                eu, ev = oracle.sor(*planes, n_sor, omega=omega, mode=mode)
                assert np.array_equal(du, eu) and np.array_equal(dv, ev), (knobs, mode, h, w, n_sor)
    finally:
        g.close()


def test_sor_exact_more_tasks_than_the_chip_keeps_resident(gpu, oracle):
    """Forward progress beyond co-residency: 1080 rows x 200 sweeps = 21 bands x 200 = 4200 one-wave tasks, more than the
    3072 waves the chip holds at the kernel's occupancy.  sor_solve issues such a solve as consecutive launches of at
    most 8 tasks per CU (ranges of sweeps); the result must still be the reference's bits."""
    planes = _sor_planes(1080, 64, 31)
    du, dv = gpu.sor(*planes, 200, mode=0)
    eu, ev = oracle.sor(*planes, 200, mode=0)
    assert np.array_equal(du, eu) and np.array_equal(dv, ev)


@pytest.mark.parametrize("res,levels,n_outer,n_sor,mode", [("1920", 1, 2, 9, 0), ("1920", 1, 2, 33, 0), ("1920", 5, 3, 30, 0),
                                                           ("960", 3, 2, 7, 0), ("480", 5, 2, 5, 0), ("1920", 3, 2, 30, 1),
                                                           ("960", 3, 2, 9, 2)])
def test_results_do_not_depend_on_other_kernels_on_the_chip(res, levels, n_outer, n_sor, mode):
    """Regression (round 2): the two-sweeps-per-wave kernel's last pair of an ODD sweep count (identity second sweep) used
    to return wrong cells when other kernels ran on the chip at the same time -- never alone, so every single-stream
    test passed.  Three handles in flight, each must reproduce the bits of the solo call: the odd
    counts that failed (1080p: 13 of 36 calls wrong at 33 sweeps), and the other solver kernels for good measure (config 4
    on five levels; odd counts on the plain kernel and on levels with XCD-affine task mapping; the blocked red-black and
    Jacobi kernels)."""
    import threading
    from papteam_opticalflow_amd import Papof
    a, b = cases.load_pair(res)
    hs = [Papof(0) for _ in range(3)]
    try:
        run = lambda g: g.coarse2fine_flow_sched(a, b, levels, 0.012, 0.75, n_outer, 0, 1, n_sor, 0, mode=mode,
                                                 omega=1.0 if mode == 2 else 1.8)[:3]
        want = run(hs[0])
        bad = [0, 0, 0]

        def work(i):
            for _ in range(8):
                bad[i] += not all(np.array_equal(x, y) for x, y in zip(run(hs[i]), want))
        th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert bad == [0, 0, 0], "wrong results per handle: %s of 8" % bad
    finally:
        for g in hs:
            g.close()


def test_sor_plan_reports_how_a_solve_is_issued(gpu):
    """papof_sor_plan (bench.py prices the roofline's per-launch figures with it): the exact-order kernels run a solve as
    ONE launch unless it has more tasks than the chip keeps resident; the blocked red-black kernel runs 10 half-sweeps
    per launch (whole sweeps per launch), a plane that fits one region in one launch; Jacobi counts sweeps."""
    assert gpu.sor_plan(1080, 1920, 30, 0) == (1, 2)      # level 0: two sweeps per wave (k_sor_fused)
    assert gpu.sor_plan(341, 607, 30, 0) == (1, 1)
    assert gpu.sor_plan(1080, 64, 200, 0)[0] > 1           # 4200 tasks > 8 per CU: consecutive launches
    assert gpu.sor_plan(1080, 1920, 30, 1) == (6, 10)
    assert gpu.sor_plan(1080, 1920, 33, 1) == (7, 10)      # 33 sweeps: 5 + 5 + 5 + 5 + 5 + 4 + 4
    assert gpu.sor_plan(42, 75, 42, 1) == (1, 84)          # one region: one launch, no ghost cells
    assert gpu.sor_plan(270, 480, 30, 2) == (5, 6)


def test_full_size_properties(gpu):
    """Size-independent properties at BASELINE.json's full 1920x1080 size."""
    a, b = cases.load_pair("1920")
    r1 = gpu.coarse2fine_flow(a, b, 5)
    r2 = gpu.coarse2fine_flow(a, b, 5)
    for x, y in zip(r1[:3], r2[:3]):  # determinism: the reference is bit-deterministic
        assert np.array_equal(x, y)
    vx, vy, wi, _ = gpu.coarse2fine_flow(a, a, 5)  # identical frames: zero flow, warp == frame
    assert not vx.any() and not vy.any()
    assert np.array_equal(wi, a)
    assert wi.min() >= 0.0 and wi.max() <= 1.0


def test_invalid_arguments_return_errors(gpu):
    from papteam_opticalflow_amd import PapofError, default_params
    a = np.zeros((8, 8, 3))
    with pytest.raises(PapofError):
        gpu.coarse2fine_flow(a, a, 0)
    with pytest.raises(PapofError):
        gpu.coarse2fine_flow(a, a, 2, default_params(n_inner=0))
    with pytest.raises(PapofError):
        gpu.coarse2fine_flow(a, a, 30)  # pyramid level smaller than one pixel
    with pytest.raises(ValueError):
        gpu.coarse2fine_flow(a, np.zeros((8, 9, 3)), 2)


def test_pyflow_dropin_entry_point(oracle):
    """The Cython module the reference's callers import (Code/Serial/OpticalFlowCalculation.py:73)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "papteam_opticalflow_amd", "dropin"))
    import pyflow
    a, b = cases.load_pair("240")
    timing, u, v, im2w = pyflow.coarse2fine_flow(a, b, 2)
    assert list(timing) == ["Allocation", "Construction", "Phase1_Generate", "Phase2_Derivatives", "Phase3_PsiData",
                            "Phase4_LinearSystem", "Phase5_SOR", "Phase6_Update", "PostProcessing",
                            "Total C++ Execution"]
    # the reference fills all ten (src/OpticalFlow.cpp:850-860) and its caller appends them to UniversalTiming.txt
    # (OpticalFlowCalculation.py:168-191): every one of them is measured on the DEFAULT call
    assert all(isinstance(x, str) and float(x) > 0 for x in timing.values()), timing
    assert float(timing["Total C++ Execution"]) >= max(float(v) for k, v in timing.items() if k != "Total C++ Execution")
    assert u.shape == (135, 240) and v.shape == (135, 240) and im2w.shape == (135, 240, 3)
    assert u.dtype == np.float64 and im2w.dtype == np.float64
    flow = np.concatenate((u[..., None], v[..., None]), axis=2)  # what the caller does next (:75)
    assert flow.shape == (135, 240, 2)
    ox, oy, ow, _ = oracle.coarse2fine_flow(a, b, 2)
    _cmp("pyflow vx", u, ox, TOL_SOLVE)
    _cmp("pyflow warpI2", im2w, ow, TOL_SOLVE)
    t2, u2, _, _ = pyflow.coarse2fine_flow(a, b, 2, 4)  # Parallel tree's 4th positional nCores
    assert np.array_equal(u, u2)
    t3 = pyflow.coarse2fine_flow(a, b, 2, phase_timing=1)[0]
    assert float(t3["Phase5_SOR"]) > 0
    with pytest.raises(ValueError):
        pyflow.coarse2fine_flow(a, b[:, :-1].copy(), 2)
    with pytest.raises(ValueError):
        pyflow.coarse2fine_flow(a[:, ::2], b[:, ::2], 2)  # non-contiguous
    with pytest.raises(TypeError):
        pyflow.coarse2fine_flow(None, b, 2)
    with pytest.raises(ValueError):
        pyflow.coarse2fine_flow(a.astype(np.float32), b.astype(np.float32), 2)


@pytest.mark.parametrize("knob,value", [("PAPOF_SOR_GROUP", "2"), ("PAPOF_SOR_GROUP", "4"), ("PAPOF_SOR_DEPTH", "4"),
                                        ("PAPOF_SOR_DEPTH", "8"), ("PAPOF_SOR_DEPTH", "12"), ("PAPOF_SOR_XCD", "0"),
                                        ("PAPOF_SOR_XCD", "2"), ("PAPOF_OVERLAP", "0"), ("PAPOF_SOR_XLANE", "shfl"),
                                        ("PAPOF_SOR_FUSE", "2"), ("PAPOF_SOR_FUSE", "1"), ("PAPOF_SOR_FUSE+DEPTH", "2+10"),
                                        ("PAPOF_SOR_RESIDENT", "24"), ("PAPOF_SOR_RESIDENT+FUSE", "24+2"),
                                        ("PAPOF_SOR_RESIDENT+GROUP", "24+4")])
def test_every_solver_variant_matches_oracle(oracle, knob, value, monkeypatch):
    """Every tuning knob selects code that must give the reference's bits too: the opt-in grouped solver
    (PAPOF_SOR_GROUP: M sweeps of a band per workgroup, LDS hand-off; sor.hip k_sor_group), the pipeline depths the
    size heuristic does not pick, the XCD mappings, the single-stream orchestration, the ds_bpermute lane shifts, and
    the fused kernel (PAPOF_SOR_FUSE=2: two sweeps per wave, k_sor_fused; even and odd sweep counts, one and many bands)
    forced on for every size / forced off; PAPOF_SOR_RESIDENT: a solve split into many consecutive launches over ranges
    of sweeps (what sor_solve does when a solve has more tasks than the chip keeps resident), for all three kernels."""
    from papteam_opticalflow_amd import Papof
    group = knob + "=" + value
    if "+" in knob:  # two knobs at once
        for kn, va in zip(knob[len("PAPOF_SOR_"):].split("+"), value.split("+")):
            monkeypatch.setenv("PAPOF_SOR_" + kn, va)
    else:
        monkeypatch.setenv(knob, value)
    g = Papof(0)  # the environment is read when the handle is created
    try:
        for h, w, n_sor in [(70, 50, 4), (130, 37, 3), (1, 5, 3), (129, 3, 2), (341, 607, 42), (540, 960, 30),
                            (200, 1000, 7), (63, 64, 1), (125, 66, 9), (61, 40, 2), (60, 33, 5), (1100, 300, 5)]:
            planes = _sor_planes(h, w, h * 7 + w)
            du, dv = g.sor(*planes, n_sor, mode=0)
            eu, ev = oracle.sor(*planes, n_sor, mode=0)
            assert np.array_equal(du, eu) and np.array_equal(dv, ev), (group, h, w, n_sor)
        a, b = cases.load_pair("240")
        got = g.coarse2fine_flow(a, b, 4)[:3]
        want = oracle.coarse2fine_flow(a, b, 4)[:3]
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
    finally:
        g.close()


@pytest.mark.parametrize("knob,value", [("PAPOF_SOR_FUSE", "1"), ("PAPOF_SOR_FUSE", "2"), ("PAPOF_SOR_GROUP", "4")])
def test_raised_abort_word_ends_every_task_and_reports_timeout(oracle, knob, value, monkeypatch):
    """The device-side waits of the exact-order kernels are bounded: a task that gives up raises an abort word, every other
    task ends when it sees it, and the call returns PAPOF_ETIMEOUT (-5).  Injected here by raising the word before the
    launch (PAPOF_SOR_INJECT_ABORT); afterwards the same handle must solve correctly again."""
    from papteam_opticalflow_amd import Papof
    from papteam_opticalflow_amd.capi import PapofError
    monkeypatch.setenv(knob, value)
    g = Papof(0)
    try:
        planes = _sor_planes(200, 300, 7)
        want = oracle.sor(*planes, 6, mode=0)
        monkeypatch.setenv("PAPOF_SOR_INJECT_ABORT", "1")
        with pytest.raises(PapofError) as err:
            g.sor(*planes, 6, mode=0)
        assert err.value.code == -5
        monkeypatch.delenv("PAPOF_SOR_INJECT_ABORT")
        got = g.sor(*planes, 6, mode=0)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    finally:
        g.close()


def test_flow16_file_roundtrip_and_reference_layout(gpu, oracle, tmp_path):
    """save_flow16 / load_flow16: the reference's SaveOpticalFlow file layout (src/Image.h:825-837) around the GPU
    quantisation; the payload must be the oracle's bytes, and reading back gives the dequantised flow."""
    import struct
    from papteam_opticalflow_amd import load_flow16, save_flow16
    a, b = cases.load_pair("240")
    vx, vy, _, _ = gpu.coarse2fine_flow(a, b, 3)
    p = str(tmp_path / "flow.bin")
    save_flow16(p, vx, vy)
    raw = open(p, "rb").read()
    assert raw[:2] == b"t\0" and struct.unpack("<iii?", raw[16:29]) == (240, 135, 2, False)
    q = oracle.flow_quantize16(vx, vy)
    assert raw[29:] == q.tobytes()
    lx, ly = load_flow16(p)
    ox, oy = oracle.flow_dequantize16(q)
    assert np.array_equal(lx, ox) and np.array_equal(ly, oy)
    assert np.abs(lx - vx).max() <= 1 / 160.0


def test_flow_visualisation_matches_its_numpy_restatement(gpu):
    """generateOutputFlowImageFile (OpticalFlowCalculation.py:143-162) without cv2.  PARITY UNPINNED against OpenCV
    (not installed; its cartToPolar is approximate): checked against a numpy restatement of the same formulas with
    one hue step / one grey level of slack, plus the properties any implementation must have."""
    rng = np.random.default_rng(3)
    vx = cases.smooth_field(rng, 70, 90, 3.0)
    vy = cases.smooth_field(rng, 70, 90, 2.0)
    got = gpu.flow_to_bgr(vx, vy).astype(np.int32)
    mag = np.sqrt(vx * vx + vy * vy)
    ang = np.mod(np.arctan2(vy, vx), 2 * np.pi)
    H = (ang * 180 / np.pi / 2).astype(np.uint8).astype(np.float32)
    V = ((mag - mag.min()) * (255.0 / (mag.max() - mag.min()))).astype(np.uint8).astype(np.float32) / np.float32(255)
    hh = H * np.float32(6.0 / 180.0)
    sec = np.floor(hh).astype(int)
    f = (hh - sec).astype(np.float32)
    tab = np.stack([V, np.zeros_like(V), V * (1 - f), V * f], axis=-1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])[sec % 6]
    want = np.rint(np.take_along_axis(tab, sd, axis=-1) * 255).astype(np.int32)
    assert np.abs(got - want).max() <= 9 and (got != want).mean() < 0.02  # one hue step moves a channel by <= 255 * 6 / 180
    z = gpu.flow_to_bgr(np.zeros((4, 5)), np.zeros((4, 5)))
    assert not z.any()  # no motion: black
    e = gpu.flow_to_bgr(np.array([[1.0, 0.0, -1.0, 0.0]]), np.array([[0.0, 1.0, 0.0, -1.0]]))
    assert e.shape == (1, 4, 3) and not e.any()  # equal magnitudes normalise to value 0
    r = gpu.flow_to_bgr(np.array([[2.0, 0.0]]), np.array([[0.0, 0.0]]))
    assert tuple(r[0, 0]) == (0, 0, 255) and not r[0, 1].any()  # rightward motion: hue 0 = red, full value


def test_laplacian_noise_guard_tripping_case(gpu, oracle):
    """The golden case `stage_lapguard` (tests/golden/cases.py) -- inputs on which the reference's `LapPara[k] < 1E-20`
    guard (src/OpticalFlow.cpp:399-400, fed by estLaplacianNoise, :594-639) trips, pinned bit for bit between the oracle and
    the untouched reference -- through the GPU stage (which always runs the exact pass of the guard, api.hip: LapGuard),
    against the oracle AND against the reference's own numbers in golden.npz."""
    f1, f2, z, alpha = cases.lapguard_inputs()
    got = gpu.smoothflow_sor(f1, f2, f2, z, z, alpha, 3, 1, 5)
    want = oracle.smoothflow_sor(f1, f2, f2, z, z, alpha, 3, 1, 5)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    gold = np.load(os.path.join(GOLD, "golden.npz"))
    for k, a in cases.stage_lapguard(gpu).items():
        w = gold["stage_lapguard|" + k]
        assert np.array_equal(cases.subsample(a), w), "stage_lapguard/%s: max-abs %.3e" % (k, np.abs(cases.subsample(a) - w).max())


@pytest.mark.parametrize("h,w,levels", [(135, 240, 3), (101, 173, 3), (270, 480, 2), (37, 53, 2)])
def test_one_kernel_and_two_kernel_forms_of_the_system_give_the_same_bits(gpu, h, w, levels, monkeypatch):
    """k_flow_system (the default: flow -> solver operands in one launch) against the pair of kernels it replaces
    (k_warp_smooth_blend + k_assemble_skew, still the path of the other branches; PAPOF_FUSED_SYSTEM=0): same bits, on sizes
    with ragged tiles at both borders; gray frames (3 feature channels) too."""
    res = "240" if w <= 240 else "480"
    a, b = cases.load_pair(res)
    a, b = np.ascontiguousarray(a[:h, :w]), np.ascontiguousarray(b[:h, :w])
    one = gpu.coarse2fine_flow(a, b, levels)[:3]
    g1 = gpu.coarse2fine_flow(np.ascontiguousarray(a[..., :1]), np.ascontiguousarray(b[..., :1]), levels)[:3]
    monkeypatch.setenv("PAPOF_FUSED_SYSTEM", "0")
    two = gpu.coarse2fine_flow(a, b, levels)[:3]
    g2 = gpu.coarse2fine_flow(np.ascontiguousarray(a[..., :1]), np.ascontiguousarray(b[..., :1]), levels)[:3]
    for x, y in zip(one + g1, two + g2):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("h,w", [(1, 9), (3, 4), (5, 7), (16, 33), (17, 16), (2, 2)])
def test_frames_of_a_few_pixels(gpu, oracle, h, w):
    """Frames far smaller than any tile (the arena once asked for the one-workgroup solver's full 8192-cell planes whatever the
    frame size: PAPOF_ENOMEM below ~25 x 25 pixels): one level, and as many levels as the size allows, against the oracle."""
    a, b = cases.load_pair("240")
    a, b = np.ascontiguousarray(a[40:40 + h, 60:60 + w]), np.ascontiguousarray(b[40:40 + h, 60:60 + w])
    for levels in (1, 2):
        if levels == 2 and (int(h * 0.75) < 1 or int(w * 0.75) < 1):
            continue
        got = gpu.coarse2fine_flow(a, b, levels)[:3]
        want = oracle.coarse2fine_flow(a, b, levels)[:3]
        for name, x, y in zip(("vx", "vy", "warpI2"), got, want):
            _cmp("%dx%d L%d %s" % (h, w, levels, name), x, y, TOL_SOLVE)


def _tiny_scale_pair(res="240", scale=1e-21):
    """float frames whose every |Im1 - warpIm2| is ~1e-21: each feature channel's noise estimate is below 1E-20 from the
    first outer iteration on, so the reference's guard leaves every psi at 0 from the second one on"""
    a, b = cases.load_pair(res)
    return np.ascontiguousarray(a * scale), np.ascontiguousarray(b * scale)


def _rel(got, want):
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))


def test_laplacian_noise_guard_whole_call_rerun_and_sticky_exact_pass(oracle):
    """The guard inside the WHOLE call.  A pair on which it trips: the optimistic pass ends without proofs, the call is run
    again in the exact pass (reruns 0 -> 1) and agrees with the oracle; the next call on the handle starts in the exact pass
    (no second run); an ordinary pair then runs once in the exact pass -- with the same bits as on a fresh handle -- proves
    the pass unnecessary, and the handle is back to the optimistic pass."""
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    try:
        assert g.lap_guard_stats() == dict(reruns=0, exact_calls=0, exact_next=False, guard_on=True)
        a, b = _tiny_scale_pair()
        want = oracle.coarse2fine_flow(a, b, 3)[:3]
        assert np.abs(want[0]).max() > 0
        got = g.coarse2fine_flow(a, b, 3)[:3]
        st = g.lap_guard_stats()
        assert st["reruns"] == 1 and st["exact_next"] is True, st
        for name, x, w in zip(("vx", "vy", "warpI2"), got, want):
            assert _rel(x, w) <= 1e-9, "%s: relative difference %.3e" % (name, _rel(x, w))
            print("tripping pair, %s: relative difference to the oracle %.3e %s" % (name, _rel(x, w), "(bit-exact)" if np.array_equal(x, w) else ""))
        again = g.coarse2fine_flow(a, b, 3)[:3]
        st = g.lap_guard_stats()
        assert st["reruns"] == 1 and st["exact_calls"] == 1 and st["exact_next"] is True, st
        for x, w in zip(again, got):
            assert np.array_equal(x, w)
        # an ordinary pair: once in the exact pass, then optimistic again; the two passes give the same bits
        p, q = cases.load_pair("240")
        first = g.coarse2fine_flow(p, q, 3)[:3]
        st = g.lap_guard_stats()
        assert st["reruns"] == 1 and st["exact_calls"] == 2 and st["exact_next"] is False, st
        second = g.coarse2fine_flow(p, q, 3)[:3]
        assert g.lap_guard_stats() == dict(reruns=1, exact_calls=2, exact_next=False, guard_on=True)
        for x, w in zip(first, second):
            assert np.array_equal(x, w)
    finally:
        g.close()


@pytest.mark.parametrize("c", [2, 4])
def test_laplacian_noise_guard_on_pass_through_channel_counts(oracle, c):
    """frames with 2 or 4..8 channels take im2feature's pass-through branch (src/OpticalFlow.cpp:956-960), which
    writes no non-zero flags -- so no channel may be taken for "all zero" (a false proof).  A tripping pair of such frames must be
    run again in the exact pass and agree with the oracle (parity for these channel counts is pinned by no reference golden:
    the oracle restates the branch)."""
    from papteam_opticalflow_amd import Papof
    a, b = _tiny_scale_pair()
    a = np.ascontiguousarray(np.concatenate([a, a[..., :1]], axis=2)[..., :c])
    b = np.ascontiguousarray(np.concatenate([b, b[..., :1]], axis=2)[..., :c])
    want = oracle.coarse2fine_flow(a, b, 3)[:3]
    g = Papof(0)
    try:
        got = g.coarse2fine_flow(a, b, 3)[:3]
        st = g.lap_guard_stats()
        assert st["reruns"] == 1 and st["exact_next"] is True, st
        for name, x, w in zip(("vx", "vy", "warpI2"), got, want):
            assert _rel(x, w) <= 1e-9, "%d channels, %s: relative difference %.3e" % (c, name, _rel(x, w))
    finally:
        g.close()


def test_laplacian_noise_guard_switch_is_what_makes_the_difference(oracle, monkeypatch):
    """PAPOF_LAP_GUARD=0 (an A/B switch for the guard's cost) on the tripping pair: far from the oracle -- i.e. the test above
    does tell a path with the guard from one without."""
    from papteam_opticalflow_amd import Papof
    a, b = _tiny_scale_pair()
    want = oracle.coarse2fine_flow(a, b, 3)[0]
    monkeypatch.setenv("PAPOF_LAP_GUARD", "0")
    g = Papof(0)
    try:
        got = g.coarse2fine_flow(a, b, 3)[0]
        assert g.lap_guard_stats()["guard_on"] is False and g.lap_guard_stats()["reruns"] == 0
    finally:
        g.close()
    assert _rel(got, want) > 1e-3


@pytest.mark.parametrize("kind", ["rgb_u8", "gray_u8", "gray_as_rgb", "float", "sequence", "identical", "black", "deep"])
def test_laplacian_noise_guard_never_reruns_ordinary_input(kind):
    """Ordinary inputs end the optimistic pass with a proof for every consulted estimate -- a witness sample, or a feature
    channel that is zero throughout (gray content in three channels: G-R and G-B; black frames) -- and run ONCE."""
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    try:
        a, b = cases.load_pair("240")
        a8, b8 = (np.clip(np.rint(a * 255), 0, 255).astype(np.uint8) for a in (a, b))
        levels = 3
        if kind == "rgb_u8":
            g.coarse2fine_flow_u8(a8, b8, levels)
        elif kind == "gray_u8":
            g.coarse2fine_flow_u8(np.ascontiguousarray(a8[..., :1]), np.ascontiguousarray(b8[..., :1]), levels)
        elif kind == "gray_as_rgb":
            ga, gb = (np.ascontiguousarray(np.repeat(x[..., 1:2], 3, axis=2)) for x in (a8, b8))
            g.coarse2fine_flow_u8(ga, gb, levels)
        elif kind == "float":
            g.coarse2fine_flow(a, b, levels)
        elif kind == "sequence":
            for f in (a8, b8, a8, b8):
                g.seq_push(f, levels)
        elif kind == "identical":  # no valid sample anywhere: LapPara = 0.001 -- but nothing proves it: see below
            pass
        elif kind == "black":
            z = np.zeros_like(a8)
            g.coarse2fine_flow_u8(z, z, levels)
        elif kind == "deep":  # 15 levels: the coarsest are a few pixels, the flow leaves them altogether in some iterations --
            g.coarse2fine_flow_u8(a8, b8, 15)  # no valid sample at all, which the one-block levels' exhaustive check proves
        st = g.lap_guard_stats()
        if kind != "identical":
            assert st["reruns"] == 0 and st["exact_calls"] == 0 and st["exact_next"] is False, (kind, st)
        else:
            # duplicate frames: every |Im1 - warpIm2| is exactly 0 while the flow stays 0 -- no witness exists, the call is run
            # again in the exact pass (where LapPara = 0.001 keeps the guard open): same results, twice the time, and the
            # handle stays in the exact pass while the frames stay duplicates
            r1 = g.coarse2fine_flow_u8(a8, a8, levels)[:3]
            st = g.lap_guard_stats()
            assert st["reruns"] == 1 and st["exact_next"] is True, st
            assert not r1[0].any() and not r1[1].any()
    finally:
        g.close()


def test_flow_system_tile_widths_give_the_same_bits(gpu):
    """k_flow_system's tile is 16 rows x TX columns, TX = 16 (256 threads) or 32 (512 threads; PAPOF_FS_TX, read once per process:
    hence child processes).  Same operations per cell, so the same bits: whole calls on ragged sizes (tiles cut at both borders),
    gray frames, the row-major form of the small levels, against THIS process's results (which the other tests pin)."""
    import subprocess
    import sys
    code = r'''
import sys, hashlib, numpy as np
sys.path[:0] = [%r, %r, %r]
import cases
from papteam_opticalflow_amd import Papof
g = Papof(0)
for res, h, w, lv, gray in (("240", 135, 240, 3, 0), ("240", 101, 173, 3, 0), ("480", 270, 480, 4, 0), ("480", 203, 311, 2, 1), ("240", 37, 53, 2, 0)):
    a, b = cases.load_pair(res)
    a, b = np.ascontiguousarray(a[:h, :w]), np.ascontiguousarray(b[:h, :w])
    if gray:
        a, b = np.ascontiguousarray(a[..., :1]), np.ascontiguousarray(b[..., :1])
    out = g.coarse2fine_flow(a, b, lv)[:3]
    print(res, h, w, lv, gray, " ".join(hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()[:16] for x in out))
''' % (ROOT, os.path.join(ROOT, "tests"), GOLD)
    outs = {}
    for tx in ("16", "32"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, PAPOF_FS_TX=tx))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tx] = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(outs[tx]) == 5, r.stdout
    assert outs["16"] == outs["32"], (outs["16"], outs["32"])
    a, b = cases.load_pair("240")
    here = gpu.coarse2fine_flow(a, b, 3)[:3]
    assert outs["32"][0].split()[5:] == [hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()[:16] for x in here]
