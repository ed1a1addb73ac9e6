"""Randomised small cases against the oracle (-m gpu): frame sizes from one pixel up, pyramid depths as far as the size allows,
schedules with odd sweep counts, gray and colour, float64 and uint8 frames -- the corners the fixed parity cases do not reach
(tile borders of k_flow_system, planes of one band, levels that k_sor_tiny solves, the guard's one-block levels)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _levels_ok(h, w, levels, ratio=0.75):
    for k in range(1, levels):
        if int(h * ratio ** k) < 1 or int(w * ratio ** k) < 1:
            return False
    return True


@pytest.mark.parametrize("seed", range(10))
def test_random_small_cases_match_the_oracle(oracle, seed):
    from papteam_opticalflow_amd import Papof, default_params
    rng = np.random.default_rng(1000 + seed)
    a0, b0 = cases.load_pair("480")
    g = Papof(0)
    try:
        for case in range(20):
            h, w = int(rng.integers(1, 90)), int(rng.integers(1, 130))
            y0, x0 = int(rng.integers(0, 270 - h)), int(rng.integers(0, 480 - w))
            a, b = a0[y0:y0 + h, x0:x0 + w], b0[y0:y0 + h, x0:x0 + w]
            if rng.integers(0, 3) == 0:  # gray
                a, b = a[..., 1:2], b[..., 1:2]
            a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
            levels = int(rng.integers(1, 6))  # (up to 5: every level is derived from level 0, its size is int(size * 0.75^k))
            while not _levels_ok(h, w, levels):
                levels -= 1
            mode = int(rng.choice([0, 0, 0, 1, 2]))  # the reference's order, red-black, Jacobi (same-mode oracle)
            kw = dict(n_outer=int(rng.integers(1, 4)), n_outer_per_level=int(rng.integers(0, 2)),
                      n_sor=int(rng.integers(1, 41)), n_sor_per_level=int(rng.integers(0, 4)), sor_mode=mode,
                      omega=1.0 if mode == 2 else 1.8)
            P = default_params(**kw)
            p = oracle.default_params()
            for k, v in kw.items():
                setattr(p, k, v)
            u8 = rng.integers(0, 2) == 0
            print("seed %d case %d: %dx%dx%d L%d %s u8=%s" % (seed, case, h, w, a.shape[2], levels, kw, u8), flush=True)
            if u8:
                a8, b8 = (np.clip(np.rint(x * 255), 0, 255).astype(np.uint8) for x in (a, b))
                got = g.coarse2fine_flow_u8(a8, b8, levels, P)[:3]
                want = oracle.coarse2fine_flow(a8 / 255.0, b8 / 255.0, levels, p)[:3]
            else:
                got = g.coarse2fine_flow(a, b, levels, P)[:3]
                want = oracle.coarse2fine_flow(a, b, levels, p)[:3]
            for name, x, y in zip(("vx", "vy", "warpI2"), got, want):
                assert x.shape == y.shape and np.all(np.isfinite(x)), (seed, case, h, w, levels, kw, name)
                d = float(np.abs(x - y).max()) if x.size else 0.0
                assert d <= 1e-9, "seed %d case %d: %dx%dx%d L%d %s u8=%s: %s max-abs %.3e" % (
                    seed, case, h, w, a.shape[2], levels, kw, u8, name, d)
    finally:
        g.close()


@pytest.mark.parametrize("seed", range(3))
def test_random_medium_cases_match_the_oracle(oracle, seed):
    """The same on frames of one to six solver bands (up to 270 x 480): the skewed layout's row padding, planes with a ragged last
    band, the grouped / plain / one-workgroup solver choices, k_flow_system's border tiles on every side."""
    from papteam_opticalflow_amd import Papof, default_params
    rng = np.random.default_rng(2000 + seed)
    a0, b0 = cases.load_pair("480")
    g = Papof(0)
    try:
        for case in range(6):
            h, w = int(rng.integers(60, 271)), int(rng.integers(60, 481))
            y0, x0 = int(rng.integers(0, 270 - h + 1)), int(rng.integers(0, 480 - w + 1))
            a, b = np.ascontiguousarray(a0[y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(b0[y0:y0 + h, x0:x0 + w])
            levels = int(rng.integers(1, 6))
            kw = dict(n_outer=int(rng.integers(1, 4)), n_outer_per_level=int(rng.integers(0, 2)),
                      n_sor=int(rng.integers(1, 61)), n_sor_per_level=int(rng.integers(0, 4)))
            P = default_params(**kw)
            p = oracle.default_params()
            for k, v in kw.items():
                setattr(p, k, v)
            print("seed %d case %d: %dx%d L%d %s" % (seed, case, h, w, levels, kw), flush=True)
            got = g.coarse2fine_flow(a, b, levels, P)[:3]
            want = oracle.coarse2fine_flow(a, b, levels, p)[:3]
            for name, x, y in zip(("vx", "vy", "warpI2"), got, want):
                d = float(np.abs(x - y).max())
                assert d <= 1e-9, "seed %d case %d: %dx%d L%d %s: %s max-abs %.3e" % (seed, case, h, w, levels, kw, name, d)
    finally:
        g.close()
