"""Host-logic test: the wave/band/skew schedule of the exact-order SOR kernel, modelled lane by lane in
tests/sim_sor_wave.py and executed under a random task scheduler, reproduces the oracle's in-place
lexicographic sweeps bit for bit."""
import numpy as np
import pytest

import sim_sor_wave as sim


def _planes(h, w, seed):
    rng = np.random.default_rng(seed)
    phi = rng.uniform(0.5, 50.0, (h, w))
    imdxy = rng.uniform(-0.02, 0.02, (h, w))
    imdx2 = rng.uniform(0, 0.05, (h, w))
    imdy2 = rng.uniform(0, 0.05, (h, w))
    r1 = rng.uniform(-0.01, 0.01, (h, w))
    r2 = rng.uniform(-0.01, 0.01, (h, w))
    return phi, imdxy, imdx2, imdy2, r1, r2


@pytest.mark.parametrize("h,w,n_sor,chunk", [(70, 50, 4, 8), (130, 37, 3, 8), (64, 20, 3, 4), (1, 5, 3, 4),
                                              (5, 1, 3, 8), (129, 3, 2, 6), (42, 75, 5, 9), (190, 90, 4, 8)])
def test_wave_schedule_is_bit_exact(oracle, h, w, n_sor, chunk):
    alpha, omega = 0.012, 1.8
    phi, imdxy, imdx2, imdy2, r1, r2 = _planes(h, w, h * 1000 + w)
    a1, a2 = sim.sor_coefficients(phi, imdx2, imdy2, alpha, omega)
    du, dv = sim.simulate(phi, imdxy, a1, a2, r1, r2, n_sor, alpha, omega, r=chunk, seed=w)
    eu, ev = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, n_sor, alpha=alpha, omega=omega, mode=0)
    assert np.array_equal(du, eu)
    assert np.array_equal(dv, ev)


@pytest.mark.parametrize("h,w,n_sor,chunk,group,ring", [(70, 50, 5, 8, 4, 16), (130, 37, 6, 8, 2, 16), (64, 20, 9, 8, 4, 8),
                                                         (1, 5, 3, 4, 4, 4), (129, 3, 4, 6, 4, 16), (42, 75, 7, 10, 4, 16),
                                                         (190, 40, 4, 8, 3, 16)])
def test_grouped_wave_schedule_is_bit_exact(oracle, h, w, n_sor, chunk, group, ring):
    """M sweeps of a band per workgroup, handed from wave to wave through an LDS ring (k_sor_group)."""
    alpha, omega = 0.012, 1.8
    phi, imdxy, imdx2, imdy2, r1, r2 = _planes(h, w, h * 1000 + w + 1)
    a1, a2 = sim.sor_coefficients(phi, imdx2, imdy2, alpha, omega)
    du, dv = sim.simulate_grouped(phi, imdxy, a1, a2, r1, r2, n_sor, alpha, omega, r=chunk, group=group, ring=ring,
                                  seed=h)
    eu, ev = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, n_sor, alpha=alpha, omega=omega, mode=0)
    assert np.array_equal(du, eu)
    assert np.array_equal(dv, ev)


@pytest.mark.parametrize("h,w,n_sor,chunk", [(70, 50, 4, 8), (130, 37, 3, 8), (61, 20, 2, 6), (1, 5, 3, 6), (5, 1, 5, 8),
                                              (129, 3, 2, 6), (42, 75, 7, 10), (190, 40, 6, 8), (60, 9, 1, 6)])
def test_fused_wave_schedule_is_bit_exact(oracle, h, w, n_sor, chunk):
    """Two sweeps of a band per wavefront (k_sor_fused): registers hand the first sweep's results to the second; planes
    whose non-tail positions start out as NaN prove that nothing is read before it is written."""
    alpha, omega = 0.012, 1.8
    phi, imdxy, imdx2, imdy2, r1, r2 = _planes(h, w, h * 1000 + w + 2)
    a1, a2 = sim.sor_coefficients(phi, imdx2, imdy2, alpha, omega)
    du, dv = sim.simulate_fused(phi, imdxy, a1, a2, r1, r2, n_sor, alpha, omega, r=chunk, seed=h + w)
    eu, ev = oracle.sor(phi, imdxy, imdx2, imdy2, r1, r2, n_sor, alpha=alpha, omega=omega, mode=0)
    assert np.array_equal(du, eu)
    assert np.array_equal(dv, ev)


def test_skew_roundtrip():
    rng = np.random.default_rng(0)
    p = rng.standard_normal((150, 33))
    lay = sim.layout(150, 33, 7)
    assert np.array_equal(sim.from_skew(sim.to_skew(p, lay), 150, 33, lay), p)
