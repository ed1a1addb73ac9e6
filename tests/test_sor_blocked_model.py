"""The blocked red-black / Jacobi solver's design on the CPU: the numpy model of csrc/sor.hip k_sor_blocked (regions,
shrinking frames, launch plan, ping-pong planes; tests/sim_sor_blocked.py) must give the oracle's bits in the same mode.
The HIP kernel itself is compared with the oracle on the GPU (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import sim_sor_blocked as sim


def _planes(h, w, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0.5, 50.0, (h, w)), rng.uniform(-0.02, 0.02, (h, w)), rng.uniform(0, 0.05, (h, w)),
            rng.uniform(0, 0.05, (h, w)), rng.uniform(-0.01, 0.01, (h, w)), rng.uniform(-0.01, 0.01, (h, w)))


def _diag(planes, alpha=0.012, omega=1.8):
    """the assembly kernel's hoisted diagonals a = omega / (imd?2 + alpha * 0.05 + coeff) (kernels.hip sor_diagonals)"""
    phi, xy, x2, y2, r1, r2 = planes
    h, w = phi.shape
    coeff = np.zeros((h, w))
    coeff[:, 1:] += phi[:, :-1]
    coeff[:, :-1] += phi[:, :-1]
    coeff[1:, :] += phi[:-1, :]
    coeff[:-1, :] += phi[:-1, :]
    coeff *= alpha
    return phi, xy, omega / (x2 + alpha * 0.05 + coeff), omega / (y2 + alpha * 0.05 + coeff), r1, r2


@pytest.mark.parametrize("mode,omega", [(1, 1.8), (2, 1.0)])
@pytest.mark.parametrize("h,w,n_sor,rh,depth", [(70, 150, 5, 48, 10), (135, 240, 7, 32, 10), (100, 300, 6, 48, 6),
                                                  (40, 100, 9, 48, 10), (1, 7, 3, 24, 6), (129, 3, 2, 32, 10),
                                                  (97, 257, 3, 24, 4), (60, 200, 4, 48, 7)])
def test_blocked_model_matches_oracle(oracle, mode, omega, h, w, n_sor, rh, depth):
    planes = _planes(h, w, 11 * h + w)
    want = oracle.sor(*planes, n_sor, omega=omega, mode=mode)
    got = sim.solve(_diag(planes, omega=omega), n_sor, mode, omega=omega, RH=rh, depth=depth if mode == 1 else max(2, depth // 2))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), np.abs(got[0] - want[0]).max()


def test_launch_plan():
    assert sim.plan(60, 1, 10, False) == [10] * 6               # 30 sweeps, depth 10: six launches of five sweeps
    assert sim.plan(66, 1, 10, False) == [10, 10, 10, 10, 10, 8, 8]  # whole sweeps per launch: every depth is even
    assert sim.plan(60, 1, 10, True) == [60]                    # a plane that fits one region: one launch
    assert sim.plan(33, 2, 6, False) == [6, 6, 6, 5, 5, 5]      # Jacobi counts sweeps
    assert sum(sim.plan(2 * 42, 1, 10, False)) == 84
