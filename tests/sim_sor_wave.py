"""Lane-level model of the exact-order SOR kernel (papteam_opticalflow_amd/csrc/sor_exact.hip).

The kernel keeps the reference's in-place lexicographic sweep order (src/OpticalFlow.cpp:458-505)
while running thousands of cells concurrently.  Work decomposition:

* the image is cut into BANDS of 64 rows; one wavefront (64 lanes, lane r = row r of the band)
  executes one TASK = (band b, sweep k);
* at STEP s lane r updates column j = s - r  ("skewed" march: the lane above is one column ahead),
  so a task takes NS = W + 63 steps;
* per-plane storage is SKEWED per band: element (row r, column j) of band b lives at
  ((b*NS + (j + r))*64 + r), so at step s the 64 lanes touch 64 consecutive doubles;
* left neighbour (new)  = the lane's own previous result (register)
  up   neighbour (new)  = previous result of lane r-1 (cross-lane shift); lane 0 reads band b-1's row 63
  right neighbour (old) = loaded from the skewed plane at step s+1 (it becomes the next centre)
  down neighbour (old)  = lane r+1's pending centre value (cross-lane shift); lane 63 reads band b+1's row 0
* tasks are independent wavefronts synchronised only by per-task progress counters:
  before running steps [s0, s1) task (b,k) needs
      prog[b  ][k-1] >= min(NS, s1 + 1)     (centre/right old values of its own band)
      prog[b-1][k  ] >= min(NS, s1 + 63)    (row 63 of the band above, this sweep)
      prog[b+1][k-1] >= min(NS, s1 - 63)    (row 0 of the band below, previous sweep)
  and publishes prog[b][k] = s1 afterwards.

`simulate()` executes exactly that dataflow with numpy (one vector op per wave instruction, same
operation order as the kernel, no FMA) under a RANDOM task scheduler that honours only the
progress conditions above, so a too-weak condition shows up as a mismatch with the oracle.
"""
import numpy as np

LANES = 64


def skew_dims(h, w):
    nb = (h + LANES - 1) // LANES
    ns = w + LANES - 1
    return nb, ns


def to_skew(plane):
    h, w = plane.shape
    nb, ns = skew_dims(h, w)
    out = np.zeros((nb, ns, LANES))
    for i in range(h):
        b, r = divmod(i, LANES)
        out[b, r:r + w, r] = plane[i]
    return out


def from_skew(sk, h, w):
    out = np.zeros((h, w))
    for i in range(h):
        b, r = divmod(i, LANES)
        out[i] = sk[b, r:r + w, r]
    return out


def shift_up(x, fill0):
    """lane r receives lane r-1's value; lane 0 receives fill0."""
    y = np.empty_like(x)
    y[1:] = x[:-1]
    y[0] = fill0
    return y


def shift_down(x, fill63):
    y = np.empty_like(x)
    y[:-1] = x[1:]
    y[-1] = fill63
    return y


class Task:
    def __init__(self, b, k):
        self.b, self.k, self.s = b, k, 0
        z = np.zeros(LANES)
        self.duL, self.dvL, self.phiL = z.copy(), z.copy(), z.copy()
        self.duC, self.dvC = z.copy(), z.copy()


def simulate(phi, imdxy, a1, a2, b1, b2, n_sor, alpha, omega, chunk=16, seed=0):
    """a1 = omega/(imdx2 + alpha*0.05 + coeff), a2 likewise: computed by the caller exactly as the
    system-assembly kernel does.  Returns du, dv (H x W) after n_sor sweeps."""
    h, w = phi.shape
    nb, ns = skew_dims(h, w)
    S = {n: to_skew(p) for n, p in dict(phi=phi, imdxy=imdxy, a1=a1, a2=a2, b1=b1, b2=b2).items()}
    du = np.full((nb, ns + 1, LANES), np.nan)  # NaN poison: any read of a not-yet-written cell shows up
    dv = np.full((nb, ns + 1, LANES), np.nan)
    prog = np.zeros((nb, n_sor), dtype=np.int64)
    nalpha, om1 = -alpha, 1 - omega
    lane = np.arange(LANES)
    rng = np.random.default_rng(seed)
    tasks = [Task(b, k) for k in range(n_sor) for b in range(nb)]
    pending = list(tasks)
    while pending:
        order = rng.permutation(len(pending))
        ran = False
        for ti in order:
            t = pending[ti]
            b, k = t.b, t.k
            s1 = min(ns, t.s + chunk)
            ok = True
            if k > 0 and prog[b, k - 1] < min(ns, s1 + 1):
                ok = False
            if b > 0 and prog[b - 1, k] < min(ns, s1 + 63):
                ok = False
            if k > 0 and b + 1 < nb and prog[b + 1, k - 1] < min(ns, s1 - 63):
                ok = False
            if not ok:
                continue
            ran = True
            row = b * LANES + lane
            if t.s == 0 and k > 0:  # centre of the very first column = skew position 0 (lane 0 only matters)
                t.duC = np.where(lane == 0, du[b, 0, :], 0.0)
                t.dvC = np.where(lane == 0, dv[b, 0, :], 0.0)
            for s in range(t.s, s1):
                j = s - lane
                valid = (j >= 0) & (j < w) & (row < h)
                ld = lambda a: np.where(valid, a[b, s, :], 0.0)
                phiC, xy, A1, A2, B1, B2 = (ld(S[n]) for n in ("phi", "imdxy", "a1", "a2", "b1", "b2"))
                # right-old = cell (r, j+1) of sweep k-1, skew position s+1
                rvalid = (j + 1 >= 0) & (j + 1 < w) & (row < h)
                if k > 0:
                    duR = np.where(rvalid, du[b, s + 1, :], 0.0)
                    dvR = np.where(rvalid, dv[b, s + 1, :], 0.0)
                else:
                    duR = np.zeros(LANES)
                    dvR = np.zeros(LANES)
                # halos
                upu = upv = upphi = 0.0
                if b > 0 and 0 <= s < w:  # lane 0, column j = s ; band above, row 63 -> skew position s+63
                    upu, upv = du[b - 1, s + 63, 63], dv[b - 1, s + 63, 63]
                    upphi = S["phi"][b - 1, s + 63, 63]
                dnu = dnv = 0.0
                j63 = s - 63
                if k > 0 and b + 1 < nb and 0 <= j63 < w and (b + 1) * LANES < h:
                    dnu, dnv = du[b + 1, j63, 0], dv[b + 1, j63, 0]
                duU, dvU, phiU = shift_up(t.duL, upu), shift_up(t.dvL, upv), shift_up(t.phiL, upphi)
                # down-old = cell (r+1, j) of sweep k-1: lane r+1 sits at column j-1, so it is ITS right-old value
                duD, dvD = shift_down(duR, dnu), shift_down(dvR, dnv)
                wL = t.phiL
                wR = np.where(j < w - 1, phiC, 0.0)
                wU = np.where(row > 0, phiU, 0.0)
                wD = np.where(row < h - 1, phiC, 0.0)
                s1_ = wL * t.duL
                s2_ = wL * t.dvL
                s1_ = s1_ + wR * duR
                s2_ = s2_ + wR * dvR
                s1_ = s1_ + wU * duU
                s2_ = s2_ + wU * dvU
                s1_ = s1_ + wD * duD
                s2_ = s2_ + wD * dvD
                s1_ = s1_ * nalpha
                s2_ = s2_ * nalpha
                s1_ = s1_ + xy * t.dvC
                duN = om1 * t.duC + A1 * (B1 - s1_)
                s2_ = s2_ + xy * duN
                dvN = om1 * t.dvC + A2 * (B2 - s2_)
                duN = np.where(valid, duN, 0.0)
                dvN = np.where(valid, dvN, 0.0)
                du[b, s, :] = np.where(valid, duN, du[b, s, :])
                dv[b, s, :] = np.where(valid, dvN, dv[b, s, :])
                t.duL, t.dvL, t.phiL = duN, dvN, phiC
                t.duC, t.dvC = duR, dvR
            t.s = s1
            prog[b, k] = s1
            if s1 == ns:
                pending.pop(ti)
            break
        assert ran, "deadlock in the task graph"
    return from_skew(du[:, :ns], h, w), from_skew(dv[:, :ns], h, w)


def sor_coefficients(phi, imdx2, imdy2, alpha, omega):
    """a1, a2 exactly as the system-assembly kernel forms them (src/OpticalFlow.cpp:468-501:
    coeff accumulated left, right, up, down; omega/(imdx2 + alpha*0.05 + coeff))."""
    h, w = phi.shape
    coeff = np.zeros((h, w))
    coeff[:, 1:] += phi[:, :-1]
    coeff[:, :-1] += phi[:, :-1]
    coeff[1:, :] += phi[:-1, :]
    coeff[:-1, :] += phi[:-1, :]
    coeff *= alpha
    return omega / (imdx2 + alpha * 0.05 + coeff), omega / (imdy2 + alpha * 0.05 + coeff)
