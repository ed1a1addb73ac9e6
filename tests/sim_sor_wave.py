"""Lane-level model of the exact-order SOR kernel `k_sor_exact<R>` (papteam_opticalflow_amd/csrc/sor.hip).

The kernel keeps the reference's in-place lexicographic sweep order (src/OpticalFlow.cpp:458-505) while running
thousands of cells concurrently.  Work decomposition, as in the HIP code:

* the image is cut into BANDS of 62 rows; one wavefront executes one TASK = (band b, sweep k); lanes 1..62 carry
  the band's rows, lane 0 / lane 63 are GHOST lanes standing for the last row of band b-1 / first row of band b+1;
* at STEP s lane l works on column j = s - l (the lane above is one column ahead), NS = W + 63 steps per task;
* operands live in per-band SKEWED planes: cell (lane l, column j) of band b at position (b, j + l, l), every
  position that is not a real cell holds 0.0, so borders need no predicates;
* left-new = the lane's own previous result; up-new = previous result of lane l-1; down-old = the pending centre of
  lane l+1; right-old is LOADED (it becomes the next centre).  Ghost lanes load the neighbour band's cells
  (band b-1 lane 62 at position +62, band b+1 lane 1 at position -62), pass the value through and never store;
* every load is issued R steps before its use (software pipeline), in SEGMENTS of H = R/G steps; before a segment
  [s_lo, s_hi) issues its loads -- which are for steps < s_hi + R =: e -- the task needs
      prog[k-1][b]   >= min(NS, e + 1)     own band, previous sweep (centre / right-old)
      prog[k][b-1]   >= min(NS, e + 63)    band above, this sweep   (ghost lane 0)
      prog[k-1][b+1] >= min(NS, e - 60)    band below, previous sweep (ghost lane 63)
  and it publishes only steps whose stores are PROVEN complete by the in-order retirement of the memory pipeline:
  with G = 1 and R >= 8 (the shipped configuration) two marker loads per iteration let it publish i*R + R/2 by the
  end of iteration i; otherwise s_hi - R at the end of a segment; finally NS after a full drain.

`simulate()` executes exactly that dataflow with numpy (same operation order, no FMA) under a RANDOM task scheduler
that honours only the progress conditions above, with loads really taken R steps early and publications really
lagging, so a too-weak condition shows up as a mismatch with the oracle.
"""
import numpy as np

LANES = 64
ROWS = LANES - 2


def skew_dims(h, w, r=8):
    nb = (h + ROWS - 1) // ROWS
    ns = w + LANES - 1
    nsp = (ns + 31) // 32 * 32 + 64
    return nb, ns, nsp


def to_skew(plane, r=8):
    h, w = plane.shape
    nb, ns, nsp = skew_dims(h, w, r)
    out = np.zeros((nb, nsp + 1, LANES))
    for i in range(h):
        b, l = divmod(i, ROWS)
        l += 1
        out[b, l:l + w, l] = plane[i]
    return out


def from_skew(sk, h, w):
    out = np.zeros((h, w))
    for i in range(h):
        b, l = divmod(i, ROWS)
        l += 1
        out[i] = sk[b, l:l + w, l]
    return out


def shift_up(x):
    """lane l receives lane l-1's value (lane 0: unspecified -> 0)."""
    y = np.zeros_like(x)
    y[1:] = x[:-1]
    return y


def shift_down(x):
    y = np.zeros_like(x)
    y[:-1] = x[1:]
    return y


class Task:
    def __init__(self, b, k, r):
        self.b, self.k, self.i = b, k, -1  # -1: prologue (initial fill) not done yet
        z = np.zeros(LANES)
        self.duL, self.dvL, self.phiL, self.duC, self.dvC = z.copy(), z.copy(), z.copy(), z.copy(), z.copy()
        self.slots = [None] * r
        self.pending_pub = 0


def simulate(phi, imdxy, a1, a2, b1, b2, n_sor, alpha, omega, r=8, seed=0, g=2):
    """a1 = omega/(imdx2 + alpha*0.05 + coeff), a2 likewise (see sor_coefficients).  Returns du, dv (H x W)."""
    h, w = phi.shape
    nb, ns, nsp = skew_dims(h, w, r)
    P = {n: to_skew(p, r) for n, p in dict(phi=phi, xy=imdxy, a1=a1, a2=a2, b1=b1, b2=b2).items()}
    du = np.zeros((nb, nsp + 1, LANES))  # memset before every solve
    dv = np.zeros((nb, nsp + 1, LANES))
    prog = np.zeros((nb, n_sor), dtype=np.int64)
    nalpha = -alpha
    om1 = np.full(LANES, 1 - omega)
    om1[0] = om1[63] = 1.0
    lane = np.arange(LANES)
    real = (lane >= 1) & (lane <= 62)
    n_iter = (ns + r - 1) // r
    assert r % g == 0
    hseg = r // g
    n_seg = n_iter * g
    rng = np.random.default_rng(seed)

    def covered(b, k, e):
        ok = True
        if k > 0:
            ok = ok and prog[b, k - 1] >= min(ns, e + 1)
        if b > 0:
            ok = ok and prog[b - 1, k] >= min(ns, e + 63)
        if k > 0 and b + 1 < nb:
            ok = ok and prog[b + 1, k - 1] >= min(ns, max(0, e - 60))
        return ok

    def rd(arr, b, pos, ln):
        """one cell of a skewed plane; positions outside the band's own range read padding (0)"""
        if b < 0 or b >= nb or pos < 0 or pos > nsp:
            return 0.0
        return arr[b, pos, ln]

    def load_pd(arr, b, pos):
        """the (du|dv) vector a task of band b loads for skew position `pos`"""
        v = np.where(real, arr[b, min(pos, nsp), :] if 0 <= pos <= nsp else 0.0, 0.0)
        v[0] = rd(arr, b - 1, pos + 62, 62) if b > 0 else 0.0
        v[63] = rd(arr, b + 1, pos - 62, 1) if b + 1 < nb else 0.0
        return v

    def load_slot(b, s):
        g = lambda n: np.where(real, P[n][b, s, :] if s <= nsp else 0.0, 0.0)
        phi_v = g("phi")
        phi_v[0] = rd(P["phi"], b - 1, s + 62, 62) if b > 0 else 0.0
        return dict(phi=phi_v, xy=g("xy"), a1=g("a1"), a2=g("a2"), b1=g("b1"), b2=g("b2"),
                    duR=load_pd(du, b, s + 1), dvR=load_pd(dv, b, s + 1))

    pending = [Task(b, k, r) for k in range(n_sor) for b in range(nb)]
    while pending:
        ran = False
        for ti in rng.permutation(len(pending)):
            t = pending[ti]
            b, k = t.b, t.k
            if t.i < 0:  # prologue: needs coverage of steps < 2R, then the first centre and the first R slots
                if not covered(b, k, 2 * r):
                    continue
                t.duC, t.dvC = load_pd(du, b, 0), load_pd(dv, b, 0)
                for s in range(r):
                    t.slots[s] = load_slot(b, s)
                t.i = 0
                ran = True
                break
            i = t.i  # segment index
            s_lo, s_hi = i * hseg, (i + 1) * hseg
            if i > 0 and not covered(b, k, s_hi + r):
                continue
            ran = True
            for s in range(s_lo, s_hi):
                tt = s % r
                c = t.slots[tt]
                duU, dvU, phiU = shift_up(t.duL), shift_up(t.dvL), shift_up(t.phiL)
                duD, dvD = shift_down(c["duR"]), shift_down(c["dvR"])
                s1 = t.phiL * t.duL
                s2 = t.phiL * t.dvL
                s1 = s1 + c["phi"] * c["duR"]
                s2 = s2 + c["phi"] * c["dvR"]
                s1 = s1 + phiU * duU
                s2 = s2 + phiU * dvU
                s1 = s1 + c["phi"] * duD
                s2 = s2 + c["phi"] * dvD
                s1 = s1 * nalpha
                s2 = s2 * nalpha
                s1 = s1 + c["xy"] * t.dvC
                duN = om1 * t.duC + c["a1"] * (c["b1"] - s1)
                s2 = s2 + c["xy"] * duN
                dvN = om1 * t.dvC + c["a2"] * (c["b2"] - s2)
                if s <= nsp:
                    du[b, s, :] = np.where(real, duN, du[b, s, :])  # ghost lanes never store
                    dv[b, s, :] = np.where(real, dvN, dv[b, s, :])
                t.duL, t.dvL, t.phiL = duN, dvN, c["phi"]
                t.duC, t.dvC = c["duR"], c["dvR"]
                t.slots[tt] = load_slot(b, s + r)  # refill R steps ahead: reads memory NOW
            if g == 1 and r >= 8:
                # marker scheme of the kernel: by the end of iteration i it has published i*R + R/2
                prog[b, k] = min(ns, i * r + r // 2)
            elif s_hi - r > 0:
                prog[b, k] = min(ns, s_hi - r)  # lagging publication
            t.i += 1
            if t.i == n_seg:
                prog[b, k] = ns
                pending.pop(ti)
            break
        assert ran, "deadlock in the task graph"
    return from_skew(du, h, w), from_skew(dv, h, w)


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
