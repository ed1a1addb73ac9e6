"""Lane-level model of the exact-order SOR kernel `k_sor_exact<R>` (papteam_opticalflow_amd/csrc/sor.hip).

The kernel keeps the reference's in-place lexicographic sweep order (src/OpticalFlow.cpp:458-505) while running
thousands of cells concurrently.  Work decomposition, as in the HIP code:

* one wavefront executes one TASK = (band b, sweep k).  Bands are TIME-SKEWED: at sweep k band b owns the 62 rows
  62b - k .. 62b - k + 61 (lanes 1..62); lane 0 / lane 63 are GHOST lanes for the row above / below.  Shifting
  the bands up by one row per sweep makes the row below a band (needed with its previous-sweep value) a row the SAME
  band owned one sweep earlier, so a task depends only on (b, k-1) and (b-1, k) -- never on the band below;
* at STEP s lane l works on row 62b - k - 1 + l, column j = s - l (the lane above is one column ahead): NS = W + 63 steps;
* the six coefficient operands live in GLOBALLY skewed planes: cell (row i, column j) at [position i + j + QT]
  [row i + RT], every non-cell 0.0, so borders need no predicates; the 64 lanes of a task read 64 consecutive rows of
  one position (ghost lanes read the neighbouring rows' phi; their a1 = a2 = 0 and omega-1 -> 1 turn the update into
  a pass-through);
* the unknowns (du, dv) live in BANDED, PING-PONG planes D[k & 1][position][band][64 cells]: task (b, k) writes, at
  step s, all 64 lanes (ghosts write their pass-through) to D[k&1][s + 1][b] -- an aligned 1-KiB block that no other
  task of the sweep touches -- and reads its old values from D[(k-1)&1][.][b] shifted by one cell (lane l <- cell
  l - 1: the bands climb one row per sweep, and cell 0 is the previous sweep's pass-through of the row above);
  ghost lane 0 reads the NEW value of the row above from D[k&1][s + 64][b-1][62];
* left-new = the lane's own previous result; up-new = previous result of lane l-1; down-old = the pending centre of
  lane l+1; right-old is LOADED (it becomes the next centre);
* every load is issued R steps before its use (software pipeline).  Before iteration i issues its loads -- which are
  for steps < (i+2)R =: e -- the task needs
      prog[k-1][b]   >= min(NS, e)         own band, previous sweep (centre / right-old / row below)
      prog[k][b-1]   >= min(NS, e + 63)    band above, this sweep   (ghost lane 0)
      prog[k-2][b+1] >= min(NS, e - 62)    band below, two sweeps ago: it must have read cell 62 of our block before
                                           this sweep reuses the block (write-after-read; practically never binding)
  and it publishes only steps whose stores are PROVEN complete by the in-order retirement of the memory pipeline
  (two marker loads per iteration: i*R + R/2 by the end of iteration i), finally NS after a full drain.

`simulate()` executes exactly that dataflow with numpy (same operation order, no FMA) under a RANDOM task scheduler
that honours only the progress conditions above, with loads really taken R steps early and publications really
lagging, so a too-weak condition shows up as a mismatch with the oracle.
"""
import numpy as np

LANES = 64
ROWS = LANES - 2


def layout(h, w, n_sor, r=8):
    nb = (h + n_sor - 1 + ROWS - 1) // ROWS
    ns = w + LANES - 1
    rt = n_sor + 1                      # rows of padding above row 0 (bands climb one row per sweep)
    qt = n_sor + 1                      # positions of padding before position 0
    hp = (rt + ROWS * nb + 2 + 7) // 8 * 8
    npos = qt + ns + 2 * r + 2 + ROWS * (nb - 1) + 2
    return dict(nb=nb, ns=ns, rt=rt, qt=qt, hp=hp, npos=npos)


def to_skew(plane, lay):
    h, w = plane.shape
    out = np.zeros((lay["npos"], lay["hp"]))
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    out[ii + jj + lay["qt"], ii + lay["rt"]] = plane
    return out


def from_skew(sk, h, w, lay):
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return sk[ii + jj + lay["qt"], ii + lay["rt"]]


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


def simulate(phi, imdxy, a1, a2, b1, b2, n_sor, alpha, omega, r=8, seed=0, use_dn2=True):
    """a1 = omega/(imdx2 + alpha*0.05 + coeff), a2 likewise (see sor_coefficients).  Returns du, dv (H x W)."""
    h, w = phi.shape
    lay = layout(h, w, n_sor, r)
    nb, ns, rt, qt = lay["nb"], lay["ns"], lay["rt"], lay["qt"]
    P = {n: to_skew(p, lay) for n, p in dict(phi=phi, xy=imdxy, a1=a1, a2=a2, b1=b1, b2=b2).items()}
    npos_d = ns + 2 * r + 4
    D = np.zeros((2, 2, npos_d, nb, LANES))  # [du|dv][parity][position][band][cell]; memset before every solve
    prog = np.zeros((nb, n_sor), dtype=np.int64)
    nalpha = -alpha
    om1 = np.full(LANES, 1 - omega)
    om1[0] = om1[63] = 1.0
    real = np.zeros(LANES, dtype=bool)
    real[1:63] = True
    n_iter = (ns + r - 1) // r
    rng = np.random.default_rng(seed)

    def covered(b, k, e):
        ok = True
        if k > 0:
            ok = ok and prog[b, k - 1] >= min(ns, e)
        if b > 0:
            ok = ok and prog[b - 1, k] >= min(ns, e + 63)
        if use_dn2 and k > 1 and b + 1 < nb:
            ok = ok and prog[b + 1, k - 2] >= min(ns, max(0, e - 62))
        return ok

    def window(b, k):
        r0 = ROWS * b - k - 1  # row of ghost lane 0; lanes 1..62 = rows 62b-k .. 62b-k+61
        return r0 + qt, slice(r0 + rt, r0 + rt + LANES)

    def load_pd(c, b, k, s):
        """right-old of step s (-1: the first centre): lanes >= 1 <- cells lane-1 of own block, previous parity;
        lane 0 <- cell 62 of the block above, this parity, 63 positions ahead"""
        v = np.zeros(LANES)
        v[1:] = D[c, (k - 1) & 1, s + 1, b, :63]
        if b > 0 and s + 64 < npos_d:
            v[0] = D[c, k & 1, s + 64, b - 1, 62]
        return v

    def load_slot(b, k, s):
        q0, rows = window(b, k)
        g = lambda n: P[n][q0 + s, rows].copy()
        z = lambda n: np.where(real, P[n][q0 + s, rows], 0.0)  # ghost lanes: a = b = 0 (descriptor out of range)
        return dict(phi=g("phi"), xy=g("xy"), a1=z("a1"), a2=z("a2"), b1=z("b1"), b2=z("b2"),
                    duR=load_pd(0, b, k, s), dvR=load_pd(1, b, k, s))

    pending = [Task(b, k, r) for k in range(n_sor) for b in range(nb)]
    while pending:
        ran = False
        for ti in rng.permutation(len(pending)):
            t = pending[ti]
            b, k = t.b, t.k
            if t.i < 0:  # prologue (staged start-up): coverage of steps < R, then the first centre and the first R slots;
                # the refills of iteration 0 (steps < 2R) are checked at its start like those of every other iteration
                if not covered(b, k, r):
                    continue
                t.duC, t.dvC = load_pd(0, b, k, -1), load_pd(1, b, k, -1)
                for s in range(r):
                    t.slots[s] = load_slot(b, k, s)
                t.i = 0
                ran = True
                break
            i = t.i
            if not covered(b, k, (i + 2) * r):
                continue
            ran = True
            for s in range(i * r, (i + 1) * r):
                c = t.slots[s % r]
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
                D[0, k & 1, s + 1, b, :] = duN  # all 64 lanes: ghosts store their pass-through
                D[1, k & 1, s + 1, b, :] = dvN
                t.duL, t.dvL, t.phiL = duN, dvN, c["phi"]
                t.duC, t.dvC = c["duR"], c["dvR"]
                t.slots[s % r] = load_slot(b, k, s + r)  # refill R steps ahead: reads memory NOW
            prog[b, k] = min(ns, i * r + r // 2)  # marker scheme: published by the end of iteration i
            t.i += 1
            if t.i == n_iter:
                prog[b, k] = ns
                pending.pop(ti)
            break
        assert ran, "deadlock in the task graph"
    # read-out: row i of the last sweep K-1 sits in band (i+K-1)//62, cell 1 + (i+K-1)%62, position j + cell + 1
    kl = n_sor - 1
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    bb, cc = (ii + kl) // ROWS, 1 + (ii + kl) % ROWS
    return D[0, kl & 1, jj + cc + 1, bb, cc], D[1, kl & 1, jj + cc + 1, bb, cc]


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


def simulate_grouped(phi, imdxy, a1, a2, b1, b2, n_sor, alpha, omega, r=8, group=4, ring=16, seed=0):
    """Model of the GROUPED kernel `k_sor_group<R, M>` (sor.hip): wave m of workgroup (band b, group g) runs sweep
    k = g*M + m.  Inside a group the (du, dv) cells of a sweep reach the next sweep through a `ring`-slot LDS ring
    guarded by two progress words per producing wave (`done`: steps written, published per half-iteration of R/2
    steps; `taken`: steps the consumer has read -- the producer may be at most `ring` steps ahead).  Only wave 0 reads
    the ping-pong planes (which alternate per GROUP) and only the last wave of a group writes them; the new value of the
    row above (ghost lane 0) comes from per-sweep halo rows HALO[k][b][position] that lane 62 of every wave writes, so
    no write-after-read dependency on the band below remains.  Cross-workgroup progress counters as in `simulate`.
    Executed under a random scheduler at half-iteration granularity; loads are taken when the kernel issues them."""
    h, w = phi.shape
    lay = layout(h, w, n_sor, r)
    nb, ns, rt, qt = lay["nb"], lay["ns"], lay["rt"], lay["qt"]
    P = {n: to_skew(p, lay) for n, p in dict(phi=phi, xy=imdxy, a1=a1, a2=a2, b1=b1, b2=b2).items()}
    half = r // 2
    n_iter = (ns + r - 1) // r
    n_total = n_iter * r
    npos_d = ns + 2 * 32 + 72
    D = np.zeros((2, 2, npos_d, nb, LANES))          # [du|dv][group parity][position][band][cell]
    HALO = np.zeros((2, n_sor, nb, npos_d))          # [du|dv][sweep][band][position]
    RING = np.zeros((2, n_sor, nb, ring, LANES))     # ring written by task (b, k) for (b, k+1) of the same group
    prog = np.zeros((nb, n_sor), dtype=np.int64)
    done = np.zeros((nb, n_sor), dtype=np.int64)     # LDS word: steps of (b, k) in its ring
    taken = np.zeros((nb, n_sor), dtype=np.int64)    # LDS word: steps of ring (b, k) read by (b, k+1)
    nalpha = -alpha
    om1 = np.full(LANES, 1 - omega)
    om1[0] = om1[63] = 1.0
    real = np.zeros(LANES, dtype=bool)
    real[1:63] = True
    rng = np.random.default_rng(seed)

    def role(k):
        m = k % group
        first = m == 0
        last = m == group - 1 or k == n_sor - 1
        return m, first, last

    def covered(b, k, e):
        m, first, _ = role(k)
        ok = True
        if first and k > 0:
            ok = ok and prog[b, k - 1] >= min(ns, e + 1)
        if b > 0:
            ok = ok and prog[b - 1, k] >= min(ns, e + 63)
        return ok

    def window(b, k):
        r0 = ROWS * b - k - 1
        return r0 + qt, slice(r0 + rt, r0 + rt + LANES)

    def load_global_pd(c, b, k, s):
        """what the kernel's (du, dv) load instruction of step s returns: wave 0 lanes >= 1 <- previous group's plane,
        lane 0 <- halo row of the band above; LDS-fed lanes load nothing (0)"""
        m, first, _ = role(k)
        g = k // group
        v = np.zeros(LANES)
        if first:
            v[1:] = D[c, (g + 1) & 1, s + 1, b, :63]
        if b > 0 and 0 <= s + 64 < npos_d:
            v[0] = HALO[c, k, b - 1, s + 64]
        return v

    def load_slot(b, k, s):
        q0, rows = window(b, k)
        gg = lambda n: P[n][q0 + s, rows].copy()
        z = lambda n: np.where(real, P[n][q0 + s, rows], 0.0)
        return dict(phi=gg("phi"), xy=gg("xy"), a1=z("a1"), a2=z("a2"), b1=z("b1"), b2=z("b2"),
                    duR=load_global_pd(0, b, k, s), dvR=load_global_pd(1, b, k, s))

    class GT:
        def __init__(self, b, k):
            self.b, self.k, self.hi = b, k, -1      # hi: next half-iteration index (-1: prologue)
            z = np.zeros(LANES)
            self.duL, self.dvL, self.phiL, self.duC, self.dvC = z.copy(), z.copy(), z.copy(), z.copy(), z.copy()
            self.slots = [None] * r

    pending = [GT(b, k) for k in range(n_sor) for b in range(nb)]
    while pending:
        ran = False
        for ti in rng.permutation(len(pending)):
            t = pending[ti]
            b, k = t.b, t.k
            m, first, last = role(k)
            g = k // group
            if t.hi < 0:
                if not covered(b, k, 2 * r):
                    continue
                t.duC, t.dvC = load_global_pd(0, b, k, -1), load_global_pd(1, b, k, -1)
                for s in range(r):
                    t.slots[s] = load_slot(b, k, s)
                t.hi = 0
                ran = True
                break
            i, sa = t.hi // 2, t.hi * half
            if t.hi % 2 == 0 and i > 0 and not covered(b, k, (i + 2) * r):
                continue
            if not first and done[b, k - 1] < sa + half:           # the wave before has not written these steps yet
                continue
            if not last and sa + half > ring and taken[b, k] < sa + half - ring:  # ring slots still unread
                continue
            ran = True
            lds = None
            if not first:  # H blocks out of the ring of the wave before, cell lane - 1
                lds = [(RING[0, k - 1, b, (sa + q) % ring].copy(), RING[1, k - 1, b, (sa + q) % ring].copy())
                       for q in range(half)]
            for q in range(half):
                s = sa + q
                c = t.slots[s % r]
                duR, dvR = c["duR"].copy(), c["dvR"].copy()
                if not first:
                    duR[1:], dvR[1:] = lds[q][0][:63], lds[q][1][:63]
                duU, dvU, phiU = shift_up(t.duL), shift_up(t.dvL), shift_up(t.phiL)
                duD, dvD = shift_down(duR), shift_down(dvR)
                s1 = t.phiL * t.duL
                s2 = t.phiL * t.dvL
                s1 = s1 + c["phi"] * duR
                s2 = s2 + c["phi"] * dvR
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
                if last:
                    D[0, g & 1, s + 1, b, :] = duN
                    D[1, g & 1, s + 1, b, :] = dvN
                else:
                    RING[0, k, b, s % ring] = duN
                    RING[1, k, b, s % ring] = dvN
                if b + 1 < nb and s + 1 < npos_d:
                    HALO[0, k, b, s + 1] = duN[62]
                    HALO[1, k, b, s + 1] = dvN[62]
                t.duL, t.dvL, t.phiL = duN, dvN, c["phi"]
                t.duC, t.dvC = duR, dvR
                t.slots[s % r] = load_slot(b, k, s + r)
            if not first:
                taken[b, k - 1] = sa + half
            if not last:
                done[b, k] = sa + half
            prog[b, k] = max(prog[b, k], min(ns, sa))  # markers: the previous half is proven complete
            t.hi += 1
            if t.hi == 2 * n_iter:
                prog[b, k] = ns
                pending.pop(ti)
            break
        assert ran, "deadlock in the task graph"
    kl = n_sor - 1
    gl = kl // group
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    bb, cc = (ii + kl) // ROWS, 1 + (ii + kl) % ROWS
    return D[0, gl & 1, jj + cc + 1, bb, cc], D[1, gl & 1, jj + cc + 1, bb, cc]
