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
    npos_d = ns + 2 * r + 72
    D = np.zeros((2, 2, npos_d, nb, LANES))  # [du|dv][parity][position][band][cell]
    D[:, :, :ns] = np.nan  # only the tail positions must be zero (sor_solve); poison whatever a task has to write first
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
        if k > 0 and s >= 0:  # sweep 0 reads du = dv = 0 as out-of-range offsets; step -1 is left of column 0
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


FROWS = LANES - 3  # fused pairs: 61 rows per band


def layout_fused(h, w, n_sor, r=8):
    last = (n_sor + 1) // 2 - 1  # index of the last pair of sweeps
    nb = (h + 2 * last + 1 + FROWS - 1) // FROWS
    ns = w + LANES - 1
    rt = qt = n_sor + 3
    hp = (rt + FROWS * nb + 4 + 7) // 8 * 8
    npos = qt + ns + 2 * r + 2 + FROWS * (nb - 1) + 2
    return dict(nb=nb, ns=ns, rt=rt, qt=qt, hp=hp, npos=npos, last=last)


def simulate_fused(phi, imdxy, a1, a2, b1, b2, n_sor, alpha, omega, r=8, seed=0, use_dn2=True):
    """Model of `k_sor_fused<R>` (sor.hip): task (band b, pair q) runs sweeps k = 2q and k + 1 in ONE wavefront.
    Lane l stands for image row 61b - 2q - 2 + l in both sweeps; at step s the first sweep updates column s - l, the
    second sweep column s - l - 2 with operands taken from the first sweep's results of steps s - 2 (centre) and s - 1
    (right; lane l + 1's for down) and the coefficient cells of step s - 2 (their slot is refilled two steps late).
      first sweep:  lane 0 carrier, lane 1 ghost (row above at sweep k), lanes 2..62 real, lane 63 ghost (row below)
      second sweep: lane 0 ghost (row above at sweep k + 1), lanes 1..61 real, lane 62 not stored, lane 63 unused
    One store per step to D[q & 1][s + 1][b]: lanes 0..61 their second-sweep result, lane 62 its FIRST-sweep result;
    planes alternate per pair.  First-sweep operands of step s: lanes >= 2 <- previous pair's cell l - 2 at position
    s + 2 (pair 0: zeros); lane 1 <- cell 62 of the block above at position s + 63 (its lane 62's first-sweep result);
    lane 0 <- cell 61 of the block above at position s + 65 (its lane 61's second-sweep result).  An odd sweep count
    makes the last pair's second sweep the identity.  Before the loads of steps < e are issued:
        prog[q-1][b] >= min(NS, e + 1),  prog[q][b-1] >= min(NS, e + 64),  prog[q-2][b+1] >= min(NS, e - 62).
    Random scheduler at iteration granularity, loads taken when the kernel issues them, publications lagging."""
    h, w = phi.shape
    lay = layout_fused(h, w, n_sor, r)
    nb, ns, rt, qt, last = lay["nb"], lay["ns"], lay["rt"], lay["qt"], lay["last"]
    pairs = last + 1
    P = {n: to_skew(p, lay) for n, p in dict(phi=phi, xy=imdxy, a1=a1, a2=a2, b1=b1, b2=b2).items()}
    npos_d = ns + 2 * r + 72
    D = np.zeros((2, 2, npos_d, nb, LANES))  # [du|dv][parity][position][band][cell]
    D[:, :, :ns] = np.nan                    # only the tail positions need to be cleared (sor_solve): poison the rest
    prog = np.zeros((nb, pairs), dtype=np.int64)
    nalpha = -alpha
    lane = np.arange(LANES)
    real1 = (lane >= 2) & (lane <= 62)
    om1a = np.where(real1, 1 - omega, 1.0)
    om1b = np.where((lane == 0) | (lane == 63), 1.0, 1 - omega)
    coef_on = (lane != 0) & (lane != 63)     # lanes 0 / 63: (a, b) out of range -> 0
    n_iter = (ns + r - 1) // r
    rng = np.random.default_rng(seed)

    def covered(b, q, e):
        ok = True
        if q > 0:
            ok = ok and prog[b, q - 1] >= min(ns, e + 1)
        if b > 0:
            ok = ok and prog[b - 1, q] >= min(ns, e + 64)
        if use_dn2 and q > 1 and b + 1 < nb:
            ok = ok and prog[b + 1, q - 2] >= min(ns, max(0, e - 62))
        return ok

    def window(b, q):
        r0 = FROWS * b - 2 * q - 2  # image row of lane 0
        return r0 + qt, slice(r0 + rt, r0 + rt + LANES)

    def load_pd(c, b, q, s):
        """first-sweep right operand of step s (-1: the first centre; lanes >= 2 are then left of column 0: zero)"""
        v = np.zeros(LANES)
        if q > 0 and s >= 0:
            v[2:] = D[c, (q - 1) & 1, s + 2, b, :62]
        if b > 0:
            if s + 63 < npos_d:
                v[1] = D[c, q & 1, s + 63, b - 1, 62]
            if s + 65 < npos_d:
                v[0] = D[c, q & 1, s + 65, b - 1, 61]
        return v

    def load_coef(b, q, s):
        q0, rows = window(b, q)
        g = lambda n: P[n][q0 + s, rows].copy()
        z = lambda n: np.where(coef_on, P[n][q0 + s, rows], 0.0)
        return dict(phi=g("phi"), xy=g("xy"), a1=z("a1"), a2=z("a2"), b1=z("b1"), b2=z("b2"))

    class FTask:
        def __init__(self, b, q):
            self.b, self.q, self.i = b, q, -1
            z = np.zeros(LANES)
            self.s1 = dict(duL=z.copy(), dvL=z.copy(), phiL=z.copy(), duC=z.copy(), dvC=z.copy())
            self.s2 = dict(duL=z.copy(), dvL=z.copy(), phiL=z.copy(), duC=z.copy(), dvC=z.copy())
            self.coef = [None] * r
            self.pd = [None] * r

    def update(st, phiC, xy, a1_, a2_, b1_, b2_, duR, dvR, om1):
        duU, dvU, phiU = shift_up(st["duL"]), shift_up(st["dvL"]), shift_up(st["phiL"])
        duD, dvD = shift_down(duR), shift_down(dvR)
        s1 = st["phiL"] * st["duL"]
        s2 = st["phiL"] * st["dvL"]
        s1 = s1 + phiC * duR
        s2 = s2 + phiC * dvR
        s1 = s1 + phiU * duU
        s2 = s2 + phiU * dvU
        s1 = s1 + phiC * duD
        s2 = s2 + phiC * dvD
        s1 = s1 * nalpha
        s2 = s2 * nalpha
        s1 = s1 + xy * st["dvC"]
        duN = om1 * st["duC"] + a1_ * (b1_ - s1)
        s2 = s2 + xy * duN
        dvN = om1 * st["dvC"] + a2_ * (b2_ - s2)
        return duN, dvN

    pending = [FTask(b, q) for q in range(pairs) for b in range(nb)]
    while pending:
        ran = False
        for ti in rng.permutation(len(pending)):
            t = pending[ti]
            b, q = t.b, t.q
            identity2 = (n_sor & 1) == 1 and q == pairs - 1
            if t.i < 0:  # staged start-up: coefficients of the first R steps at once, unknowns once steps < R are covered
                if not covered(b, q, r):
                    continue
                t.s1["duC"], t.s1["dvC"] = load_pd(0, b, q, -1), load_pd(1, b, q, -1)
                for s in range(r):
                    t.coef[s] = load_coef(b, q, s)
                    t.pd[s] = (load_pd(0, b, q, s), load_pd(1, b, q, s))
                t.i = 0
                ran = True
                break
            i = t.i
            if not covered(b, q, (i + 2) * r):
                continue
            ran = True
            with np.errstate(invalid="ignore"):
                for s in range(i * r, (i + 1) * r):
                    c1 = t.coef[s % r]
                    duR, dvR = t.pd[s % r]
                    duR2, dvR2 = t.s1["duL"], t.s1["dvL"]  # first-sweep results of step s - 1
                    if s < 2:
                        duN2 = np.zeros(LANES)
                        dvN2 = np.zeros(LANES)
                    elif identity2:
                        duN2, dvN2 = t.s2["duC"], t.s2["dvC"]
                    else:
                        c2 = t.coef[(s - 2) % r]
                        duN2, dvN2 = update(t.s2, c2["phi"], c2["xy"], c2["a1"], c2["a2"], c2["b1"], c2["b2"], duR2, dvR2,
                                            om1b)
                        t.s2["phiL"] = c2["phi"]
                    a1m, a2m = np.where(real1, c1["a1"], 0.0), np.where(real1, c1["a2"], 0.0)
                    duN, dvN = update(t.s1, c1["phi"], c1["xy"], a1m, a2m, c1["b1"], c1["b2"], duR, dvR, om1a)
                    out_u = np.where(lane == 62, duN, duN2)
                    out_v = np.where(lane == 62, dvN, dvN2)
                    D[0, q & 1, s + 1, b, :63] = out_u[:63]  # lane 63 does not store
                    D[1, q & 1, s + 1, b, :63] = out_v[:63]
                    t.s2["duL"], t.s2["dvL"], t.s2["duC"], t.s2["dvC"] = duN2, dvN2, duR2, dvR2
                    t.s1["duL"], t.s1["dvL"], t.s1["phiL"] = duN, dvN, c1["phi"]
                    t.s1["duC"], t.s1["dvC"] = duR, dvR
                    t.pd[s % r] = (load_pd(0, b, q, s + r), load_pd(1, b, q, s + r))  # refilled at once
                    if s >= 2:
                        t.coef[(s - 2) % r] = load_coef(b, q, s - 2 + r)              # refilled two steps late
            prog[b, q] = min(ns, i * r + r // 2)
            t.i += 1
            if t.i == n_iter:
                prog[b, q] = ns
                pending.pop(ti)
            break
        assert ran, "deadlock in the task graph"
    # read-out: t = i + 2*last + 1, band t // 61, cell 1 + t % 61, position j + cell + 3, plane last & 1
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    tt = ii + 2 * last + 1
    bb, cc = tt // FROWS, 1 + tt % FROWS
    return D[0, last & 1, jj + cc + 3, bb, cc], D[1, last & 1, jj + cc + 3, bb, cc]
