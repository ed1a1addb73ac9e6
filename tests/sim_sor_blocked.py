"""CPU model of the LDS-tiled, temporally blocked red-black / Jacobi solver (csrc/sor.hip: k_sor_blocked and its host
side) -- the ghost-zone algebra, the launch plan and the ping-pong of the two (du, dv) plane pairs, restated in numpy with
the kernel's operation order, so that the design is checked bit for bit against the oracle WITHOUT a GPU:

  * a workgroup's region = its core tile grown by g cells (clamped to the plane; regions start on an even column);
  * half-sweep m of a launch is applied on the region shrunk by m + 1 cells from every edge that is not a plane border;
  * only the core tile is written back, into the OTHER pair of planes;
  * a solve of `units` half-sweeps (Jacobi: sweeps) is ceil(units / depth) launches of nearly equal depth, whole sweeps
    per launch in red-black mode; the first launch starts from zeros without reading any unknowns.
"""
import numpy as np

RW = 128


def _cell_update(w_l, w_u, pc, xy, a1, a2, b1, b2, own_u, own_v, L, R, U, D, nalpha, om1):
    """the kernel's `cell` lambda: terms left, right, up, down in the reference's order (src/OpticalFlow.cpp:468-504)"""
    s1 = w_l * L[0]
    s2 = w_l * L[1]
    s1 = s1 + pc * R[0]
    s2 = s2 + pc * R[1]
    s1 = s1 + w_u * U[0]
    s2 = s2 + w_u * U[1]
    s1 = s1 + pc * D[0]
    s2 = s2 + pc * D[1]
    s1 = s1 * nalpha
    s2 = s2 * nalpha
    s1 = s1 + xy * own_v
    nu = om1 * own_u + a1 * (b1 - s1)
    s2 = s2 + xy * nu
    nv = om1 * own_v + a2 * (b2 - s2)
    return nu, nv


def _launch(planes, su, sv, H, W, g, hs0, mode, RH, alpha, omega, out=None):
    """one launch that delivers the rectangle `out` = (x0, y0, x1, y1) (default: the whole plane; a tile grown by its
    remaining ghost depth in the multi-GPU path, csrc/tiles.hip): returns the new (du, dv) planes, NaN outside `out`"""
    phi, xy, a1, a2, b1, b2 = planes
    nalpha, om1 = -alpha, 1 - omega
    ox0_, oy0_, ox1_, oy1_ = out if out is not None else (0, 0, W, H)
    du, dv = np.full((H, W), np.nan), np.full((H, W), np.nan)
    if ox1_ <= ox0_ or oy1_ <= oy0_:
        return du, dv
    span_x = ox0_ == 0 and ox1_ == W and W <= RW
    span_y = oy0_ == 0 and oy1_ == H and H <= RH
    shift = 0 if span_x else ((ox0_ - g) & 1)
    cw = W if span_x else RW - 2 * g - 2 * shift
    ch = H if span_y else RH - 2 * g
    assert cw >= 1 and ch >= 1
    for cy0 in range(oy0_, oy1_, ch):
        for cx0 in range(ox0_, ox1_, cw):
            cx1, cy1 = min(cx0 + cw, ox1_), min(cy0 + ch, oy1_)
            rx0, rx1 = max(0, cx0 - g - shift), min(W, cx1 + g)
            ry0, ry1 = max(0, cy0 - g), min(H, cy1 + g)
            assert rx1 - rx0 <= RW and ry1 - ry0 <= RH and rx0 % 2 == 0
            rh, rw = ry1 - ry0, rx1 - rx0
            # zero-padded region copies (the LDS arrays with their pad ring; cells outside the plane stay zero)
            u = np.zeros((rh + 2, rw + 2))
            v = np.zeros((rh + 2, rw + 2))
            if su is not None:
                u[1:-1, 1:-1] = su[ry0:ry1, rx0:rx1]
                v[1:-1, 1:-1] = sv[ry0:ry1, rx0:rx1]
            reg = lambda p: p[ry0:ry1, rx0:rx1]
            pc = reg(phi)
            w_l = np.zeros((rh, rw))
            w_l[:, 1:] = pc[:, :-1]
            if rx0 > 0:
                w_l[:, 0] = phi[ry0:ry1, rx0 - 1]  # never used where it matters (that column is outside every frame)
            w_u = np.zeros((rh, rw))
            w_u[1:, :] = pc[:-1, :]
            if ry0 > 0:
                w_u[0, :] = phi[ry0 - 1, rx0:rx1]
            ii, jj = np.meshgrid(np.arange(ry0, ry1), np.arange(rx0, rx1), indexing="ij")
            colour = (ii + jj) & 1
            ox0, ox1, oy0, oy1 = int(rx0 > 0), int(rx1 < W), int(ry0 > 0), int(ry1 < H)
            for m in range(g):
                fx0, fx1 = ox0 * (m + 1), rw - ox1 * (m + 1)
                fy0, fy1 = oy0 * (m + 1), rh - oy1 * (m + 1)
                frame = np.zeros((rh, rw), dtype=bool)
                frame[fy0:fy1, fx0:fx1] = True
                if mode == 1:
                    frame &= colour == ((hs0 + m) & 1)
                cu, cv = u[1:-1, 1:-1], v[1:-1, 1:-1]
                nu, nv = _cell_update(w_l, w_u, pc, reg(xy), reg(a1), reg(a2), reg(b1), reg(b2), cu, cv,
                                      (u[1:-1, :-2], v[1:-1, :-2]), (u[1:-1, 2:], v[1:-1, 2:]),
                                      (u[:-2, 1:-1], v[:-2, 1:-1]), (u[2:, 1:-1], v[2:, 1:-1]), nalpha, om1)
                cu[frame] = nu[frame]  # in red-black mode only other-colour cells were read: in-place is exact
                cv[frame] = nv[frame]
            du[cy0:cy1, cx0:cx1] = u[1 + cy0 - ry0:1 + cy1 - ry0, 1 + cx0 - rx0:1 + cx1 - rx0]
            dv[cy0:cy1, cx0:cx1] = v[1 + cy0 - ry0:1 + cy1 - ry0, 1 + cx0 - rx0:1 + cx1 - rx0]
    assert not np.isnan(du[oy0_:oy1_, ox0_:ox1_]).any()
    return du, dv


def plan(units, mode, depth, one_block):
    """(half-)sweeps per launch: sor.hip blocked_plan()"""
    q = 2 if (mode == 1 and units % 2 == 0) else 1
    n_units = units // q
    d = n_units if one_block else max(1, depth // q)
    n_launch = -(-n_units // d)
    base, rem = divmod(n_units, n_launch)
    return [q * (base + (1 if l < rem else 0)) for l in range(n_launch)]


def solve(planes, n_sor, mode, alpha=0.012, omega=1.8, RH=48, depth=10):
    """mode 1 = red-black (in-place two-colour sweeps), 2 = Jacobi; from du = dv = 0"""
    H, W = planes[0].shape
    units = 2 * n_sor if mode == 1 else n_sor
    gs = plan(units, mode, depth, W <= RW and H <= RH)
    su = sv = None
    done = 0
    for g in gs:
        su, sv = _launch(planes, su, sv, H, W, g, done, mode, RH, alpha, omega)
        done += g
    assert done == units
    return su, sv
