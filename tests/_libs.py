"""ctypes front-ends used by the tests only.

`OracleLib`  -> oracle/libpapof_oracle.so  (our CPU restatement; travels to the GPU box)
`RefLib`     -> oracle/_ref/libpapof_ref.so (the untouched reference, build container only)

Both expose the same method names on reference-layout (HWC, float64) numpy arrays so a golden
case can be evaluated against either.  Nothing in the product package imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libpapof_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libpapof_ref.so")

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)
c_int, c_double = ctypes.c_int, ctypes.c_double


def _p(a):
    return a.ctypes.data_as(_D)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OrcParams(ctypes.Structure):
    _fields_ = [("alpha", c_double), ("ratio", c_double), ("n_outer", c_int), ("n_outer_per_level", c_int),
                ("n_inner", c_int), ("n_sor", c_int), ("n_sor_per_level", c_int), ("omega", c_double),
                ("sor_mode", c_int), ("interpolation", c_int), ("noise_model", c_int)]


def build_oracle():
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "papof_oracle.c"))):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def _split_levels(data, dims, n, c):
    out, off = [], 0
    for i in range(n):
        lw, lh = int(dims[2 * i]), int(dims[2 * i + 1])
        out.append(data[off:off + lw * lh * c].reshape(lh, lw, c).copy())
        off += lw * lh * c
    return out


def gm_default(c):
    """GaussianMixture::reset (src/NoiseModel.h:97-107): alpha .95, sigma .05, beta .5 and their squares, per channel"""
    return np.concatenate([np.full(c, 0.95), np.full(c, 0.05), np.full(c, 0.5), np.full(c, 0.05) ** 2,
                           np.full(c, 0.5) ** 2])


class _OracleBranches:
    """The reference's non-default branches on the oracle; mixed into OracleLib."""

    def pyramid_minwidth(self, im, ratio, min_width):
        im = _c(im)
        h, w, c = im.shape
        self.L.orc_pyramid_levels_for_min_width.argtypes = [c_int, c_double, c_int]
        n = self.L.orc_pyramid_levels_for_min_width(w, ratio, min_width)
        return self.pyramid(im, ratio, n)

    def coarse2fine_flow_opts(self, im1, im2, levels, interpolation, noise_model):
        p = self.default_params()
        p.interpolation, p.noise_model = interpolation, noise_model
        return self.coarse2fine_flow(im1, im2, levels, p)[:3]

    def smoothflow_sor_opts(self, im1, im2, warp, u, v, alpha, n_outer, n_inner, n_sor, interpolation, noise_model):
        im1, im2 = _c(im1), _c(im2)
        warp, u, v = _c(warp).copy(), _c(u).copy(), _c(v).copy()
        h, w, c = im1.shape
        lp = np.full(max(c, 8), 0.02)
        gm = gm_default(c)
        self.L.orc_smoothflow_sor_ex.argtypes = [_D, _D, _D, _D, _D, c_int, c_int, c_int, c_double, c_int, c_int, c_int,
                                                 c_double, c_int, _D, _D, c_int, _D]
        self.L.orc_smoothflow_sor_ex(_p(im1), _p(im2), _p(warp), _p(u), _p(v), h, w, c, alpha, n_outer, n_inner, n_sor,
                                     1.8, 0, _p(lp), None, interpolation, _p(gm) if noise_model else None)
        return warp, u, v, gm

    def est_gaussian_mixture(self, im1, im2, gm=None):
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        gm = gm_default(c) if gm is None else _c(gm).copy()
        self.L.orc_est_gaussian_mixture.argtypes = [_D, _D, ctypes.c_long, c_int, _D, c_double]
        self.L.orc_est_gaussian_mixture(_p(im1), _p(im2), h * w, c, _p(gm), 0.9)
        return gm

    def bicubic_warp_noclamp(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1), _c(im2), _c(vx), _c(vy)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        self.L.orc_bicubic_warp_noclamp.argtypes = [_D, _D, _D, _D, c_int, c_int, c_int, _D]
        self.L.orc_bicubic_warp_noclamp(_p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out))
        return out


class OracleLib(_OracleBranches):
    name = "oracle"

    def __init__(self):
        build_oracle()
        L = self.L = ctypes.CDLL(ORACLE_SO)
        L.orc_pyramid.restype = ctypes.c_long
        L.orc_pyramid.argtypes = [_D, c_int, c_int, c_int, c_double, c_int, _I, _D]
        L.orc_gaussian_smoothing.argtypes = [_D, _D, c_int, c_int, c_int, c_double, c_int]
        L.orc_resize_ratio.argtypes = [_D, _D, c_int, c_int, c_int, c_double]
        L.orc_resize_wh.argtypes = [_D, _D, c_int, c_int, c_int, c_int, c_int]
        L.orc_im2feature.argtypes = [_D, c_int, c_int, c_int, _D]
        L.orc_warpFL.argtypes = [_D, _D, _D, _D, c_int, c_int, c_int, _D]
        L.orc_getDxs.argtypes = [_D, _D, c_int, c_int, c_int, _D, _D, _D]
        L.orc_laplacian.argtypes = [_D, _D, c_int, c_int, _D]
        L.orc_bicubic_warp.argtypes = [_D, _D, _D, _D, c_int, c_int, c_int, _D]
        L.orc_linear_system.argtypes = [_D, _D, _D, _D, _D, _D, _D, c_int, c_int, c_int, c_double, _D] + [_D] * 6
        L.orc_sor.argtypes = [_D] * 8 + [c_int, c_int, c_double, c_double, c_int, c_int]
        L.orc_smoothflow_sor.argtypes = [_D, _D, _D, _D, _D, c_int, c_int, c_int, c_double, c_int, c_int, c_int,
                                         c_double, c_int, _D, _D]
        L.orc_coarse2fine_flow.argtypes = [_D, _D, c_int, c_int, c_int, c_int, ctypes.POINTER(OrcParams), _D, _D,
                                           _D, _D]
        L.orc_default_params.argtypes = [ctypes.POINTER(OrcParams)]

    def default_params(self):
        p = OrcParams()
        self.L.orc_default_params(ctypes.byref(p))
        return p

    def pyramid(self, im, ratio, levels):
        im = _c(im)
        h, w, c = im.shape
        dims = np.zeros(2 * levels, dtype=np.int32)
        total = self.L.orc_pyramid(_p(im), h, w, c, ratio, levels, dims.ctypes.data_as(_I), None)
        data = np.zeros(total)
        self.L.orc_pyramid(_p(im), h, w, c, ratio, levels, dims.ctypes.data_as(_I), _p(data))
        out, off = [], 0
        for i in range(levels):
            lw, lh = int(dims[2 * i]), int(dims[2 * i + 1])
            out.append(data[off:off + lw * lh * c].reshape(lh, lw, c).copy())
            off += lw * lh * c
        return out

    def gaussian_smoothing(self, im, sigma, fsize):
        im = _c(im)
        h, w, c = im.shape
        out = np.zeros_like(im)
        self.L.orc_gaussian_smoothing(_p(im), _p(out), w, h, c, sigma, fsize)
        return out

    def resize_ratio(self, im, ratio):
        im = _c(im)
        h, w, c = im.shape
        dw, dh = int(float(w) * ratio), int(float(h) * ratio)
        out = np.zeros((dh, dw, c))
        self.L.orc_resize_ratio(_p(im), _p(out), w, h, c, ratio)
        return out

    def resize_wh(self, im, dw, dh):
        im = _c(im)
        h, w, c = im.shape
        out = np.zeros((dh, dw, c))
        self.L.orc_resize_wh(_p(im), _p(out), w, h, c, dw, dh)
        return out

    def im2feature(self, im):
        im = _c(im)
        h, w, c = im.shape
        fc = self.L.orc_im2feature(None, h, w, c, None)
        out = np.zeros((h, w, fc))
        self.L.orc_im2feature(_p(im), h, w, c, _p(out))
        return out

    def warpFL(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1), _c(im2), _c(vx), _c(vy)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        self.L.orc_warpFL(_p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out))
        return out

    def getDxs(self, im1, im2):
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        dx, dy, dt = np.zeros_like(im1), np.zeros_like(im1), np.zeros_like(im1)
        self.L.orc_getDxs(_p(im1), _p(im2), h, w, c, _p(dx), _p(dy), _p(dt))
        return dx, dy, dt

    def laplacian(self, x, weight):
        x, weight = _c(x), _c(weight)
        h, w = x.shape
        out = np.zeros_like(x)
        self.L.orc_laplacian(_p(x), _p(weight), h, w, _p(out))
        return out

    def bicubic_warp(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1), _c(im2), _c(vx), _c(vy)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        self.L.orc_bicubic_warp(_p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out))
        return out

    def flow_quantize16(self, vx, vy):
        vx, vy = _c(vx), _c(vy)
        h, w = vx.shape
        q = np.zeros((h, w, 2), dtype=np.uint16)
        self.L.orc_flow_quantize16.argtypes = [_D, _D, c_int, c_int, ctypes.c_void_p]
        self.L.orc_flow_quantize16(_p(vx), _p(vy), h, w, q.ctypes.data_as(ctypes.c_void_p))
        return q

    def flow_dequantize16(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint16)
        h, w, _ = q.shape
        vx, vy = np.zeros((h, w)), np.zeros((h, w))
        self.L.orc_flow_dequantize16.argtypes = [ctypes.c_void_p, c_int, c_int, _D, _D]
        self.L.orc_flow_dequantize16(q.ctypes.data_as(ctypes.c_void_p), h, w, _p(vx), _p(vy))
        return vx, vy

    def linear_system(self, imdx, imdy, imdt, u, v, du=None, dv=None, alpha=0.012, lappara=None):
        imdx, imdy, imdt, u, v = _c(imdx), _c(imdy), _c(imdt), _c(u), _c(v)
        h, w, c = imdx.shape
        lp = _c(np.full(c, 0.02) if lappara is None else lappara)
        outs = [np.zeros((h, w)) for _ in range(6)]
        dup = _p(_c(du)) if du is not None else None
        dvp = _p(_c(dv)) if dv is not None else None
        self.L.orc_linear_system(_p(imdx), _p(imdy), _p(imdt), _p(u), _p(v), dup, dvp, h, w, c, alpha, _p(lp),
                                 *[_p(o) for o in outs])
        return outs  # phi, imdxy, imdx2, imdy2, imdtdx(rhs1), imdtdy(rhs2)

    def sor(self, phi, imdxy, imdx2, imdy2, rhs1, rhs2, n_sor, alpha=0.012, omega=1.8, mode=0, du=None, dv=None):
        arrs = [_c(a) for a in (phi, imdxy, imdx2, imdy2, rhs1, rhs2)]
        h, w = arrs[0].shape
        du = np.zeros((h, w)) if du is None else _c(du).copy()
        dv = np.zeros((h, w)) if dv is None else _c(dv).copy()
        self.L.orc_sor(*[_p(a) for a in arrs], _p(du), _p(dv), h, w, alpha, omega, n_sor, mode)
        return du, dv

    def smoothflow_sor(self, im1, im2, warp, u, v, alpha, n_outer, n_inner, n_sor, omega=1.8, mode=0):
        im1, im2 = _c(im1), _c(im2)
        warp, u, v = _c(warp).copy(), _c(u).copy(), _c(v).copy()
        h, w, c = im1.shape
        lp = np.full(max(c, 8), 0.02)
        self.L.orc_smoothflow_sor(_p(im1), _p(im2), _p(warp), _p(u), _p(v), h, w, c, alpha, n_outer, n_inner,
                                  n_sor, omega, mode, _p(lp), None)
        return warp, u, v

    def coarse2fine_flow(self, im1, im2, levels, params=None):
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        vx, vy, wi, t = np.zeros((h, w)), np.zeros((h, w)), np.zeros((h, w, c)), np.zeros(10)
        pp = ctypes.byref(params) if params is not None else None
        rc = self.L.orc_coarse2fine_flow(_p(im1), _p(im2), h, w, c, levels, pp, _p(vx), _p(vy), _p(wi), _p(t))
        if rc != 0:
            raise ValueError("orc_coarse2fine_flow rc=%d" % rc)
        return vx, vy, wi, t

    def coarse2fine_flow_sched(self, im1, im2, levels, alpha, ratio, n_outer, outer_step, n_inner, n_sor, sor_step,
                               mode=0, omega=1.8):
        p = self.default_params()
        p.alpha, p.ratio, p.n_outer, p.n_outer_per_level = alpha, ratio, n_outer, outer_step
        p.n_inner, p.n_sor, p.n_sor_per_level, p.sor_mode, p.omega = n_inner, n_sor, sor_step, mode, omega
        return self.coarse2fine_flow(im1, im2, levels, p)[:3]


class RefLib:
    """The untouched reference (only where oracle/_ref/libpapof_ref.so has been built)."""
    name = "reference"

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        L = self.L = ctypes.CDLL(REF_SO)
        L.ref_coarse2fine_flow.argtypes = [_D, _D, c_int, c_int, c_int, c_int, _D, _D, _D, _D]
        L.ref_coarse2fine_flow_sched.argtypes = [_D, _D, c_int, c_int, c_int, c_int, c_double, c_double, c_int,
                                                 c_int, c_int, c_int, c_int, _D, _D, _D]
        L.ref_pyramid.argtypes = [_D, c_int, c_int, c_int, c_double, c_int, _I, _D]
        L.ref_gaussian_smoothing.argtypes = [_D, c_int, c_int, c_int, c_double, c_int, _D]
        L.ref_resize_ratio.argtypes = [_D, c_int, c_int, c_int, c_double, _I, _I, _D]
        L.ref_resize_wh.argtypes = [_D, c_int, c_int, c_int, c_int, c_int, _D]
        L.ref_im2feature.argtypes = [_D, c_int, c_int, c_int, _D]
        L.ref_warpFL.argtypes = [_D, _D, _D, _D, c_int, c_int, c_int, _D]
        L.ref_getDxs.argtypes = [_D, _D, c_int, c_int, c_int, _D, _D, _D]
        L.ref_laplacian.argtypes = [_D, _D, c_int, c_int, _D]
        L.ref_smoothflow_sor.argtypes = [_D, _D, _D, _D, _D, c_int, c_int, c_int, c_double, c_int, c_int, c_int]
        L.ref_bicubic_warp.argtypes = [_D, _D, _D, _D, c_int, c_int, c_int, _D]

    def pyramid(self, im, ratio, levels):
        im = _c(im)
        h, w, c = im.shape
        dims = np.zeros(2 * levels, dtype=np.int32)
        self.L.ref_pyramid(_p(im), h, w, c, ratio, levels, dims.ctypes.data_as(_I), None)
        total = sum(int(dims[2 * i]) * int(dims[2 * i + 1]) * c for i in range(levels))
        data = np.zeros(total)
        self.L.ref_pyramid(_p(im), h, w, c, ratio, levels, dims.ctypes.data_as(_I), _p(data))
        out, off = [], 0
        for i in range(levels):
            lw, lh = int(dims[2 * i]), int(dims[2 * i + 1])
            out.append(data[off:off + lw * lh * c].reshape(lh, lw, c).copy())
            off += lw * lh * c
        return out

    def gaussian_smoothing(self, im, sigma, fsize):
        im = _c(im)
        h, w, c = im.shape
        out = np.zeros_like(im)
        self.L.ref_gaussian_smoothing(_p(im), h, w, c, sigma, fsize, _p(out))
        return out

    def resize_ratio(self, im, ratio):
        im = _c(im)
        h, w, c = im.shape
        dw, dh = c_int(0), c_int(0)
        self.L.ref_resize_ratio(_p(im), h, w, c, ratio, ctypes.byref(dw), ctypes.byref(dh), None)
        out = np.zeros((dh.value, dw.value, c))
        self.L.ref_resize_ratio(_p(im), h, w, c, ratio, ctypes.byref(dw), ctypes.byref(dh), _p(out))
        return out

    def resize_wh(self, im, dw, dh):
        im = _c(im)
        h, w, c = im.shape
        out = np.zeros((dh, dw, c))
        self.L.ref_resize_wh(_p(im), h, w, c, dw, dh, _p(out))
        return out

    def im2feature(self, im):
        im = _c(im)
        h, w, c = im.shape
        fc = self.L.ref_im2feature(_p(im), h, w, c, None)
        out = np.zeros((h, w, fc))
        self.L.ref_im2feature(_p(im), h, w, c, _p(out))
        return out

    def warpFL(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1), _c(im2), _c(vx), _c(vy)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        self.L.ref_warpFL(_p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out))
        return out

    def getDxs(self, im1, im2):
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        dx, dy, dt = np.zeros_like(im1), np.zeros_like(im1), np.zeros_like(im1)
        self.L.ref_getDxs(_p(im1), _p(im2), h, w, c, _p(dx), _p(dy), _p(dt))
        return dx, dy, dt

    def laplacian(self, x, weight):
        x, weight = _c(x), _c(weight)
        h, w = x.shape
        out = np.zeros_like(x)
        self.L.ref_laplacian(_p(x), _p(weight), h, w, _p(out))
        return out

    def bicubic_warp(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1), _c(im2), _c(vx), _c(vy)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        self.L.ref_bicubic_warp(_p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out))
        return out

    def flow_file16(self, vx, vy):
        """bytes of the file the reference's SaveOpticalFlow writes (header: 16-byte type name, 3 ints, 1 bool)"""
        import tempfile
        vx, vy = _c(vx), _c(vy)
        h, w = vx.shape
        self.L.ref_flow_save16.argtypes = [_D, _D, c_int, c_int, ctypes.c_char_p]
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "flow.bin")
            assert self.L.ref_flow_save16(_p(vx), _p(vy), h, w, path.encode()) == 0
            return open(path, "rb").read()

    def flow_quantize16(self, vx, vy):
        h, w = np.shape(vx)
        raw = self.flow_file16(vx, vy)
        return np.frombuffer(raw[29:], dtype=np.uint16).reshape(h, w, 2).copy()

    def flow_dequantize16(self, q):
        """through the reference's LoadOpticalFlow, from a file with the reference's own header layout"""
        import struct
        import tempfile
        q = np.ascontiguousarray(q, dtype=np.uint16)
        h, w, _ = q.shape
        vx, vy = np.zeros((h, w)), np.zeros((h, w))
        self.L.ref_flow_load16.argtypes = [ctypes.c_char_p, c_int, c_int, _D, _D]
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "flow.bin")
            with open(path, "wb") as f:
                f.write(b"t".ljust(16, b"\0") + struct.pack("<iii?", w, h, 2, False) + q.tobytes())
            assert self.L.ref_flow_load16(path.encode(), h, w, _p(vx), _p(vy)) == 0
        return vx, vy

    def smoothflow_sor(self, im1, im2, warp, u, v, alpha, n_outer, n_inner, n_sor, omega=1.8, mode=0):
        assert omega == 1.8 and mode == 0, "the reference hard-codes omega and the sweep order"
        im1, im2 = _c(im1), _c(im2)
        warp, u, v = _c(warp).copy(), _c(u).copy(), _c(v).copy()
        h, w, c = im1.shape
        self.L.ref_smoothflow_sor(_p(im1), _p(im2), _p(warp), _p(u), _p(v), h, w, c, alpha, n_outer, n_inner, n_sor)
        return warp, u, v

    def coarse2fine_flow(self, im1, im2, levels, params=None):
        assert params is None
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        vx, vy, wi, t = np.zeros((h, w)), np.zeros((h, w)), np.zeros((h, w, c)), np.zeros(10)
        self.L.ref_coarse2fine_flow(_p(im1), _p(im2), h, w, c, levels, _p(vx), _p(vy), _p(wi), _p(t))
        return vx, vy, wi, t

    # ---- the non-default branches, selected through the reference's public statics (oracle/ref_driver.cpp) ----
    def pyramid_minwidth(self, im, ratio, min_width):
        im = _c(im)
        h, w, c = im.shape
        dims = np.zeros(128, dtype=np.int32)
        self.L.ref_pyramid_minwidth.argtypes = [_D, c_int, c_int, c_int, c_double, c_int, _I, _D]
        n = self.L.ref_pyramid_minwidth(_p(im), h, w, c, ratio, min_width, dims.ctypes.data_as(_I), None)
        total = sum(int(dims[2 * i]) * int(dims[2 * i + 1]) * c for i in range(n))
        data = np.zeros(total)
        self.L.ref_pyramid_minwidth(_p(im), h, w, c, ratio, min_width, dims.ctypes.data_as(_I), _p(data))
        return _split_levels(data, dims, n, c)

    def coarse2fine_flow_opts(self, im1, im2, levels, interpolation, noise_model):
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        vx, vy, wi = np.zeros((h, w)), np.zeros((h, w)), np.zeros((h, w, c))
        self.L.ref_coarse2fine_flow_opts.argtypes = [_D, _D, c_int, c_int, c_int, c_int, c_int, c_int, _D, _D, _D]
        self.L.ref_coarse2fine_flow_opts(_p(im1), _p(im2), h, w, c, levels, interpolation, noise_model, _p(vx), _p(vy),
                                         _p(wi))
        return vx, vy, wi

    def smoothflow_sor_opts(self, im1, im2, warp, u, v, alpha, n_outer, n_inner, n_sor, interpolation, noise_model):
        im1, im2 = _c(im1), _c(im2)
        warp, u, v = _c(warp).copy(), _c(u).copy(), _c(v).copy()
        h, w, c = im1.shape
        gm = gm_default(c)
        self.L.ref_smoothflow_sor_opts.argtypes = [_D, _D, _D, _D, _D, c_int, c_int, c_int, c_double, c_int, c_int, c_int,
                                                   c_int, c_int, _D]
        self.L.ref_smoothflow_sor_opts(_p(im1), _p(im2), _p(warp), _p(u), _p(v), h, w, c, alpha, n_outer, n_inner, n_sor,
                                       interpolation, noise_model, _p(gm))
        return warp, u, v, gm

    def est_gaussian_mixture(self, im1, im2, gm=None):
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        gm = gm_default(c) if gm is None else _c(gm).copy()
        self.L.ref_est_gaussian_mixture.argtypes = [_D, _D, c_int, c_int, c_int, _D]
        self.L.ref_est_gaussian_mixture(_p(im1), _p(im2), h, w, c, _p(gm))
        return gm

    def bicubic_warp_noclamp(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1), _c(im2), _c(vx), _c(vy)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        self.L.ref_bicubic_warp_noclamp.argtypes = [_D, _D, _D, _D, c_int, c_int, c_int, _D]
        self.L.ref_bicubic_warp_noclamp(_p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out))
        return out

    def coarse2fine_flow_sched(self, im1, im2, levels, alpha, ratio, n_outer, outer_step, n_inner, n_sor, sor_step,
                               mode=0, omega=1.8):
        assert omega == 1.8 and mode == 0
        im1, im2 = _c(im1), _c(im2)
        h, w, c = im1.shape
        vx, vy, wi = np.zeros((h, w)), np.zeros((h, w)), np.zeros((h, w, c))
        self.L.ref_coarse2fine_flow_sched(_p(im1), _p(im2), h, w, c, levels, alpha, ratio, n_outer, outer_step,
                                          n_inner, n_sor, sor_step, _p(vx), _p(vy), _p(wi))
        return vx, vy, wi
