"""ctypes binding of the C ABI in include/papof.h (libpapof.so, HIP / gfx950).

Plumbing only: numpy arrays in the reference's layout (HWC float64) go in and come out; all compute
happens in the HIP library.  There is no CPU fallback -- if the library is missing or there is no
usable gfx950 device every call raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpapof.so")

SOR_EXACT, SOR_REDBLACK, SOR_JACOBI = 0, 1, 2
N_TIMERS = 10

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)
c_int, c_double, c_void_p = ctypes.c_int, ctypes.c_double, ctypes.c_void_p


class Params(ctypes.Structure):
    """struct papof_params (include/papof.h); defaults = the reference's hard-coded constants."""
    _fields_ = [("alpha", c_double), ("ratio", c_double), ("n_outer", c_int), ("n_outer_per_level", c_int),
                ("n_inner", c_int), ("n_sor", c_int), ("n_sor_per_level", c_int), ("omega", c_double),
                ("sor_mode", c_int), ("phase_timing", c_int), ("interpolation", c_int), ("noise_model", c_int)]


class PapofError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__("%s failed: %s (%d)%s" % (what, detail[0], code, (": " + detail[1]) if detail[1] else ""))
        self.code = code


_lib = None

# every symbol include/papof.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "papof_version", "papof_default_params", "papof_strerror", "papof_last_error", "papof_timing_key",
    "papof_device_count", "papof_create", "papof_destroy", "papof_coarse2fine_flow", "papof_flow",
    "papof_flow_device", "papof_dev_alloc", "papof_dev_free", "papof_dev_upload", "papof_dev_download",
    "papof_stream", "papof_stage_pyramid", "papof_stage_gaussian", "papof_stage_resize_ratio",
    "papof_stage_resize_wh", "papof_stage_im2feature", "papof_stage_warpFL", "papof_stage_getDxs",
    "papof_stage_linear_system", "papof_stage_laplacian", "papof_stage_sor", "papof_stage_smoothflow",
    "papof_stage_bicubic_warp", "papof_bench_sor", "papof_flow_u8", "papof_flow_device_u8", "papof_seq_reset",
    "papof_seq_push", "papof_seq_push_u8", "papof_seq_push_device", "papof_tiles_grid", "papof_tiles_rect",
    "papof_tiles_halo_message", "papof_tiles_unique_id", "papof_tiles_create", "papof_tiles_create_local",
    "papof_tiles_flow_device", "papof_tiles_stats", "papof_tiles_destroy", "papof_flow_quantize16",
    "papof_flow_dequantize16", "papof_flow_to_bgr", "papof_set_graph_mode", "papof_set_stream_overlap", "papof_sor_plan", "papof_last_sor_stats", "papof_strip_plan", "papof_test_sor_strips",
    "papof_pyramid_levels_for_min_width", "papof_stage_smoothflow_ex", "papof_stage_est_gaussian_mixture",
    "papof_stage_bicubic_warp_ex", "papof_tiles_comm_info", "papof_host_alloc", "papof_host_free",
    "papof_last_sor_solves", "papof_bands_plan", "papof_lap_guard_stats", "papof_last_host_times",
    "papof_flow_batch", "papof_flow_batch_u8",
]


def load():
    """dlopen libpapof.so (raises OSError with a build hint if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C papteam_opticalflow_amd/csrc` (needs hipcc)" % LIB_PATH)
    # Several calls in flight (flow_collection, one handle + two streams each) need more than the HIP runtime's default
    # of 4 hardware queues, or streams share queues and serialise: 24 pairs of 1080p, 4 in flight, 153 -> 210 Mpix/s.
    # Only effective if the runtime has not been initialised yet in this process; an explicit setting wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    L = ctypes.CDLL(os.environ.get("PAPOF_LIB") or LIB_PATH)  # PAPOF_LIB: another build of the same ABI (A/B measurements)
    L.papof_strerror.restype = ctypes.c_char_p
    L.papof_strerror.argtypes = [c_int]
    L.papof_last_error.restype = ctypes.c_char_p
    L.papof_timing_key.restype = ctypes.c_char_p
    L.papof_timing_key.argtypes = [c_int]
    L.papof_default_params.argtypes = [ctypes.POINTER(Params)]
    L.papof_default_params.restype = None
    L.papof_create.argtypes = [c_int, ctypes.POINTER(c_void_p)]
    L.papof_destroy.argtypes = [c_void_p]
    L.papof_destroy.restype = None
    L.papof_stream.argtypes = [c_void_p]
    L.papof_stream.restype = c_void_p
    PP = ctypes.POINTER(Params)
    L.papof_coarse2fine_flow.argtypes = [_D, _D, c_int, c_int, c_int, c_int, PP, _D, _D, _D, _D]
    L.papof_flow.argtypes = [c_void_p, _D, _D, c_int, c_int, c_int, c_int, PP, _D, _D, _D, _D]
    L.papof_flow_device.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, PP, c_void_p, c_void_p,
                                    c_void_p, _D]
    _B = ctypes.POINTER(ctypes.c_ubyte)
    L.papof_flow_u8.argtypes = [c_void_p, _B, _B, c_int, c_int, c_int, c_int, PP, _D, _D, _D, _D]
    L.papof_flow_device_u8.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, PP, c_void_p,
                                       c_void_p, c_void_p, _D]
    L.papof_seq_reset.argtypes = [c_void_p]
    L.papof_seq_push.argtypes = [c_void_p, _D, c_int, c_int, c_int, c_int, PP, _D, _D, _D, _D, _I]
    L.papof_seq_push_u8.argtypes = [c_void_p, _B, c_int, c_int, c_int, c_int, PP, _D, _D, _D, _D, _I]
    L.papof_seq_push_device.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, PP, c_void_p,
                                        c_void_p, c_void_p, _D, _I]
    L.papof_tiles_grid.argtypes = [c_int, _I, _I]
    L.papof_tiles_rect.argtypes = [c_int, c_int, c_int, c_int, c_int, _I]
    L.papof_tiles_halo_message.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _I]
    L.papof_bands_plan.argtypes = [c_int, c_int, c_int, c_int, c_int, _I]
    L.papof_tiles_unique_id.argtypes = [ctypes.c_char_p]
    L.papof_tiles_create.argtypes = [c_void_p, ctypes.c_char_p, c_int, c_int, c_int, c_int, c_int,
                                     ctypes.POINTER(c_void_p)]
    L.papof_tiles_create_local.argtypes = [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int,
                                           ctypes.POINTER(c_void_p)]
    L.papof_tiles_flow_device.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, PP, c_void_p,
                                          c_void_p, c_void_p, _D]
    L.papof_tiles_stats.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_size_t)]
    L.papof_tiles_destroy.argtypes = [c_void_p]
    L.papof_tiles_destroy.restype = None
    L.papof_flow_quantize16.argtypes = [c_void_p, _D, _D, c_int, c_int, c_void_p]
    L.papof_flow_dequantize16.argtypes = [c_void_p, c_void_p, c_int, c_int, _D, _D]
    L.papof_flow_to_bgr.argtypes = [c_void_p, _D, _D, c_int, c_int, c_void_p]
    L.papof_set_graph_mode.argtypes = [c_void_p, c_int]
    L.papof_set_stream_overlap.argtypes = [c_void_p, c_int]
    L.papof_dev_alloc.argtypes = [c_void_p, ctypes.c_size_t, ctypes.POINTER(c_void_p)]
    L.papof_dev_free.argtypes = [c_void_p, c_void_p]
    L.papof_dev_upload.argtypes = [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]
    L.papof_dev_download.argtypes = [c_void_p, c_void_p, c_void_p, ctypes.c_size_t]
    L.papof_stage_pyramid.argtypes = [c_void_p, _D, c_int, c_int, c_int, c_double, c_int, _I, _D,
                                      ctypes.POINTER(ctypes.c_long)]
    L.papof_stage_gaussian.argtypes = [c_void_p, _D, c_int, c_int, c_int, c_double, c_int, _D]
    L.papof_stage_resize_ratio.argtypes = [c_void_p, _D, c_int, c_int, c_int, c_double, _D]
    L.papof_stage_resize_wh.argtypes = [c_void_p, _D, c_int, c_int, c_int, c_int, c_int, _D]
    L.papof_stage_im2feature.argtypes = [c_void_p, _D, c_int, c_int, c_int, _D, _I]
    L.papof_stage_warpFL.argtypes = [c_void_p, _D, _D, _D, _D, c_int, c_int, c_int, _D]
    L.papof_stage_getDxs.argtypes = [c_void_p, _D, _D, c_int, c_int, c_int, _D, _D, _D]
    L.papof_stage_linear_system.argtypes = [c_void_p, _D, _D, _D, _D, c_int, c_int, c_int, c_double] + [_D] * 6
    L.papof_stage_laplacian.argtypes = [c_void_p, _D, _D, c_int, c_int, _D]
    L.papof_stage_sor.argtypes = [c_void_p] + [_D] * 6 + [c_int, c_int, c_double, c_double, c_int, c_int, _D, _D]
    L.papof_stage_smoothflow.argtypes = [c_void_p, _D, _D, _D, _D, _D, c_int, c_int, c_int, c_double, c_int, c_int,
                                         c_int, c_double, c_int]
    L.papof_stage_bicubic_warp.argtypes = [c_void_p, _D, _D, _D, _D, c_int, c_int, c_int, _D]
    L.papof_bench_sor.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint, _D]
    L.papof_stage_smoothflow_ex.argtypes = [c_void_p, _D, _D, _D, _D, _D, c_int, c_int, c_int, c_double, c_int, c_int,
                                            c_int, c_double, c_int, c_int, c_int, _D]
    L.papof_stage_est_gaussian_mixture.argtypes = [c_void_p, _D, _D, c_int, c_int, c_int, _D]
    L.papof_stage_bicubic_warp_ex.argtypes = [c_void_p, _D, _D, _D, _D, c_int, c_int, c_int, c_int, _D]
    L.papof_pyramid_levels_for_min_width.argtypes = [c_int, c_double, c_int, ctypes.POINTER(c_int)]
    L.papof_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(c_void_p)]
    L.papof_host_free.argtypes = [c_void_p]
    L.papof_sor_plan.argtypes = [c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    L.papof_last_sor_stats.argtypes = [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double)]
    L.papof_last_sor_solves.argtypes = [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                        ctypes.POINTER(ctypes.c_double)]
    L.papof_lap_guard_stats.argtypes = [c_void_p, ctypes.POINTER(c_int)]
    L.papof_last_host_times.argtypes = [c_void_p, ctypes.POINTER(c_double)]
    for fn in (L.papof_flow_batch, L.papof_flow_batch_u8):
        fn.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, _D]
        fn.restype = c_int
    L.papof_test_sor_strips.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(c_int)]
    L.papof_strip_plan.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int),
                                   ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                   ctypes.POINTER(c_int)]
    _lib = L
    return L


# ---- result arrays in page-locked memory, recycled when the caller drops them (include/papof.h: papof_host_alloc).
# The accounting (size classes, a budget that counts idle AND live blocks, LRU eviction, atexit drain) is pinned_pool.py.
# NOTE for callers that fork(): the ROCm runtime maps page-locked host memory so that a forked CHILD does not inherit it
# (arrays returned by coarse2fine_flow would be unmapped in multiprocessing workers started by fork after the call).
# Copy the results (np.array(vx)) before forking, use the spawn start method, or set PAPOF_PINNED_OUT=0.
from . import pinned_pool as _pinned_pool_mod

_pool = None


def _pinned_alloc(nbytes):
    p = c_void_p()
    if load().papof_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(p)) != 0 or not p.value:
        return 0
    return p.value


def _pinned_free(addr):
    load().papof_host_free(c_void_p(addr))


def pinned_pool():
    """the process-wide pool of page-locked result blocks (created on first use)"""
    global _pool
    if _pool is None:
        _pool = _pinned_pool_mod.PinnedPool(_pinned_alloc, _pinned_free)
    return _pool


class _PinnedBlock(object):
    """Owns one pinned block; returns it to the pool when the numpy array built on it is garbage-collected."""
    __slots__ = ("addr", "cls", "__weakref__")

    def __init__(self, addr, cls):
        self.addr, self.cls = addr, cls

    def __del__(self):
        try:
            pinned_pool().give_back(self.addr, self.cls)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def result_array(shape):
    """float64 array for (vx, vy, warpI2): pinned and recycled (PAPOF_PINNED_OUT=0: plain np.zeros).  Every element is
    overwritten by the call it is handed to, as the reference overwrites its np.zeros arrays."""
    nbytes = 8 * int(np.prod(shape))
    if os.environ.get("PAPOF_PINNED_OUT", "1") == "0" or nbytes < (1 << 20):
        return np.zeros(shape)
    got = pinned_pool().take(nbytes)
    if got is None:  # budget exhausted (PAPOF_PINNED_BUDGET_MB, default 1024) or the allocator failed
        return np.zeros(shape)
    addr, cls = got
    buf = (ctypes.c_char * nbytes).from_address(addr)
    buf._papof_block = _PinnedBlock(addr, cls)  # the array keeps `buf` alive, `buf` keeps the block alive
    return np.frombuffer(buf, dtype=np.float64).reshape(shape)


def default_params(**overrides):
    p = Params()
    load().papof_default_params(ctypes.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError("unknown solver parameter %r" % k)
        setattr(p, k, v)
    return p


def timing_keys():
    L = load()
    return [L.papof_timing_key(i).decode() for i in range(N_TIMERS)]


def _p(a):
    return a.ctypes.data_as(_D)


def _c(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError("expected a %d-D array, got shape %r" % (ndim, a.shape))
    return a


def _pb(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))


def _u8(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8 or a.ndim != 3:
        raise ValueError("expected a uint8 H x W x C array, got %s %r" % (a.dtype, a.shape))
    return a


def _chk(rc, what):
    if rc != 0:
        L = load()
        raise PapofError(rc, what, (L.papof_strerror(rc).decode(), L.papof_last_error().decode()))


def format_timing(t):
    """The reference returns std::to_string(double) values (src/OpticalFlow.cpp:850-860): '%f' strings."""
    return {k: "%f" % float(v) for k, v in zip(timing_keys(), t)}


class Papof:
    """One device handle (arena + stream).  Methods mirror the reference functions on the hot path and take /
    return numpy arrays in the reference's HWC float64 layout."""

    def __init__(self, device=0):
        self.L = load()
        self.h = c_void_p()
        _chk(self.L.papof_create(device, ctypes.byref(self.h)), "papof_create")

    def set_graph_mode(self, on=True):
        """hipGraph replay of whole calls (include/papof.h: papof_set_graph_mode)."""
        _chk(self.L.papof_set_graph_mode(self.h, 1 if on else 0), "papof_set_graph_mode")

    def set_stream_overlap(self, on):
        """several streams per call (default) or one (include/papof.h: papof_set_stream_overlap)."""
        _chk(self.L.papof_set_stream_overlap(self.h, 1 if on else 0), "papof_set_stream_overlap")

    def close(self):
        if self.h:
            self.L.papof_destroy(self.h)
            self.h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- whole call -------------------------------------------------------------------------
    def coarse2fine_flow(self, im1, im2, levels, params=None):
        im1, im2 = _c(im1, 3), _c(im2, 3)
        if im1.shape != im2.shape:
            raise ValueError("Im1 %r and Im2 %r differ in shape" % (im1.shape, im2.shape))
        h, w, c = im1.shape
        vx, vy, wi, t = result_array((h, w)), result_array((h, w)), result_array((h, w, c)), np.zeros(N_TIMERS)
        pp = ctypes.byref(params) if params is not None else None
        _chk(self.L.papof_flow(self.h, _p(im1), _p(im2), h, w, c, levels, pp, _p(vx), _p(vy), _p(wi), _p(t)),
             "papof_flow")
        return vx, vy, wi, t

    def coarse2fine_flow_u8(self, im1, im2, levels, params=None):
        """uint8 HWC frames as decoded from the JPEGs; the `/ 255.` of OpticalFlowCalculation.py:69-70 happens on the
        device.  Same return as coarse2fine_flow."""
        im1, im2 = _u8(im1), _u8(im2)
        if im1.shape != im2.shape:
            raise ValueError("Im1 %r and Im2 %r differ in shape" % (im1.shape, im2.shape))
        h, w, c = im1.shape
        vx, vy, wi, t = result_array((h, w)), result_array((h, w)), result_array((h, w, c)), np.zeros(N_TIMERS)
        pp = ctypes.byref(params) if params is not None else None
        _chk(self.L.papof_flow_u8(self.h, _pb(im1), _pb(im2), h, w, c, levels, pp, _p(vx), _p(vy), _p(wi), _p(t)),
             "papof_flow_u8")
        return vx, vy, wi, t

    # ---- sequence mode: one upload + one pyramid per frame (TestSuite.py:69-81 walks overlapping pairs) ----------
    def seq_reset(self):
        _chk(self.L.papof_seq_reset(self.h), "papof_seq_reset")

    def seq_push(self, frame, levels, params=None, out=None):
        """Push the next frame of a video (float64 in [0,1] or uint8, HWC).  Returns None for the frame that primes
        the sequence, else (vx, vy, warpI2, timing) of the flow from the previous frame to this one.  `out` = (vx, vy,
        warpI2) float64 arrays to write into (reusing them spares the page faults of 83 MB of fresh memory per 1080p
        pair, which cost more than the PCIe transfers)."""
        u8 = isinstance(frame, np.ndarray) and frame.dtype == np.uint8
        frame = _u8(frame) if u8 else _c(frame, 3)
        h, w, c = frame.shape
        t = np.zeros(N_TIMERS)
        if out is not None:
            vx, vy, wi = out
            for a_, shp in ((vx, (h, w)), (vy, (h, w)), (wi, (h, w, c))):
                if a_.dtype != np.float64 or a_.shape != shp or not a_.flags["C_CONTIGUOUS"]:
                    raise ValueError("out arrays must be C-contiguous float64 of shapes (h,w), (h,w), (h,w,c)")
        else:
            vx, vy, wi = result_array((h, w)), result_array((h, w)), result_array((h, w, c))
        pp = ctypes.byref(params) if params is not None else None
        have = c_int(0)
        if u8:
            _chk(self.L.papof_seq_push_u8(self.h, _pb(frame), h, w, c, levels, pp, _p(vx), _p(vy), _p(wi), _p(t),
                                          ctypes.byref(have)), "papof_seq_push_u8")
        else:
            _chk(self.L.papof_seq_push(self.h, _p(frame), h, w, c, levels, pp, _p(vx), _p(vy), _p(wi), _p(t),
                                       ctypes.byref(have)), "papof_seq_push")
        return (vx, vy, wi, t) if have.value else None

    def seq_push_device(self, d_frame, is_u8, h, w, c, levels, params, d_vx, d_vy, d_warp):
        t = np.zeros(N_TIMERS)
        have = c_int(0)
        pp = ctypes.byref(params) if params is not None else None
        _chk(self.L.papof_seq_push_device(self.h, d_frame, int(is_u8), h, w, c, levels, pp, d_vx, d_vy, d_warp, _p(t),
                                          ctypes.byref(have)), "papof_seq_push_device")
        return t if have.value else None

    def coarse2fine_flow_sched(self, im1, im2, levels, alpha, ratio, n_outer, outer_step, n_inner, n_sor, sor_step,
                               mode=SOR_EXACT, omega=1.8):
        p = default_params(alpha=alpha, ratio=ratio, n_outer=n_outer, n_outer_per_level=outer_step, n_inner=n_inner,
                           n_sor=n_sor, n_sor_per_level=sor_step, sor_mode=mode, omega=omega)
        return self.coarse2fine_flow(im1, im2, levels, p)[:3]

    # ---- device-resident buffers (bench) ----------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = c_void_p()
        _chk(self.L.papof_dev_alloc(self.h, nbytes, ctypes.byref(p)), "papof_dev_alloc")
        return p

    def dev_free(self, p):
        _chk(self.L.papof_dev_free(self.h, p), "papof_dev_free")

    def dev_upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        _chk(self.L.papof_dev_upload(self.h, dptr, arr.ctypes.data_as(c_void_p), arr.nbytes), "papof_dev_upload")

    def dev_download(self, arr, dptr):
        _chk(self.L.papof_dev_download(self.h, arr.ctypes.data_as(c_void_p), dptr, arr.nbytes), "papof_dev_download")

    def flow_device(self, d_im1, d_im2, h, w, c, levels, params, d_vx, d_vy, d_warp):
        t = np.zeros(N_TIMERS)
        pp = ctypes.byref(params) if params is not None else None
        _chk(self.L.papof_flow_device(self.h, d_im1, d_im2, h, w, c, levels, pp, d_vx, d_vy, d_warp, _p(t)),
             "papof_flow_device")
        return t

    def stream(self):
        return self.L.papof_stream(self.h)

    # ---- stages --------------------------------------------------------------------------------------
    def pyramid(self, im, ratio, levels):
        im = _c(im, 3)
        h, w, c = im.shape
        dims = np.zeros(2 * levels, dtype=np.int32)
        total = ctypes.c_long(0)
        _chk(self.L.papof_stage_pyramid(self.h, _p(im), h, w, c, ratio, levels, dims.ctypes.data_as(_I), None,
                                        ctypes.byref(total)), "papof_stage_pyramid")
        data = np.zeros(total.value)
        _chk(self.L.papof_stage_pyramid(self.h, _p(im), h, w, c, ratio, levels, dims.ctypes.data_as(_I), _p(data),
                                        ctypes.byref(total)), "papof_stage_pyramid")
        out, off = [], 0
        for i in range(levels):
            lw, lh = int(dims[2 * i]), int(dims[2 * i + 1])
            out.append(data[off:off + lw * lh * c].reshape(lh, lw, c).copy())
            off += lw * lh * c
        return out

    def gaussian_smoothing(self, im, sigma, fsize):
        im = _c(im, 3)
        h, w, c = im.shape
        out = np.zeros_like(im)
        _chk(self.L.papof_stage_gaussian(self.h, _p(im), h, w, c, sigma, fsize, _p(out)), "papof_stage_gaussian")
        return out

    def resize_ratio(self, im, ratio):
        im = _c(im, 3)
        h, w, c = im.shape
        out = np.zeros((int(float(h) * ratio), int(float(w) * ratio), c))
        _chk(self.L.papof_stage_resize_ratio(self.h, _p(im), h, w, c, ratio, _p(out)), "papof_stage_resize_ratio")
        return out

    def resize_wh(self, im, dw, dh):
        im = _c(im, 3)
        h, w, c = im.shape
        out = np.zeros((dh, dw, c))
        _chk(self.L.papof_stage_resize_wh(self.h, _p(im), h, w, c, dw, dh, _p(out)), "papof_stage_resize_wh")
        return out

    def im2feature(self, im):
        im = _c(im, 3)
        h, w, c = im.shape
        fc = c_int(0)
        _chk(self.L.papof_stage_im2feature(self.h, None, h, w, c, None, ctypes.byref(fc)), "papof_stage_im2feature")
        out = np.zeros((h, w, fc.value))
        _chk(self.L.papof_stage_im2feature(self.h, _p(im), h, w, c, _p(out), ctypes.byref(fc)),
             "papof_stage_im2feature")
        return out

    def warpFL(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1, 3), _c(im2, 3), _c(vx, 2), _c(vy, 2)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        _chk(self.L.papof_stage_warpFL(self.h, _p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out)),
             "papof_stage_warpFL")
        return out

    def getDxs(self, im1, im2):
        im1, im2 = _c(im1, 3), _c(im2, 3)
        h, w, c = im1.shape
        dx, dy, dt = np.zeros_like(im1), np.zeros_like(im1), np.zeros_like(im1)
        _chk(self.L.papof_stage_getDxs(self.h, _p(im1), _p(im2), h, w, c, _p(dx), _p(dy), _p(dt)),
             "papof_stage_getDxs")
        return dx, dy, dt

    def linear_system(self, im1, warp, u, v, alpha=0.012):
        im1, warp, u, v = _c(im1, 3), _c(warp, 3), _c(u, 2), _c(v, 2)
        h, w, c = im1.shape
        outs = [np.zeros((h, w)) for _ in range(6)]
        _chk(self.L.papof_stage_linear_system(self.h, _p(im1), _p(warp), _p(u), _p(v), h, w, c, alpha,
                                              *[_p(o) for o in outs]), "papof_stage_linear_system")
        return outs  # phi, imdxy, imdx2, imdy2, rhs1, rhs2

    def laplacian(self, x, weight):
        x, weight = _c(x, 2), _c(weight, 2)
        h, w = x.shape
        out = np.zeros_like(x)
        _chk(self.L.papof_stage_laplacian(self.h, _p(x), _p(weight), h, w, _p(out)), "papof_stage_laplacian")
        return out

    def sor(self, phi, imdxy, imdx2, imdy2, rhs1, rhs2, n_sor, alpha=0.012, omega=1.8, mode=SOR_EXACT):
        arrs = [_c(a, 2) for a in (phi, imdxy, imdx2, imdy2, rhs1, rhs2)]
        h, w = arrs[0].shape
        du, dv = np.zeros((h, w)), np.zeros((h, w))
        _chk(self.L.papof_stage_sor(self.h, *[_p(a) for a in arrs], h, w, alpha, omega, n_sor, mode, _p(du), _p(dv)),
             "papof_stage_sor")
        return du, dv

    def smoothflow_sor(self, im1, im2, warp, u, v, alpha, n_outer, n_inner, n_sor, omega=1.8, mode=SOR_EXACT):
        im1, im2 = _c(im1, 3), _c(im2, 3)
        warp, u, v = _c(warp, 3).copy(), _c(u, 2).copy(), _c(v, 2).copy()
        h, w, c = im1.shape
        _chk(self.L.papof_stage_smoothflow(self.h, _p(im1), _p(im2), _p(warp), _p(u), _p(v), h, w, c, alpha, n_outer,
                                           n_inner, n_sor, omega, mode), "papof_stage_smoothflow")
        return warp, u, v

    # ---- the reference's non-default branches (include/papof.h: PAPOF_INTERP_*, PAPOF_NOISE_*, min-width pyramid) ----
    def pyramid_minwidth(self, im, ratio, min_width):
        """GaussianPyramid::ConstructPyramid(image, ratio, minWidth), src/GaussianPyramid.cpp:47-77"""
        n = c_int(0)
        _chk(self.L.papof_pyramid_levels_for_min_width(np.shape(im)[1], ratio, min_width, ctypes.byref(n)),
             "papof_pyramid_levels_for_min_width")
        return self.pyramid(im, ratio, n.value)

    def coarse2fine_flow_opts(self, im1, im2, levels, interpolation, noise_model):
        return self.coarse2fine_flow(im1, im2, levels, default_params(interpolation=interpolation,
                                                                      noise_model=noise_model))[:3]

    def smoothflow_sor_opts(self, im1, im2, warp, u, v, alpha, n_outer, n_inner, n_sor, interpolation, noise_model):
        im1, im2 = _c(im1, 3), _c(im2, 3)
        warp, u, v = _c(warp, 3).copy(), _c(u, 2).copy(), _c(v, 2).copy()
        h, w, c = im1.shape
        gm = np.concatenate([np.full(c, 0.95), np.full(c, 0.05), np.full(c, 0.5), np.full(c, 0.05) ** 2,
                             np.full(c, 0.5) ** 2])
        _chk(self.L.papof_stage_smoothflow_ex(self.h, _p(im1), _p(im2), _p(warp), _p(u), _p(v), h, w, c, alpha, n_outer,
                                              n_inner, n_sor, 1.8, SOR_EXACT, interpolation, noise_model, _p(gm)),
             "papof_stage_smoothflow_ex")
        return warp, u, v, gm

    def est_gaussian_mixture(self, im1, im2, gm=None):
        im1, im2 = _c(im1, 3), _c(im2, 3)
        h, w, c = im1.shape
        if gm is None:
            gm = np.concatenate([np.full(c, 0.95), np.full(c, 0.05), np.full(c, 0.5), np.full(c, 0.05) ** 2,
                                 np.full(c, 0.5) ** 2])
        gm = np.ascontiguousarray(gm, dtype=np.float64).copy()
        _chk(self.L.papof_stage_est_gaussian_mixture(self.h, _p(im1), _p(im2), h, w, c, _p(gm)),
             "papof_stage_est_gaussian_mixture")
        return gm

    def bicubic_warp_noclamp(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1, 3), _c(im2, 3), _c(vx, 2), _c(vy, 2)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        _chk(self.L.papof_stage_bicubic_warp_ex(self.h, _p(im1), _p(im2), _p(vx), _p(vy), h, w, c, 0, _p(out)),
             "papof_stage_bicubic_warp_ex")
        return out

    def bicubic_warp(self, im1, im2, vx, vy):
        im1, im2, vx, vy = _c(im1, 3), _c(im2, 3), _c(vx, 2), _c(vy, 2)
        h, w, c = im1.shape
        out = np.zeros_like(im1)
        _chk(self.L.papof_stage_bicubic_warp(self.h, _p(im1), _p(im2), _p(vx), _p(vy), h, w, c, _p(out)),
             "papof_stage_bicubic_warp")
        return out

    def flow_quantize16(self, vx, vy):
        vx, vy = _c(vx, 2), _c(vy, 2)
        h, w = vx.shape
        q = np.zeros((h, w, 2), dtype=np.uint16)
        _chk(self.L.papof_flow_quantize16(self.h, _p(vx), _p(vy), h, w, q.ctypes.data_as(c_void_p)),
             "papof_flow_quantize16")
        return q

    def flow_dequantize16(self, q):
        q = np.ascontiguousarray(q, dtype=np.uint16)
        if q.ndim != 3 or q.shape[2] != 2:
            raise ValueError("expected an H x W x 2 uint16 array, got %r" % (q.shape,))
        h, w, _ = q.shape
        vx, vy = np.zeros((h, w)), np.zeros((h, w))
        _chk(self.L.papof_flow_dequantize16(self.h, q.ctypes.data_as(c_void_p), h, w, _p(vx), _p(vy)),
             "papof_flow_dequantize16")
        return vx, vy

    def flow_to_bgr(self, vx, vy):
        """uint8 H x W x 3 (B, G, R) visualisation of a flow field: generateOutputFlowImageFile of the reference's
        caller (OpticalFlowCalculation.py:143-162) without cv2; parity unpinned (see include/papof.h)."""
        vx, vy = _c(vx, 2), _c(vy, 2)
        h, w = vx.shape
        out = np.zeros((h, w, 3), dtype=np.uint8)
        _chk(self.L.papof_flow_to_bgr(self.h, _p(vx), _p(vy), h, w, out.ctypes.data_as(c_void_p)), "papof_flow_to_bgr")
        return out

    def sor_plan(self, h, w, n_sor, mode=SOR_EXACT):
        """(solver-kernel launches per solve, sweeps [red-black: half-sweeps] per launch) on this handle"""
        nl, d = c_int(0), c_int(0)
        _chk(self.L.papof_sor_plan(self.h, h, w, n_sor, mode, ctypes.byref(nl), ctypes.byref(d)), "papof_sor_plan")
        return nl.value, d.value

    def last_sor_stats(self):
        """(exact-order solver launches of the last flow call, seconds of Phase5_SOR that ran on the strip streams)"""
        nl, sec = c_int(0), c_double(0)
        _chk(self.L.papof_last_sor_stats(self.h, ctypes.byref(nl), ctypes.byref(sec)), "papof_last_sor_stats")
        return nl.value, sec.value

    def last_sor_solves(self):
        """[dict(h, w, n_sor, kind, depth, launches, sec)] for every solve of the last flow call, in stream order
        (include/papof.h: papof_last_sor_solves)"""
        n = c_int(0)
        _chk(self.L.papof_last_sor_solves(self.h, 0, ctypes.byref(n), None, None), "papof_last_sor_solves")
        cap = n.value
        if cap == 0:
            return []
        info, sec = (c_int * (6 * cap))(), (ctypes.c_double * cap)()
        _chk(self.L.papof_last_sor_solves(self.h, cap, ctypes.byref(n), info, sec), "papof_last_sor_solves")
        keys = ("h", "w", "n_sor", "kind", "depth", "launches")
        return [dict(zip(keys, info[6 * i:6 * i + 6]), sec=sec[i]) for i in range(min(cap, n.value))]

    def flow_batch(self, frames, levels, params=None, sequence=True, out=None):
        """B frame pairs of one shape in ONE launch chain (include/papof.h: papof_flow_batch*).  frames: a list of HWC arrays,
        all float64 in [0, 1] or all uint8; sequence=True: pair i = (frames[i], frames[i + 1]); False: (frames[2i], frames[2i+1]).
        Returns [(vx, vy, warpI2), ...] per pair (out: a list of such triples to be reused) and the ten timers of the batch.
        Every pair's arrays are bit-identical to coarse2fine_flow(pair)."""
        u8 = np.asarray(frames[0]).dtype == np.uint8
        if isinstance(frames, np.ndarray) and frames.ndim == 4 and frames.flags["C_CONTIGUOUS"]:
            fr = frames if u8 else np.ascontiguousarray(frames, dtype=np.float64)
        else:
            first = np.asarray(frames[0])
            for f in frames:
                if np.shape(f) != first.shape or first.ndim != 3:
                    raise ValueError("all frames of a batch must have one H x W x C shape")
            fr = np.stack([_u8(f) if u8 else _c(f, 3) for f in frames])  # one host block: one upload
        h, w, c = fr[0].shape
        n_pairs = len(fr) - 1 if sequence else len(fr) // 2
        if n_pairs < 1 or (not sequence and len(fr) != 2 * n_pairs):
            raise ValueError("a batch needs at least one pair (sequence: n + 1 frames, else 2 n)")
        if out is None:  # one (page-locked) block per kind, laid out as the device's: the results come back as two copies
            uv, wi = result_array((n_pairs, 2, h, w)), result_array((n_pairs, h, w, c))
            out = [(uv[i, 0], uv[i, 1], wi[i]) for i in range(n_pairs)]
        t = np.zeros(N_TIMERS)
        PP = ctypes.c_void_p * len(fr)
        fp = PP(*[f.ctypes.data for f in fr])
        OP = ctypes.c_void_p * n_pairs
        ox, oy, ow = (OP(*[o[i].ctypes.data for o in out]) for i in range(3))
        pp = ctypes.byref(params) if params is not None else None
        fn = self.L.papof_flow_batch_u8 if u8 else self.L.papof_flow_batch
        _chk(fn(self.h, n_pairs, 1 if sequence else 0, fp, h, w, c, int(levels), pp, ox, oy, ow, _p(t)), "papof_flow_batch")
        return out, t

    def last_host_times(self):
        """(enqueue_sec, wait_sec) of the last call on this handle: host wall time spent enqueueing / waiting for the streams"""
        out = (c_double * 3)()
        _chk(self.L.papof_last_host_times(self.h, out), "papof_last_host_times")
        return out[0], out[1]

    def lap_guard_stats(self):
        """dict(reruns, exact_calls, exact_next, guard_on): the Laplacian-noise guard (include/papof.h: papof_lap_guard_stats)"""
        out = (c_int * 4)()
        _chk(self.L.papof_lap_guard_stats(self.h, out), "papof_lap_guard_stats")
        return dict(reruns=out[0], exact_calls=out[1], exact_next=bool(out[2]), guard_on=bool(out[3]))

    def test_sor_strips(self, h, w, n_sor, split_band, reps=3, delay_us=0):
        """(mismatching cells of a solve cut into two strips vs the whole solve, bands of the layout)"""
        mm, nb = ctypes.c_longlong(0), c_int(0)
        _chk(self.L.papof_test_sor_strips(self.h, h, w, n_sor, split_band, reps, delay_us, ctypes.byref(mm),
                                          ctypes.byref(nb)), "papof_test_sor_strips")
        return mm.value, nb.value

    def strip_plan(self, h, w, n_sor, n_outer, want=0):
        """strip schedule of one level: dict(S, band_rows, koff, bands, plan[n][s] = (band, rU, rP, rS, rA))"""
        S, br, ko, nb = c_int(0), c_int(0), c_int(0), c_int(0)
        cap = (n_outer + 1) * 5 * 5
        buf = (c_int * cap)()
        _chk(self.L.papof_strip_plan(self.h, h, w, n_sor, n_outer, want, ctypes.byref(S), buf, cap, ctypes.byref(br),
                                     ctypes.byref(ko), ctypes.byref(nb)), "papof_strip_plan")
        plan = []
        if S.value > 1:
            for n in range(n_outer + 1):
                plan.append([tuple(buf[(n * (S.value + 1) + s) * 5 + i] for i in range(5)) for s in range(S.value + 1)])
        return {"S": S.value, "band_rows": br.value, "koff": ko.value, "bands": nb.value, "plan": plan}

    def bench_sor(self, h, w, n_sor, mode=SOR_EXACT, reps=5, seed=2):
        ms = c_double(0)
        _chk(self.L.papof_bench_sor(self.h, h, w, n_sor, mode, reps, seed, ctypes.byref(ms)), "papof_bench_sor")
        return ms.value


# ---- one frame pair sharded as 2-D tiles over several ranks (include/papof.h, csrc/tiles.hip) ---------------------
TILES_ID_BYTES = 128


def tiles_grid(nranks):
    r, c = c_int(0), c_int(0)
    _chk(load().papof_tiles_grid(nranks, ctypes.byref(r), ctypes.byref(c)), "papof_tiles_grid")
    return r.value, c.value


def tiles_rect(width, height, rows, cols, rank):
    out = (c_int * 4)()
    _chk(load().papof_tiles_rect(width, height, rows, cols, rank, out), "papof_tiles_rect")
    return tuple(out)


def tiles_halo_message(width, height, rows, cols, halo, src, dst):
    out = (c_int * 4)()
    _chk(load().papof_tiles_halo_message(width, height, rows, cols, halo, src, dst, out), "papof_tiles_halo_message")
    return tuple(out)


def bands_plan(height, width, n_sor, nranks, rank):
    """exact-order band split of one level: dict(B0, B1, coef_rows, final_rows) of `rank` (include/papof.h)"""
    out = (c_int * 6)()
    _chk(load().papof_bands_plan(height, width, n_sor, nranks, rank, out), "papof_bands_plan")
    return {"B0": out[0], "B1": out[1], "coef_rows": (out[2], out[3]), "final_rows": (out[4], out[5])}


def tiles_unique_id():
    buf = ctypes.create_string_buffer(TILES_ID_BYTES)
    _chk(load().papof_tiles_unique_id(buf), "papof_tiles_unique_id")
    return buf.raw


class TileRank:
    """One rank of a tile group (RCCL transport: one per process / GPU)."""

    def __init__(self, gpu, t):
        self.gpu, self.t = gpu, t

    @classmethod
    def create(cls, gpu, unique_id, rank, nranks, rows=0, cols=0, halo=0):
        if not rows:
            rows, cols = tiles_grid(nranks)
        t = c_void_p()
        _chk(gpu.L.papof_tiles_create(gpu.h, unique_id, rank, nranks, rows, cols, halo, ctypes.byref(t)),
             "papof_tiles_create")
        return cls(gpu, t)

    def flow_device(self, d_im1, d_im2, h, w, c, levels, params, d_vx=None, d_vy=None, d_warp=None):
        t = np.zeros(N_TIMERS)
        pp = ctypes.byref(params) if params is not None else None
        _chk(self.gpu.L.papof_tiles_flow_device(self.t, d_im1, d_im2, h, w, c, levels, pp, d_vx, d_vy, d_warp, _p(t)),
             "papof_tiles_flow_device")
        return t

    def stats(self):
        n, b = ctypes.c_long(0), ctypes.c_size_t(0)
        _chk(self.gpu.L.papof_tiles_stats(self.t, ctypes.byref(n), ctypes.byref(b)), "papof_tiles_stats")
        return n.value, b.value

    def comm_info(self):
        """{nranks_seen, rank_seen, rows, cols, halo}: the first two as the transport reports them (RCCL communicator)"""
        v = [c_int(0) for _ in range(5)]
        _chk(self.gpu.L.papof_tiles_comm_info(self.t, *[ctypes.byref(x) for x in v]), "papof_tiles_comm_info")
        return dict(zip(("nranks_seen", "rank_seen", "rows", "cols", "halo"), (x.value for x in v)))

    def close(self):
        if self.t:
            self.gpu.L.papof_tiles_destroy(self.t)
            self.t = c_void_p()


class _TileGroup:
    """All ranks of a tile group inside this process, each driven by its own host thread: the parity tests of the sharded
    paths on a one-GPU box."""

    def __init__(self, nranks, rows=0, cols=0, halo=0, device=0):
        if not rows:
            rows, cols = tiles_grid(nranks)
        self.n, self.rows, self.cols, self.halo = nranks, rows, cols, halo
        self.gpus = [Papof(device) for _ in range(nranks)]
        self.ranks = []

    def coarse2fine_flow(self, im1, im2, levels, params):
        """Host arrays in / out (HWC float64); every rank gets its own device copy of the frames, as one process per
        GPU would have."""
        import threading
        im1, im2 = _c(im1, 3), _c(im2, 3)
        h, w, c = im1.shape
        bufs = []
        for g in self.gpus:
            d1, d2 = g.dev_alloc(im1.nbytes), g.dev_alloc(im2.nbytes)
            g.dev_upload(d1, im1)
            g.dev_upload(d2, im2)
            bufs.append((d1, d2))
        g0 = self.gpus[0]
        dvx, dvy, dwp = g0.dev_alloc(h * w * 8), g0.dev_alloc(h * w * 8), g0.dev_alloc(im1.nbytes)
        errs, times = [None] * self.n, [None] * self.n

        def run(r):
            try:
                o = (dvx, dvy, dwp) if r == 0 else (None, None, None)
                times[r] = self.ranks[r].flow_device(bufs[r][0], bufs[r][1], h, w, c, levels, params, *o)
            except Exception as e:  # noqa: BLE001
                errs[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(self.n)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        try:
            for e in errs:
                if e is not None:
                    raise e
            vx, vy, wi = np.zeros((h, w)), np.zeros((h, w)), np.zeros((h, w, c))
            g0.dev_download(vx, dvx)
            g0.dev_download(vy, dvy)
            g0.dev_download(wi, dwp)
        finally:
            for g, (d1, d2) in zip(self.gpus, bufs):
                g.dev_free(d1)
                g.dev_free(d2)
            for p in (dvx, dvy, dwp):
                g0.dev_free(p)
        return vx, vy, wi, times[0]

    def close(self):
        import threading
        # (a communicator's destruction may wait for its peers', as ncclCommDestroy does: one thread per rank)
        th = [threading.Thread(target=r.close) for r in self.ranks]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for g in self.gpus:
            g.close()


class LocalTileGroup(_TileGroup):
    """LOCAL transport (csrc/tiles.hip): messages are device copies between the ranks' buffers, every exchange synchronises the
    rank's stream and passes two host barriers of the whole group; kernels of one rank may store into a peer's planes (the
    DIRECT protocol of the exact-order band split)."""

    def __init__(self, nranks, rows=0, cols=0, halo=0, device=0):
        _TileGroup.__init__(self, nranks, rows, cols, halo, device)
        hs = (c_void_p * nranks)(*[g.h for g in self.gpus])
        ts = (c_void_p * nranks)()
        _chk(self.gpus[0].L.papof_tiles_create_local(hs, nranks, self.rows, self.cols, halo, ts), "papof_tiles_create_local")
        self.ranks = [TileRank(g, c_void_p(t)) for g, t in zip(self.gpus, ts)]


class RcclTileGroup(_TileGroup):
    """The RCCL transport itself -- papof_tiles_create, one communicator rank per thread -- which is what one process per GPU
    runs on a multi-GPU node.  On ONE device the real librccl refuses several ranks, so PAPOF_RCCL_LIB must name a library
    with RCCL's API whose ranks may share a device: tests/fake_rccl (stream-ordered group kernels, no host synchronisation)."""

    def __init__(self, nranks, rows=0, cols=0, halo=0, device=0):
        import threading
        _TileGroup.__init__(self, nranks, rows, cols, halo, device)
        uid = tiles_unique_id()
        made, errs = [None] * nranks, [None] * nranks

        def init(r):  # ncclCommInitRank returns when every rank has called it
            try:
                made[r] = TileRank.create(self.gpus[r], uid, r, nranks, self.rows, self.cols, halo)
            except Exception as e:  # noqa: BLE001
                errs[r] = e
        th = [threading.Thread(target=init, args=(r,)) for r in range(nranks)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        self.ranks = [m for m in made if m is not None]
        for e in errs:
            if e is not None:
                self.close()
                raise e
