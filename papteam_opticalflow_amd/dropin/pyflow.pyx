# cython: language_level=3
# distutils: language = c
"""Drop-in replacement for the reference's Cython module `pyflow`
(/root/reference/Code/Serial/pyflow.pyx:31-70; Code/Parallel/pyflow.pyx adds a 4th positional `nCores`).

    timing, vx, vy, warpI2 = pyflow.coarse2fine_flow(Im1, Im2, pyramidLevels[, nCores])

Same call signature, same 4-tuple, same buffer typing (C-contiguous float64, ndim 3, not None -> the same
TypeError/ValueError from Cython's buffer checks), same dict of ten '%f'-formatted strings in the reference's
std::map key order.  Underneath, host glue only: the work is one call through the C ABI of include/papof.h
into the HIP library (libpapof.so, gfx950).  Differences from the reference, all additive:
  * Im2's shape is checked against Im1's (the reference reads out of bounds, pyflow.pyx:44-46,63-66);
  * pyramidLevels < 1 raises ValueError (undefined behaviour in the reference, src/GaussianPyramid.cpp:87-88);
  * keyword-only solver parameters (names of struct papof_params) override the reference's hard-coded constants;
  * nCores is accepted and ignored (there are no host worker threads).
"""
import numpy as np
cimport numpy as np

np.import_array()

cdef extern from "papof.h":
    ctypedef struct papof_params:
        double alpha
        double ratio
        int n_outer
        int n_outer_per_level
        int n_inner
        int n_sor
        int n_sor_per_level
        double omega
        int sor_mode
        int phase_timing
        int interpolation
        int noise_model
    void papof_default_params(papof_params* p)
    const char* papof_strerror(int code)
    const char* papof_last_error()
    const char* papof_timing_key(int index)
    int papof_host_alloc(size_t nbytes, void** out)
    int papof_host_free(void* p)
    int papof_coarse2fine_flow(const double* im1, const double* im2, int h, int w, int c, int pyramid_levels,
                               const papof_params* params, double* vx, double* vy, double* warpI2,
                               double* timing_sec) nogil

# ---- result arrays in page-locked memory, recycled when the caller drops them (include/papof.h: papof_host_alloc).
# The reference allocates vx, vy, warpI2 with np.zeros for every call (Code/Serial/pyflow.pyx:44-52); every element is
# overwritten by the call, so recycled memory gives the same arrays -- without 83 MB of first-touch page faults, a staged
# copy and an munmap per 1080p pair.  The pool is BOUNDED (papteam_opticalflow_amd/pinned_pool.py: the budget counts idle
# and live blocks, PAPOF_PINNED_BUDGET_MB, default 1024; size classes; LRU eviction; drained at exit); beyond it, or with
# PAPOF_PINNED_OUT=0, the arrays are plain np.zeros as in the reference.
# fork(): page-locked host memory is not inherited by a forked child (ROCm maps it that way) -- a caller that forks
# workers AFTER a call and reads the returned arrays in the child must copy them first (np.array(vx)), use the spawn
# start method, or set PAPOF_PINNED_OUT=0.
import os as _os
from cpython.ref cimport Py_INCREF
from libc.stdint cimport uintptr_t

_PINNED = _os.environ.get("PAPOF_PINNED_OUT", "1") != "0"


def _load_pool_module():
    import importlib.util
    path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), _os.pardir, "pinned_pool.py")
    spec = importlib.util.spec_from_file_location("_papof_pinned_pool", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _host_alloc(nbytes):
    cdef void* p = NULL
    if papof_host_alloc(<size_t> nbytes, &p) != 0 or p == NULL:
        return 0
    return <uintptr_t> p


def _host_free(addr):
    papof_host_free(<void*> <uintptr_t> addr)


_pool = None


def _pinned_pool():
    """the process-wide PinnedPool of this module (None: pinned_pool.py not found -> plain np.zeros results)"""
    global _pool, _PINNED
    if _pool is None and _PINNED:
        try:
            _pool = _load_pool_module().PinnedPool(_host_alloc, _host_free)
        except Exception:
            _PINNED = False
    return _pool


cdef class _PinnedBlock:
    cdef uintptr_t addr
    cdef size_t cls

    def __dealloc__(self):
        if self.addr != 0:
            try:
                if _pool is not None:
                    _pool.give_back(self.addr, self.cls)
            except Exception:  # interpreter shutdown
                pass
            self.addr = 0


cdef object _result_array(tuple shape):
    cdef size_t nbytes = 8
    for d in shape:
        nbytes *= <size_t> d
    if not _PINNED or nbytes < (1 << 20):
        return np.zeros(shape, dtype=np.float64)
    pool = _pinned_pool()
    got = pool.take(nbytes) if pool is not None else None
    if got is None:
        return np.zeros(shape, dtype=np.float64)
    cdef _PinnedBlock blk = _PinnedBlock.__new__(_PinnedBlock)
    blk.addr = <uintptr_t> got[0]
    blk.cls = <size_t> got[1]
    cdef np.npy_intp dims[3]
    cdef int nd = len(shape)
    for i in range(nd):
        dims[i] = shape[i]
    arr = np.PyArray_SimpleNewFromData(nd, dims, np.NPY_FLOAT64, <void*> blk.addr)
    Py_INCREF(blk)
    np.PyArray_SetBaseObject(arr, blk)
    return arr


SOR_EXACT = 0
SOR_REDBLACK = 1
SOR_JACOBI = 2


def coarse2fine_flow(np.ndarray[double, ndim=3, mode="c"] Im1 not None,
                     np.ndarray[double, ndim=3, mode="c"] Im2 not None,
                     int pyramidLevels, int nCores=1, **solver):
    cdef int h = Im1.shape[0]
    cdef int w = Im1.shape[1]
    cdef int c = Im1.shape[2]
    if Im2.shape[0] != h or Im2.shape[1] != w or Im2.shape[2] != c:
        raise ValueError("Im2 shape (%d, %d, %d) differs from Im1 shape (%d, %d, %d)"
                         % (Im2.shape[0], Im2.shape[1], Im2.shape[2], h, w, c))
    if pyramidLevels < 1:
        raise ValueError("pyramidLevels must be >= 1")
    cdef np.ndarray[double, ndim=2, mode="c"] vx = _result_array((h, w))
    cdef np.ndarray[double, ndim=2, mode="c"] vy = _result_array((h, w))
    cdef np.ndarray[double, ndim=3, mode="c"] warpI2 = _result_array((h, w, c))
    cdef double timing[10]
    cdef papof_params P
    papof_default_params(&P)
    for key, value in solver.items():
        if key == "alpha": P.alpha = value
        elif key == "ratio": P.ratio = value
        elif key == "n_outer": P.n_outer = value
        elif key == "n_outer_per_level": P.n_outer_per_level = value
        elif key == "n_inner": P.n_inner = value
        elif key == "n_sor": P.n_sor = value
        elif key == "n_sor_per_level": P.n_sor_per_level = value
        elif key == "omega": P.omega = value
        elif key == "sor_mode": P.sor_mode = value
        elif key == "phase_timing": P.phase_timing = value
        elif key == "interpolation": P.interpolation = value
        elif key == "noise_model": P.noise_model = value
        else:
            raise TypeError("coarse2fine_flow() got an unexpected keyword argument %r" % key)
    cdef int rc
    cdef int i
    cdef double* p1 = <double*> np.PyArray_DATA(Im1)
    cdef double* p2 = <double*> np.PyArray_DATA(Im2)
    cdef double* pvx = <double*> np.PyArray_DATA(vx)
    cdef double* pvy = <double*> np.PyArray_DATA(vy)
    cdef double* pw = <double*> np.PyArray_DATA(warpI2)
    if h * w * c == 0:
        raise ValueError("empty image")
    with nogil:
        rc = papof_coarse2fine_flow(p1, p2, h, w, c, pyramidLevels, &P, pvx, pvy, pw, timing)
    if rc != 0:
        raise RuntimeError("papof_coarse2fine_flow failed: %s (%d) %s"
                           % (papof_strerror(rc).decode(), rc, papof_last_error().decode()))
    TIMER_AS_DICTIONARY = {}
    for i in range(10):
        TIMER_AS_DICTIONARY[papof_timing_key(i).decode("utf-8")] = "%f" % timing[i]
    return TIMER_AS_DICTIONARY, vx, vy, warpI2
