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
# copy and an munmap per 1080p pair.  PAPOF_PINNED_OUT=0 (or more than 1 GiB of such arrays alive) falls back to np.zeros.
import os as _os
from cpython.ref cimport Py_INCREF
from libc.stdint cimport uintptr_t

_pool = {}        # nbytes -> [pointer, ...] ready for reuse
_live_bytes = 0   # bytes of pinned result arrays currently owned by callers
_PINNED = _os.environ.get("PAPOF_PINNED_OUT", "1") != "0"


cdef class _PinnedBlock:
    cdef void* ptr
    cdef size_t nbytes

    def __dealloc__(self):
        global _live_bytes
        if self.ptr != NULL:
            _live_bytes -= self.nbytes
            lst = _pool.setdefault(self.nbytes, [])
            if len(lst) < 6:
                lst.append(<uintptr_t> self.ptr)
            else:
                papof_host_free(self.ptr)
            self.ptr = NULL


cdef object _result_array(tuple shape):
    global _live_bytes
    cdef size_t nbytes = 8
    for d in shape:
        nbytes *= <size_t> d
    if not _PINNED or nbytes < (1 << 20) or _live_bytes + nbytes > (1 << 30):
        return np.zeros(shape, dtype=np.float64)
    cdef void* p = NULL
    lst = _pool.get(nbytes)
    if lst:
        p = <void*> <uintptr_t> lst.pop()
    elif papof_host_alloc(nbytes, &p) != 0 or p == NULL:
        return np.zeros(shape, dtype=np.float64)
    cdef _PinnedBlock blk = _PinnedBlock.__new__(_PinnedBlock)
    blk.ptr = p
    blk.nbytes = nbytes
    _live_bytes += nbytes
    cdef np.npy_intp dims[3]
    cdef int nd = len(shape)
    for i in range(nd):
        dims[i] = shape[i]
    arr = np.PyArray_SimpleNewFromData(nd, dims, np.NPY_FLOAT64, p)
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
