"""MI355X-native coarse-to-fine variational optical flow: the one hot path of
ElijahHyndman/PAPTeam_OpticalFlow (`pyflow.coarse2fine_flow`) as hand-written HIP kernels for gfx950 behind
a C ABI (include/papof.h).

    from papteam_opticalflow_amd import coarse2fine_flow          # same signature / return as the reference
    timing, vx, vy, warpI2 = coarse2fine_flow(im1, im2, pyramidLevels)

`papteam_opticalflow_amd/dropin/` holds the Cython module named `pyflow` that the reference's
OpticalFlowCalculation.py / TestSuite.py import unchanged (put that directory on PYTHONPATH).
There is no CPU fallback: without libpapof.so or without a gfx950 device every call raises.
"""
from . import capi  # noqa: F401
from .capi import SOR_EXACT, SOR_JACOBI, SOR_REDBLACK, Papof, PapofError, default_params  # noqa: F401

__version__ = "0.1.0"

_default = None


def _handle():
    global _default
    if _default is None:
        import os
        _default = Papof(int(os.environ.get("PAPOF_DEVICE", "0")))
    return _default


def coarse2fine_flow(Im1, Im2, pyramidLevels, nCores=1, **solver):
    """ctypes twin of dropin/pyflow.pyx (reference: Code/Serial/pyflow.pyx:31-70): returns
    (dict[str, str] of ten timers, vx[h,w], vy[h,w], warpI2[h,w,c]); `nCores` is accepted and ignored."""
    import numpy as np
    for name, a in (("Im1", Im1), ("Im2", Im2)):
        if a is None:
            raise TypeError("Argument '%s' must not be None" % name)
        if not isinstance(a, np.ndarray) or a.dtype != np.float64:
            raise ValueError("Buffer dtype mismatch for %s, expected 'double'" % name)
        if a.ndim != 3:
            raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % a.ndim)
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("ndarray is not C-contiguous")
    if int(pyramidLevels) < 1:
        raise ValueError("pyramidLevels must be >= 1")
    params = default_params(**solver) if solver else None
    vx, vy, warp, t = _handle().coarse2fine_flow(Im1, Im2, int(pyramidLevels), params)
    return capi.format_timing(t), vx, vy, warp
