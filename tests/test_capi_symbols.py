"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/papof.h declares, and
fails LOUDLY (no CPU fallback) when there is no GPU.  No compute call is made here."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from papteam_opticalflow_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return capi.load()


def test_every_declared_symbol_is_exported(lib):
    from papteam_opticalflow_amd import capi
    header = open(os.path.join(ROOT, "include", "papof.h")).read()
    declared = sorted(set(re.findall(r"\b(papof_[a-z0-9_A-Z]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libpapof.so does not export " + name
    assert sorted(capi.SYMBOLS) == declared


def test_defaults_are_the_reference_constants(lib):
    from papteam_opticalflow_amd import capi
    p = capi.default_params()
    # src/OpticalFlow.cpp:747-751, :451, :823
    assert (p.alpha, p.ratio, p.n_outer, p.n_outer_per_level, p.n_inner, p.n_sor, p.n_sor_per_level, p.omega,
            p.sor_mode) == (0.012, 0.75, 7, 1, 1, 30, 3, 1.8, 0)
    assert capi.timing_keys() == sorted(capi.timing_keys())  # std::map order
    assert capi.timing_keys()[-1] == "Total C++ Execution"
    assert capi.format_timing([0.5] * 10)["Allocation"] == "0.500000"  # std::to_string(double)


def test_library_does_not_link_the_oracle():
    """The product path must not route through the checker."""
    import subprocess
    from papteam_opticalflow_amd import capi
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "papof_ref" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "papteam_opticalflow_amd")):
        for f in files:
            if f.endswith((".py", ".pyx", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "libpapof_oracle" not in text and "import _libs" not in text, f


def test_no_gpu_means_loud_failure(lib):
    from papteam_opticalflow_amd import capi
    if lib.papof_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.PapofError) as e:
        capi.Papof(0)
    assert e.value.code == -2
    import papteam_opticalflow_amd as pkg
    with pytest.raises(capi.PapofError):
        pkg.coarse2fine_flow(np.zeros((8, 8, 3)), np.zeros((8, 8, 3)), 2)


def test_pyflow_argument_checks_without_gpu(lib):
    sys.path.insert(0, os.path.join(ROOT, "papteam_opticalflow_amd", "dropin"))
    import pyflow
    a = np.zeros((8, 8, 3))
    with pytest.raises(ValueError):
        pyflow.coarse2fine_flow(a, np.zeros((8, 9, 3)), 2)
    with pytest.raises(ValueError):
        pyflow.coarse2fine_flow(a, a, 0)
    with pytest.raises(TypeError):
        pyflow.coarse2fine_flow(a, None, 2)
    with pytest.raises(ValueError):
        pyflow.coarse2fine_flow(a.astype(np.float32), a.astype(np.float32), 2)
    with pytest.raises(TypeError):
        pyflow.coarse2fine_flow(a, a, 2, bogus=1)
    if lib.papof_device_count() == 0:
        with pytest.raises(RuntimeError):
            pyflow.coarse2fine_flow(a, a, 2)


def test_collection_in_flight_policy_and_setter_arguments(lib):
    """flow_collection()'s default number of sequences in flight is a function of the frame size (tools/collection_probe.py on
    MI355X); the C-ABI switches it relies on refuse a null handle instead of crashing."""
    from papteam_opticalflow_amd import capi, collection_in_flight
    sizes = [(135, 240), (270, 480), (540, 960), (1080, 1920), (2160, 3840)]
    got = [collection_in_flight(h, w) for h, w in sizes]
    assert got == [16, 16, 8, 4, 4] and got == sorted(got, reverse=True)
    assert lib.papof_set_stream_overlap(None, 0) == capi.EINVAL if hasattr(capi, "EINVAL") else lib.papof_set_stream_overlap(None, 0) != 0
    out = (capi.c_int * 4)()
    assert lib.papof_lap_guard_stats(None, out) != 0


def test_rccl_standin_builds_loads_and_exports_the_api_the_transport_binds():
    """tests/fake_rccl (TEST infrastructure: the stand-in for librccl behind PAPOF_RCCL_LIB): the library must load without a GPU
    and export every entry point csrc/tiles.hip looks up (rccl(): PAPOF_SYM), plus the hooks the GPU tests read.  No call that
    touches a device is made here."""
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(so)])
    lib = ctypes.CDLL(so)
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclCommCount", "ncclCommUserRank",
                "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString", "fake_rccl_error_count",
                "fake_rccl_error_kinds", "fake_rccl_reset_errors", "fake_rccl_stats", "fake_rccl_identity"):
        assert hasattr(lib, sym), sym
    lib.fake_rccl_identity.restype = ctypes.c_char_p
    assert b"fake rccl" in lib.fake_rccl_identity()
    # the product library must not depend on it (it is dlopen'ed by path, on request only)
    out = subprocess.run(["ldd", os.path.join(root, "papteam_opticalflow_amd", "csrc", "libpapof.so")], capture_output=True, text=True)
    assert "fake_rccl" not in out.stdout
    # grouping without a device: nested starts / ends balance, an unbalanced end is refused
    assert lib.ncclGroupStart() == 0 and lib.ncclGroupStart() == 0 and lib.ncclGroupEnd() == 0 and lib.ncclGroupEnd() == 0
    assert lib.ncclGroupEnd() != 0
