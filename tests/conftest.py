import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


# One HIP stream per hardware queue for everything a test keeps in flight at once: the 8-rank groups on the RCCL stand-in
# (tests/fake_rccl) are 8 handles x 2 streams whose group kernels WAIT FOR EACH OTHER on the device, and two streams that share
# a hardware queue run their kernels one after the other -- a receive in front of the send it waits for would never end
# (on the multi-GPU node every rank has a device to itself).  Must be in the environment before the HIP runtime starts;
# the binding's own default is 16 (capi.load).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: > 15 s of CPU work; enabled by PAPOF_SLOW=1")


@pytest.fixture(scope="session")
def oracle():
    from _libs import OracleLib
    return OracleLib()


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


@pytest.fixture(params=["local", "rccl"])
def tile_group(request, monkeypatch):
    """The class that builds a multi-rank tile group inside this process, once per TRANSPORT of csrc/tiles.hip:
    `local`  LocalTileGroup: device copies behind hipStreamSynchronize + two host barriers of all ranks per exchange;
    `rccl`   RcclTileGroup: the RCCL transport itself (papof_tiles_create, ncclSend / ncclRecv groups on the rank's stream, no host
             synchronisation, barrier() a no-op) bound to tests/fake_rccl -- a stand-in with RCCL's stream-ordered semantics whose
             ranks may share the one device of this box.  Afterwards the stand-in must have carried messages and seen no error
             (a receive or send that gave up, byte counts that disagree)."""
    import ctypes
    from papteam_opticalflow_amd import capi
    if request.param == "local":
        yield capi.LocalTileGroup
        return
    assert os.path.exists(FAKE_RCCL), "tests/fake_rccl/libfake_rccl.so is not built (__graft_entry__.build())"
    # The ranks' group kernels wait for each other on the device, so every rank's stream needs a hardware queue of its own (see the
    # top of this file): give back the streams that earlier tests left pooled -- flow_collection()'s handles, the default handle
    import gc
    import papteam_opticalflow_amd as pkg
    for pool in pkg._collection_handles.values():
        for hnd in pool:
            hnd.close()
        pool.clear()
    if pkg._default is not None:
        pkg._default.close()
        pkg._default = None
    gc.collect()
    monkeypatch.setenv("PAPOF_RCCL_LIB", FAKE_RCCL)
    monkeypatch.setenv("PAPOF_TILES_TIMEOUT_S", os.environ.get("PAPOF_TILES_TIMEOUT_S", "40"))
    fake = ctypes.CDLL(FAKE_RCCL)
    fake.fake_rccl_error_count.restype = ctypes.c_uint
    fake.fake_rccl_reset_errors()
    before = (ctypes.c_long * 3)()
    fake.fake_rccl_stats(before)
    capi.RcclTileGroup.standin = fake
    yield capi.RcclTileGroup
    after = (ctypes.c_long * 3)()
    fake.fake_rccl_stats(after)
    kinds = (ctypes.c_uint * 4)()
    fake.fake_rccl_error_kinds(kinds)
    fake.fake_rccl_reset_errors()
    assert kinds[0] == 0 or getattr(request.node, "_standin_errors_expected", False), ("the RCCL stand-in saw %d errors: %d sends and %d receives gave up, %d messages whose two ends disagree "
                           "about the byte count" % tuple(kinds))
    if not getattr(request.node, "_standin_may_be_idle", False):
        assert after[1] > before[1] and after[2] > before[2], "no message went through the RCCL stand-in"
    print("RCCL stand-in: %d group kernels, %d sends, %d receives" % tuple(after[i] - before[i] for i in range(3)))
