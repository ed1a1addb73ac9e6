"""BASELINE.json configs[0] (plumbing): the reference's own, UNMODIFIED OpticalFlowCalculation.py / InputCreation
package driven against this repository's drop-in `pyflow` module.

Only possible in the build container (the reference tree does not travel to the GPU box, and this container has no
GPU), so the run is expected to reach our module through the reference's unchanged call
`pyflow.coarse2fine_flow(im1, im2, pyramidLevels)` (OpticalFlowCalculation.py:73) and fail THERE with the loud
no-device error -- which proves the binding: same module name, same positional signature, float64 HWC arrays from PIL.
The positive half of the contract (4-tuple, str-valued timing dict, `u[..., None]` concatenation, tab-joined timing
line) is exercised on the GPU by tests/test_gpu_parity.py::test_pyflow_dropin_entry_point.

`cv2` is not installed here; the reference imports it at module level (OpticalFlowCalculation.py:13) but only uses
it in a function the Serial tree disables (:137), so an empty stub module is put on PYTHONPATH.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

DRIVER = r'''
import os, sys
from OpticalFlowCalculation import *          # the reference's caller, unmodified
gen = TestImagePairGenerator()                # needs cwd under .../Code and images_New beside it
pairs = gen.generateTestImagePairsFromCollectionName("HoChiMinhTraffic_10FPS_240")
print("PAIRS", len(pairs), pairs[0].asStorageString(" -> ", long=False))
try:
    CalculateOpticalFlow(pairs[0], 2, "_plumbing")
    print("RESULT ok", os.path.exists("output/UniversalTiming.txt"))
except RuntimeError as e:
    print("RESULT RuntimeError", e)
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Code", "Serial")), reason="reference tree not present")
def test_unmodified_reference_caller_reaches_our_pyflow(tmp_path):
    import __graft_entry__
    if not os.path.exists(os.path.join(ROOT, "papteam_opticalflow_amd", "csrc", "libpapof.so")):
        __graft_entry__.build()
    serial = tmp_path / "proj" / "Code" / "Serial"
    serial.mkdir(parents=True)
    (serial / "output").mkdir()
    os.symlink(os.path.join(REF, "Code", "Serial", "OpticalFlowCalculation.py"), serial / "OpticalFlowCalculation.py")
    os.symlink(os.path.join(REF, "Code", "Serial", "InputCreation"), serial / "InputCreation")
    os.symlink(os.path.join(REF, "images_New"), tmp_path / "proj" / "images_New")
    stubs = tmp_path / "stubs"
    stubs.mkdir()
    (stubs / "cv2.py").write_text("# empty stand-in: only used by a code path the Serial tree disables\n")
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "papteam_opticalflow_amd", "dropin"), str(stubs)]))
    out = subprocess.run([sys.executable, "-c", DRIVER], cwd=str(serial), env=env, capture_output=True, text=True,
                         timeout=300)
    text = out.stdout + out.stderr
    assert "PAIRS 101" in text, text[-3000:]
    import ctypes
    have_gpu = ctypes.CDLL(os.path.join(ROOT, "papteam_opticalflow_amd", "csrc", "libpapof.so")).papof_device_count() > 0
    if have_gpu:
        assert "RESULT ok True" in text, text[-3000:]
    else:
        assert "RESULT RuntimeError" in text and "no usable gfx950 device" in text, text[-3000:]
