"""Build-time check of the generated gfx950 code (no GPU needed: hipcc cross-compiles): no VALU write of a 16-byte store's
data registers within fewer than two wait states behind the store.  The compiler keeps its own instructions apart but
not the inline-asm moves of sor.hip's software pipeline; round 2 found such a site to corrupt quads of lanes whenever a
second wave shared the SIMD -- sor.hip guards every solver store since (store_data_guard)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "papteam_opticalflow_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.parametrize("unit,extra", [("sor", ["-mllvm", "-structurizecfg-skip-uniform-regions=1"]), ("kernels", []), ("tiles", [])])
def test_no_store_data_hazard_in_generated_code(tmp_path, unit, extra):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = str(tmp_path / (unit + ".s"))
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
           "--offload-device-only", "-S", "-o", asm, os.path.join(CSRC, unit + ".hip")] + extra
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_store_hazard.py"), asm], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout
