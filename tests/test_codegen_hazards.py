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


def _scan_text(tmp_path, body):
    p = tmp_path / "t.s"
    p.write_text("k_test:\n" + body)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_store_hazard.py"), str(p)], capture_output=True, text=True)
    return out.returncode, out.stdout


@pytest.mark.parametrize("name,body,bad", [
    ("store-data WAR, inline-asm writer", "buffer_store_dwordx4 v[4:7], v1, s[0:3], s9 offen\n;;#ASMSTART\nv_mov_b64 v[4:5], v[8:9]\n;;#ASMEND\n", True),
    ("store-data WAR, compiler writer", "buffer_store_dwordx4 v[4:7], v1, s[0:3], s9 offen\nv_add_f64 v[6:7], v[8:9], v[10:11]\n", True),
    ("store-data WAR, guarded", "buffer_store_dwordx4 v[4:7], v1, s[0:3], s9 offen\ns_nop 1\n;;#ASMSTART\nv_mov_b64 v[4:5], v[8:9]\n;;#ASMEND\n", False),
    ("asm write -> DPP read", ";;#ASMSTART\nv_mov_b64 v[4:5], v[8:9]\n;;#ASMEND\nv_mov_b32_dpp v6, v4 wave_shr:1 row_mask:0xf bank_mask:0xf\n", True),
    ("asm write -> DPP read, 2 apart", ";;#ASMSTART\nv_mov_b64 v[4:5], v[8:9]\n;;#ASMEND\nv_add_f64 v[20:21], v[22:23], v[24:25]\nv_mul_f64 v[30:31], v[22:23], v[24:25]\nv_mov_b32_dpp v6, v4 wave_shr:1 row_mask:0xf bank_mask:0xf\n", False),
    ("asm write -> readfirstlane", ";;#ASMSTART\nv_mov_b32 v4, v8\n;;#ASMEND\nv_readfirstlane_b32 s5, v4\n", True),
    ("VALU writes SGPR -> asm VMEM", "v_readfirstlane_b32 s9, v4\ns_nop 1\n;;#ASMSTART\nbuffer_load_dword v7, v1, s[0:3], s9 offen\n;;#ASMEND\n", True),
    ("VALU writes SGPR -> asm VMEM, 5 apart", "v_readfirstlane_b32 s9, v4\ns_nop 4\n;;#ASMSTART\nbuffer_load_dword v7, v1, s[0:3], s9 offen\n;;#ASMEND\n", False),
    ("trans -> non-trans", ";;#ASMSTART\nv_rcp_f32 v4, v8\n;;#ASMEND\nv_mul_f32 v5, v4, v9\n", True),
    ("compiler pair (recognizer's job, not reported)", "v_mov_b32 v4, v8\nv_mov_b32_dpp v6, v4 wave_shr:1 row_mask:0xf bank_mask:0xf\n", False),
])
def test_scanner_knows_the_hazard_kinds(tmp_path, name, body, bad):
    rc, out = _scan_text(tmp_path, body)
    assert (rc != 0) == bad, (name, out)
