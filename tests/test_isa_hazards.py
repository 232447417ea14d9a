"""Build-time check of the hand-written kernels' ISA: hipcc pads no hazards around inline-asm MFMAs and asm VALU
helpers, so every build is scanned for (a) VALU-written registers read by an MFMA within 2 wait states and
(b) gfx940+ transcendental forwarding (v_exp result read by the next VALU).  tools/isa_hazards.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("unit", ["fa_fwd_api.hip", "fa_bwd_api.hip"])
def test_no_unpadded_mfma_or_trans_hazards(tmp_path, unit):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazards
    out = tmp_path / "fa.s"
    csrc = os.path.join(ROOT, "flash_attention_annotated_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-I", csrc, "-S", "--cuda-device-only", os.path.join(csrc, unit), "-o", str(out)],
                   check=True, stderr=subprocess.DEVNULL)
    violations = isa_hazards.scan(str(out))
    assert not violations, violations[:5]
