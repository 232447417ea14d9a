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
    # (c) register spills: a scratch reload is an awaited memory round trip.  The pipelined forward kernel has none at all;
    #     the compiler-scheduled forward shapes have none between the first and the last MFMA of their tile loop (soft-cap
    #     instantiations may keep one between the two products); the backward kernels none for head dims <= 128.
    import re
    txt = open(out).read()
    for m in re.finditer(r"^(_ZN2fa\w+):.*?\n(.*?)\.end_amdhsa_kernel", txt, re.S | re.M):
        name, lines = m.group(1), m.group(2).split("\n")
        total = sum("v_mfma" in l for l in lines)
        n, inside, any_scratch = 0, [], 0
        for l in lines:
            n += "v_mfma" in l
            if "scratch_" in l:
                any_scratch += 1
                if "scratch_load" in l and 0 < n < total:
                    inside.append(n)
        if "fwd_kernel_fp8" in name:
            # no scratch at all (the pipeline state crosses tile boundaries through LDS, not through loop-carried registers
            # that hipcc would have to save around the generated block's fixed register map)
            assert any_scratch == 0, (name, any_scratch)
            body = "\n".join(lines)
            i0, i1 = body.index(".Lf8_t0_"), body.rindex(".Lf8_exit_")
            assert body[i0:i1].count("v_mfma") == 96, name  # 6 unrolled tiles x 16
        elif "fwd_kernel_w64" in name and name.endswith("ELb1EEEvNS_7KParamsE"):
            # the persistent form (PERSIST = true): hipcc keeps a few item-level values in scratch around the item switch; none
            # inside the generated blocks (the labels .Lfa_t0 .. .Lfa_exit bracket each one)
            body = "\n".join(lines)
            for blk in re.findall(r"\.Lfa_in1_\d+:.*?\.Lfa_exit_\d+:", body, re.S):
                assert "scratch_" not in blk, name
                assert blk.count("v_mfma") >= 3 * 64, name
        elif "fwd_kernel_w64" in name or "fwd_kernel_d256" in name:
            assert any_scratch == 0, (name, any_scratch)
        elif "fwd_kernel" in name:
            softcap = re.search(r"Li\d+ELi\d+ELb1", name) is not None
            assert len(inside) <= (1 if softcap else 0), (name, inside)
        elif "bwd_dkdv_kernel" in name and ".Lfb_exit_" in "\n".join(lines):
            # head dim 128, plain: the bulk of the sweep is the generated block (tools/gen_bwd_loop.py), whose fixed register map
            # leaves the C++ tile path (boundary tiles only) a few spilled loop invariants; none inside the block, 64 MFMAs per
            # unrolled tile
            body = "\n".join(lines)
            blk = body[body.index(".Lfb_t1_"):body.rindex(".Lfb_exit_")]
            # MFMAs per tile and wave (head dim 128 / 96 on the 128-wide tiles / 64), two unrolled tiles
            per_tile = 32 if "Li64ELi1E" in name else (48 if re.search(r"Li96E(Li\dE)?EEvNS_7BParamsE$", name) else 64)  # (DEFF, then PART)
            assert "scratch_" not in blk and blk.count("v_mfma") == 2 * per_tile, name
            assert any_scratch <= 24, (name, any_scratch)
        elif "bwd_dq_kernel" in name and ".Ldq_exit_" in "\n".join(lines):
            # head dim 128, plain: as above (tools/gen_bwd_dq_loop.py; all 256 AGPRs are operands of the block): 96 MFMAs per
            # unrolled tile x 3 LDS slots, 32 per prologue x 3, 16 per tail x 3
            body = "\n".join(lines)
            blk = body[body.index(".Ldq_p1_"):body.rindex(".Ldq_exit_")]
            assert "scratch_" not in blk, name
            # MFMAs of one dQ group (head dim 128 / 96 on the 128-wide tiles / 64)
            u = 8 if "Li64ELi2E" in name else (12 if name.endswith("Li96EEEvNS_7BParamsE") else 16)
            assert blk.count("v_mfma") == 3 * 2 * u + 3 * 6 * u + 3 * u, (name, blk.count("v_mfma"))
            assert any_scratch <= 64, (name, any_scratch)
        elif "bwd_" in name:
            # (head-dim tile 256 included since dV and dK are a launch each, PART 1 / 2: one pinned accumulator set per sweep)
            assert any_scratch == 0, (name, any_scratch)
