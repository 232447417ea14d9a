"""The generated asm loops (csrc/fa_fwd_loop_gen.h, csrc/fa_fwd_loop_fp8_gen.h, csrc/fa_bwd_loop_gen.h, csrc/fa_bwd_dq_loop_gen.h) are build artefacts of their
generators (tools/gen_fwd_loop.py, tools/gen_fwd_loop_fp8.py, tools/gen_bwd_loop.py, tools/gen_bwd_dq_loop.py), committed so that the GPU box needs no generation step: the committed
headers must be exactly what the generators emit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gen", ["gen_fwd_loop.py", "gen_fwd_loop_fp8.py", "gen_fwd_loop_d256.py", "gen_bwd_loop.py",
                                 "gen_bwd_dq_loop.py"])
def test_generated_header_is_current(gen):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, f"run `python tools/{gen}` and commit the header"


@pytest.mark.parametrize("header", ["fa_fwd_loop_gen.h", "fa_fwd_loop_fp8_gen.h", "fa_fwd_loop_d256_gen.h", "fa_bwd_loop_gen.h",
                                    "fa_bwd_dq_loop_gen.h"])
def test_generated_blocks_drain_the_matrix_pipe_at_their_exit(header):
    """tools/isa_hazards.py drops its MFMA hazard windows at every unconditional branch, so the edges between a generated
    block and the compiler-visible code around it are not scanned.  They are covered by construction instead: every
    block's single exit label is followed by `s_nop 15; s_nop 7` (>= the 18 wait states an MFMA result needs before a VALU /
    memory reader) inside the asm statement, and every asm statement has exactly one such exit."""
    import re
    text = open(os.path.join(ROOT, "flash_attention_annotated_amd", "csrc", header)).read()
    blocks = re.findall(r"asm volatile\((.*?)\n\s*:", text, re.S)
    assert blocks, header
    for body in blocks:
        exits = re.findall(r'"(\.L\w+_exit_%=):\\n"\n(.*?)$', body, re.S)
        assert len(exits) == 1, (header, len(exits))
        tail = exits[0][1]
        assert '"s_nop 15\\n\\t"' in tail and '"s_nop 7\\n\\t"' in tail, (header, tail[-300:])
        assert "v_mfma" not in tail, header   # nothing is issued to the matrix pipe behind the drain
