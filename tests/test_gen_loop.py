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
