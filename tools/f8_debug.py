"""Developer aid: how often the generated fp8 block is left through its guards (needs a -DFA_F8_DEBUG build as FA_FWD_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flash_attention_annotated_amd import _lib, hopper_interface as fa3
lib = _lib.load()
lib.fa_debug_read_f8.argtypes = [ctypes.c_void_p]
buf = np.zeros(8, dtype=np.uint64)
for causal in (False, True):
    b, s, h, d = 4, 8192, 16, 128
    q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16).to(torch.float8_e4m3fn) for _ in range(3))
    lib.fa_debug_read_f8(buf.ctypes.data)
    fa3.flash_attn_func(q, k, v, causal=causal)
    lib.fa_debug_read_f8(buf.ctypes.data)
    waves = b * h * (s // 256) * 4
    print(f"causal={causal}: block runs {buf[0]} ({buf[0] / waves:.2f} per wave), tiles in blocks {buf[1]} ({buf[1] / waves:.1f} per wave of {s // 64}), "
          f"ended by pend {buf[2]}, by tripb {buf[3]}")
