"""Developer aid: persistent vs hand-over form of the 256-row kernel over the reference's benchmark grid (interleaved, same process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib
lib = _lib.load()

def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]

shapes = [(32, 512, 16, 16), (16, 1024, 16, 16), (8, 2048, 16, 16), (4, 4096, 16, 16), (2, 8192, 16, 16), (1, 16384, 16, 16),
          (4, 8192, 16, 16), (4, 16384, 16, 16), (8, 4096, 32, 8), (2, 8192, 32, 8), (64, 512, 8, 8), (16, 2048, 32, 4)]
for causal in (False, True):
    for b, s, h, hk in shapes:
        q = torch.randn(b, s, h, 128, device="cuda", dtype=torch.bfloat16)
        k = torch.randn(b, s, hk, 128, device="cuda", dtype=torch.bfloat16)
        v = torch.randn(b, s, hk, 128, device="cuda", dtype=torch.bfloat16)
        fl = 4 * b * h * s * s * 128 * (0.5 if causal else 1.0)
        r = {-1: [], 1: []}
        for rep in range(3):
            for mode in (-1, 1):
                lib.fa_set_persist_mode(mode)
                r[mode].append(t(lambda: fa.flash_attn_func(q, k, v, causal=causal)))
        lib.fa_set_persist_mode(0)
        a, p_ = sorted(r[-1])[1], sorted(r[1])[1]
        print(f"b{b} s{s} h{h}/{hk} causal={int(causal)}: hand-over {a*1e3:8.1f} us {fl/a/1e9:6.0f} TF | persistent {p_*1e3:8.1f} us {fl/p_/1e9:6.0f} TF | {100*(a/p_-1):+5.1f} %", flush=True)
        del q, k, v
