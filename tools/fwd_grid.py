"""Developer aid: forward TFLOP/s over the reference's benchmark grid (benchmarks/benchmark_flash_attention.py:71-79:
batch * seqlen = 16k tokens, model dim 2048, seqlen 512..16k, head dim 64/128, causal or not), for each kernel shape
(0 = library policy, 1 = 8 waves x 32 rows, 2 = 4 waves x 32 rows).  GPU only.  Usage: python tools/fwd_grid.py [variants]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib

def t(f, n=15):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]

variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
lib = _lib.load()
for d in (64, 128):
    for causal in (False, True):
        for s in (512, 1024, 2048, 4096, 8192, 16384):
            b, h = 16384 // s, 2048 // d
            q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
            fl = 4 * b * h * s * s * d / (2 if causal else 1)
            res = []
            for var in variants:
                lib.fa_set_default_variant(var)
                ms = t(lambda: fa.flash_attn_func(q, k, v, causal=causal))
                res.append(f"v{var}: {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF")
            lib.fa_set_default_variant(0)
            print(f"d{d:3d} causal={int(causal)} s{s:5d} b{b:2d} h{h:2d}  " + "   ".join(res), flush=True)
