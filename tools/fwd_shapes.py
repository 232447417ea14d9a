"""Developer aid: forward TFLOP/s over head dims / dtypes / masks at seq 8192 (reference benchmark grid,
benchmarks/benchmark_flash_attention.py:71-79: batch * seqlen = 16k-32k tokens, dim 2048).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]

for dtype in (torch.bfloat16, torch.float16):
    for d in (64, 128, 256):
        for causal in (False, True):
            for s in (2048, 8192):
                b, h = 32768 // s, 2048 // d
                q = torch.randn(b, s, h, d, device="cuda", dtype=dtype)
                k = torch.randn(b, s, h, d, device="cuda", dtype=dtype)
                v = torch.randn(b, s, h, d, device="cuda", dtype=dtype)
                ms = t(lambda: fa.flash_attn_func(q, k, v, causal=causal))
                fl = 4 * b * h * s * s * d / (2 if causal else 1)
                print(f"{str(dtype)[6:]:8s} d{d:3d} causal={int(causal)} s{s:5d} b{b:2d} h{h:2d}: {ms:7.3f} ms {fl/ms/1e9:6.0f} TF")
