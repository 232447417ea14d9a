"""Developer aid: forward throughput per head dim (64, 96, 128, 160, 192, 256) at 16k tokens, model dim 2048, bf16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
def t(f, n=15):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]
for d in (64, 96, 128, 160, 192, 256):
    for causal in (False, True):
        s = 8192
        b, h = 16384 // s, 2048 // d
        q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        ms = t(lambda: fa.flash_attn_func(q, k, v, causal=causal))
        fl = 4 * b * h * s * s * d / (2 if causal else 1)
        print(f"d{d:3d} causal={int(causal)} s{s} b{b} h{h:2d}: {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
