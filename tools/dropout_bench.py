"""Time forward / backward with and without dropout on the C2 shape (b4 h16 d128 s8192 bf16).  Usage: python tools/dropout_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

b, s, h, d = 4, 8192, 16, 128
q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
g = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
flops = 4 * b * h * s * s * d
for p in (0.0, 0.1):
    for causal in (False, True):
        f = flops / (2 if causal else 1)
        t_f = timed(lambda: fa.flash_attn_func(q.detach(), k.detach(), v.detach(), p, causal=causal))
        out = fa.flash_attn_func(q, k, v, p, causal=causal)
        t_b = timed(lambda: torch.autograd.grad(out, (q, k, v), g, retain_graph=True))
        print(f"p={p} causal={causal}: fwd {t_f:.3f} ms = {f / t_f / 1e9:.0f} TF   bwd {t_b:.3f} ms = {2.5 * f / t_b / 1e9:.0f} TF", flush=True)
