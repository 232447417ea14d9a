"""Developer aid: forward / backward time for a few (d, seqlen) shapes.  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_attention_annotated_amd import flash_attn_2_cuda as ext

def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]

for (b, s, h, hk, d, causal) in [(2, 4096, 16, 16, 64, False), (2, 4096, 16, 16, 128, False), (2, 4096, 8, 8, 256, False),
                                 (2, 4096, 16, 4, 128, True), (2, 4096, 16, 16, 96, False)]:
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    sc = d ** -0.5
    out, lse, _, _ = ext.fwd(q, k, v, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    tf = t(lambda: ext.fwd(q, k, v, out, None, 0.0, sc, causal, -1, -1, 0.0, False, None))
    tb = t(lambda: ext.bwd(g, q, k, v, out, lse, dq, dk, dv, None, 0.0, sc, causal, -1, -1, 0.0, False, None, None))
    fl = 4 * b * h * s * s * d / (2 if causal else 1)
    print(f"b{b} s{s} h{h}/{hk} d{d} causal={causal}: fwd {tf:.3f} ms {fl/tf/1e9:.0f} TF   bwd {tb:.3f} ms {2.5*fl/tb/1e9:.0f} TF")
