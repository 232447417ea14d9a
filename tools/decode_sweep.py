"""Developer aid: decode step time vs batch, split-KV off (num_splits=1) and heuristic (0).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

h, hk, d, s = 32, 8, 128, 8192
for b in (1, 2, 4, 8, 16, 32):
    q = torch.randn(b, 1, h, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
    cs = torch.full((b,), s, dtype=torch.int32, device="cuda")
    row = []
    for ns in (1, 0):
        f = lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cs, num_splits=ns)
        for _ in range(5): f()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, e in ev:
            a.record(); f(); e.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(e) for a, e in ev)[len(ev) // 2]
        row.append((ms, 2 * b * s * hk * d * 2 / ms / 1e6))
    print(f"b={b:3d}  unsplit {row[0][0]*1e3:7.1f} us {row[0][1]:7.0f} GB/s   heuristic {row[1][0]*1e3:7.1f} us {row[1][1]:7.0f} GB/s")
