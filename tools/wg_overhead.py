"""Developer aid: per-workgroup fixed cost of the forward = time of a launch whose workgroups have 1..N key tiles."""
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

b, h, d, sq = 4, 16, 128, 8192
q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
for sk in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    k = torch.randn(b, sk, h, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, h, d, device="cuda", dtype=torch.bfloat16)
    ms = t(lambda: fa.flash_attn_func(q, k, v))
    wgs = b * h * sq // 256
    print(f"sk={sk:5d} tiles/WG={sk//64:4d}: {ms*1e3:8.1f} us  -> {ms*1e3/(wgs/256):7.2f} us per WG round, {ms*1e3/(wgs/256)/(sk//64):6.2f} us per tile")
