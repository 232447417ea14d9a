"""Developer aid: host wall time per un-graphed decode step (flash_attn_with_kvcache, b = 1, hq32/hkv8 d128, cache 8192):
the figure the binding overhead shows up in.  Prints wall us per call (back-to-back launches, one sync at the end) and the
device time per step from HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

for b in (1, 8):
    q = torch.randn(b, 1, 32, 128, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(b, 8192, 8, 128, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(b, 8192, 8, 128, device="cuda", dtype=torch.bfloat16)
    kn = torch.randn(b, 1, 8, 128, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn(b, 1, 8, 128, device="cuda", dtype=torch.bfloat16)
    cs = torch.full((b,), 8000, dtype=torch.int32, device="cuda")
    for _ in range(20):
        fa.flash_attn_with_kvcache(q, kc, vc, kn, vn, cache_seqlens=cs)
    torch.cuda.synchronize()
    n = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fa.flash_attn_with_kvcache(q, kc, vc, kn, vn, cache_seqlens=cs)
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"b{b}: host issue {t_issue / n * 1e6:.1f} us/step, wall incl. drain {t_all / n * 1e6:.1f} us/step, device {e0.elapsed_time(e1) / n * 1e3:.1f} us/step")
