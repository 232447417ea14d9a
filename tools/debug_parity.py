"""Developer diagnostic: run a few small shapes on the GPU and print where the HIP output deviates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from oracle import attention_ref as oracle

dev = "cuda"
def run(b, sq, sk, h, hk, d, causal, dtype=torch.bfloat16, window=(-1, -1)):
    torch.manual_seed(0)
    q = torch.randn(b, sq, h, d, dtype=dtype); k = torch.randn(b, sk, hk, d, dtype=dtype); v = torch.randn(b, sk, hk, d, dtype=dtype)
    out, lse, _ = fa.flash_attn_func(q.to(dev), k.to(dev), v.to(dev), causal=causal, window_size=window, return_attn_probs=True)
    torch.cuda.synchronize()
    ref, _, lse_ref = oracle.attention_ref(q, k, v, causal=causal, window_size=window, return_lse=True)
    pt, _ = oracle.attention_ref(q, k, v, causal=causal, window_size=window, upcast=False, reorder_ops=True)
    e = (out.float().cpu() - ref.float()).abs()
    bound = 2 * (pt.float() - ref.float()).abs().max().item() + 1e-5
    le = (lse.cpu() - lse_ref)
    le = le[torch.isfinite(lse_ref)].abs().max().item() if torch.isfinite(lse_ref).any() else 0
    status = "OK " if e.max().item() <= bound else "BAD"
    print(f"{status} b{b} sq{sq} sk{sk} h{h}/{hk} d{d} causal={causal} win={window} {dtype}: err {e.max().item():.3e} bound {bound:.3e} lse_err {le:.2e}")
    if status == "BAD":
        idx = (e > bound).nonzero()
        print("   bad count", idx.shape[0], "of", e.numel(), "first", idx[:5].tolist())
        rows = torch.unique(idx[:, 1]); cols = torch.unique(idx[:, 3])
        print("   bad rows", rows[:40].tolist(), "... bad d-cols", cols[:40].tolist())
        bi, r, hh, c = idx[0].tolist()
        print("   got", out[bi, r, hh, :8].float().cpu().tolist()); print("   ref", ref[bi, r, hh, :8].float().tolist())

for args in [(1, 32, 64, 1, 1, 128, False), (1, 256, 64, 1, 1, 128, False), (1, 256, 512, 2, 1, 128, False),
             (2, 113, 203, 4, 2, 128, False), (2, 113, 203, 4, 2, 128, True), (1, 256, 256, 2, 2, 64, False),
             (1, 300, 300, 2, 2, 64, True), (1, 130, 150, 2, 1, 256, True), (1, 512, 512, 2, 2, 128, True),
             (1, 2048, 2048, 2, 2, 128, False)]:
    run(*args)
run(1, 256, 512, 2, 1, 128, False, dtype=torch.float16)
run(1, 200, 200, 2, 2, 64, False, window=(64, 0))
