import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from oracle import attention_ref as oracle
dev = "cuda"
def run(sq, sk, d, softcap, qs, causal=False, h=2):
    torch.manual_seed(0)
    q = (torch.randn(1, sq, h, d) * qs).bfloat16(); k = torch.randn(1, sk, h, d).bfloat16(); v = torch.randn(1, sk, h, d).bfloat16()
    out = fa.flash_attn_func(q.to(dev), k.to(dev), v.to(dev), softcap=softcap, causal=causal)
    ref, _ = oracle.attention_ref(q, k, v, softcap=softcap, causal=causal)
    pt, _ = oracle.attention_ref(q, k, v, softcap=softcap, causal=causal, upcast=False, reorder_ops=True)
    e = (out.float().cpu() - ref.float()).abs()
    bound = 3 * (pt.float() - ref.float()).abs().max().item() + 1e-3
    bad = (e > bound)
    rows = torch.unique(bad.nonzero()[:, 1])
    print(f"sq{sq} sk{sk} d{d} softcap{softcap} qs{qs} causal{causal}: err {e.max().item():.3e} bound {bound:.3e} badrows {rows[:24].tolist()} n={rows.numel()}")
run(128, 160, 64, 30.0, 7.5)
run(128, 160, 128, 30.0, 7.5)
run(128, 64, 64, 30.0, 7.5)
run(128, 128, 64, 30.0, 7.5)
run(32, 128, 64, 30.0, 7.5)
run(128, 160, 64, 0.0, 7.5)
run(128, 160, 64, 0.0, 20.0)
run(256, 512, 64, 0.0, 20.0)
run(256, 512, 128, 0.0, 20.0)
run(256, 512, 128, 30.0, 7.5, True)
