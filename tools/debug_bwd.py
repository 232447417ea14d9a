"""Developer aid: per-tensor backward error on small problems (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from oracle import attention_ref as oracle

def run(b, sq, sk, h, hk, d, causal=False, dtype=torch.bfloat16, seed=0):
    torch.manual_seed(seed)
    q = torch.randn(b, sq, h, d, dtype=dtype); k = torch.randn(b, sk, hk, d, dtype=dtype)
    v = torch.randn(b, sk, hk, d, dtype=dtype); g = torch.randn(b, sq, h, d, dtype=dtype)
    ql, kl, vl = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = fa.flash_attn_func(ql, kl, vl, causal=causal)
    got = torch.autograd.grad(out, (ql, kl, vl), g.cuda())
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    oref = oracle.attention_ref(qr, kr, vr, causal=causal)[0]
    ref = torch.autograd.grad(oref, (qr, kr, vr), g)
    print(f"b{b} sq{sq} sk{sk} h{h}/{hk} d{d} causal={causal}: out err {(out.cpu().float()-oref.float()).abs().max():.3e}", end="  ")
    for n, a, r in zip(("dq", "dk", "dv"), got, ref):
        e = (a.cpu().float() - r.float()).abs()
        print(f"{n} err {e.max():.3e} (ref max {r.float().abs().max():.2f})", end="  ")
        if e.max() > 0.1:
            idx = (e > 0.1).nonzero()
            print(f"\n   bad {n}: {len(idx)} elems; rows {sorted(set(idx[:,1].tolist()))[:20]} heads {sorted(set(idx[:,2].tolist()))} d {sorted(set(idx[:,3].tolist()))[:20]}")
    print()

run(1, 32, 32, 1, 1, 64)
run(1, 64, 64, 1, 1, 64)
run(1, 128, 128, 1, 1, 64)
run(1, 128, 256, 1, 1, 128)
run(1, 200, 300, 2, 1, 64, causal=True)
run(2, 64, 96, 4, 2, 32)
run(1, 70, 90, 2, 2, 128, causal=True, dtype=torch.float16)
run(1, 64, 64, 1, 1, 128)
run(1, 64, 300, 1, 1, 128)
run(1, 300, 64, 1, 1, 128)
run(2, 1023, 1024, 4, 2, 128, causal=True)
