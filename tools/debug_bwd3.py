import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd.flash_attn_2_cuda as m
torch.manual_seed(0)
b, sq, sk, h, d = 1, 32, 32, 1, 64
for dt in (torch.bfloat16, torch.float16):
    q = torch.randn(b, sq, h, d, dtype=dt, device="cuda"); k = torch.randn(b, sk, h, d, dtype=dt, device="cuda")
    v = torch.randn(b, sk, h, d, dtype=dt, device="cuda")
    lse = torch.zeros(b, h, sq, device="cuda")
    for name, o, g in (("ones*ones", torch.ones_like(q), torch.ones_like(q)), ("2*3", torch.full_like(q, 2), torch.full_like(q, 3)),
                       ("arange*1", torch.arange(d, device="cuda", dtype=torch.float32).expand(b, sq, h, d).to(dt).contiguous(), torch.ones_like(q)),
                       ("1*arange", torch.ones_like(q), torch.arange(d, device="cuda", dtype=torch.float32).expand(b, sq, h, d).to(dt).contiguous())):
        dq, dk, dv, sd = m.bwd(g, q, k, v, o, lse, None, None, None, None, 0.0, 0.125, False, -1, -1, 0.0, False, None, None)
        print(dt, name, sd[0, 0, :4].tolist(), "expect", (o.float() * g.float()).sum(-1)[0, 0, 0].item())
