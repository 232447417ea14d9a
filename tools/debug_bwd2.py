import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd.flash_attn_2_cuda as m
torch.manual_seed(0)
b, sq, sk, h, d = 1, 32, 32, 1, 64
q = torch.randn(b, sq, h, d, dtype=torch.bfloat16, device="cuda"); k = torch.randn(b, sk, h, d, dtype=torch.bfloat16, device="cuda")
v = torch.randn(b, sk, h, d, dtype=torch.bfloat16, device="cuda"); g = torch.randn(b, sq, h, d, dtype=torch.bfloat16, device="cuda")
sc = d ** -0.5
out, lse, _, _ = m.fwd(q, k, v, None, None, 0.0, sc, False, -1, -1, 0.0, False, None)
dq, dk, dv, sd = m.bwd(g, q, k, v, out, lse, None, None, None, None, 0.0, sc, False, -1, -1, 0.0, False, None, None)
D_ref = (g.float() * out.float()).sum(-1).permute(0, 2, 1)  # b h s
print("softmax_d err", (sd[:, :, :sq] - D_ref).abs().max().item(), "lse sample", lse[0,0,:4].tolist())
qf, kf, vf, gf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v, g))
S = qf @ kf.transpose(-1, -2) * sc
P = torch.exp(S - lse[..., None])
dP = gf @ vf.transpose(-1, -2)
dS = P * (dP - D_ref[..., None])
dq_t = (dS @ kf) * sc; dk_t = (dS.transpose(-1, -2) @ qf) * sc; dv_t = P.transpose(-1, -2) @ gf
for n, a, r in (("dq", dq, dq_t), ("dk", dk, dk_t), ("dv", dv, dv_t)):
    e = (a.float().permute(0, 2, 1, 3) - r).abs()
    print(n, "err", e.max().item())
# isolate: what dq would be if D were 0 / if dP were 0
dq_noD = ((P * dP) @ kf) * sc
dq_noP = ((P * (-D_ref[..., None])) @ kf) * sc
print("dq vs noD", (dq.float().permute(0,2,1,3) - dq_noD).abs().max().item(), "vs noP", (dq.float().permute(0,2,1,3) - dq_noP).abs().max().item())
print("sd   ", sd[0, 0, :6].tolist())
print("D_ref", D_ref[0, 0, :6].tolist())
part = (g.float() * out.float())[0, :, 0, :].reshape(sq, 8, 8).sum(-1)  # per 8-element chunk
print("chunk sums row0", part[0].tolist())
print("sd beyond sq", sd[0, 0, sq:sq+4].tolist())
cands = {"g*out": g*out.float(), "g*q": g.float()*q.float(), "g*k": g.float()*k.float(), "g*v": g.float()*v.float(), "out*out": out.float()*out.float(),
         "g*g": g.float()*g.float(), "q*out": q.float()*out.float()}
for n, t in cands.items():
    print(n, t.float().sum(-1)[0, :3, 0].tolist())
print(out.stride(), g.stride(), out.dtype, g.dtype, out.data_ptr() % 16, g.data_ptr() % 16)
