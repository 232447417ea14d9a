"""Developer aid: which work items of the persistent kernel differ from the non-persistent one (per batch, head, 64-row group)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib
lib = _lib.load()
b, s, h, causal = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 512, 16, 0)
torch.manual_seed(0)
q, k, v = (torch.randn(b, s, h, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
lib.fa_set_persist_mode(-1); o0, l0, _ = fa.flash_attn_func(q, k, v, causal=bool(causal), return_attn_probs=True)
lib.fa_set_persist_mode(1); o1, l1, _ = fa.flash_attn_func(q, k, v, causal=bool(causal), return_attn_probs=True)
err = (o0.float() - o1.float()).abs().amax(dim=-1)          # (b, s, h)
err = err.view(b, s // 64, 64, h).amax(dim=2)                # (b, s/64, h): per 64-row group (= wave of an m-block)
lerr = (l0 - l1).abs().view(b, h, s // 64, 64).amax(dim=-1)  # (b, h, s/64)
bad = (err > 0)
print("bad 64-row groups:", int(bad.sum()), "of", bad.numel(), " lse bad:", int((lerr > 0).sum()))
nmb = s // 256
for bi in range(min(b, 4)):
    for hi in range(min(h, 16)):
        row = "".join("X" if bad[bi, g, hi] else "." for g in range(s // 64))
        print(f"b{bi} h{hi}: {row}   maxerr {err[bi, :, hi].max().item():.3f}")
# tile order: tile = (batch*h_k + kvh) * per_kvh + r; m_block = nmb-1 - r/h_ratio  (h_ratio 1 here)
tiles = b * h * nmb
print("tiles", tiles)
# row / column permutation probes on (b0, h0)
A = o0[0, :, 0, :].float(); B = o1[0, :, 0, :].float()
d = torch.cdist(B[:128], A[:512])
mn, ix = d.min(dim=1)
print("rows 0..127 of persistent O: nearest non-persistent row and distance")
print([(i, int(ix[i]), round(float(mn[i]), 3)) for i in range(0, 128, 5)])
# 8-column chunk permutation: for row 0, which chunk of A[0] matches each chunk of B[0]
for row in (0, 1, 17, 40):
    m = []
    for c in range(16):
        dd = [(A[row, 8 * c2: 8 * c2 + 8] - B[row, 8 * c: 8 * c + 8]).abs().max().item() for c2 in range(16)]
        c2 = min(range(16), key=lambda x: dd[x])
        m.append((c2, round(dd[c2], 3)))
    print("row", row, "chunk map (src chunk, err):", m)
