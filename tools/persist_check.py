"""Developer aid: the persistent form of the 256-row kernel against the non-persistent one (bit-for-bit: same arithmetic in
the same order) and against the fp32 oracle, plus their times.  FA_FWD_PERSIST is switched through the test hook."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib

lib = _lib.load()

def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]

shapes = [(2, 512, 512, 4, 4, False), (2, 512, 512, 4, 4, True), (1, 768, 1024, 8, 2, True), (3, 1000, 1024, 5, 5, False),
          (32, 512, 512, 16, 16, False), (16, 1024, 1024, 16, 16, False), (8, 2048, 2048, 16, 16, False), (8, 2048, 2048, 16, 16, True),
          (4, 8192, 8192, 16, 16, False), (2, 8192, 8192, 16, 16, True), (2, 4096, 4096, 32, 8, True), (1, 300, 4096, 16, 4, True)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:4]
bad = 0
for b, sq, sk, h, hk, causal in shapes:
    torch.manual_seed(b * 1000 + sq)
    q = torch.randn(b, sq, h, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, 128, device="cuda", dtype=torch.bfloat16)
    res = {}
    for mode in (-1, 1):
        lib.fa_set_persist_mode(mode)
        o, lse, _ = fa.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
        ms = t(lambda: fa.flash_attn_func(q, k, v, causal=causal))
        res[mode] = (o.clone(), lse.clone(), ms)
    lib.fa_set_persist_mode(0)
    same = torch.equal(res[-1][0], res[1][0]) and torch.equal(res[-1][1], res[1][1])
    diff = (res[-1][0].float() - res[1][0].float()).abs().max().item()
    nan = torch.isnan(res[1][0]).any().item()
    fl = 4 * b * h * sq * sk * 128 * (0.5 if causal else 1.0) if sq == sk else 4 * b * h * sq * sk * 128
    bad += (not same)
    print(f"b{b} sq{sq} sk{sk} h{h}/{hk} causal={int(causal)}: identical={same} maxdiff={diff:.3e} nan={nan}  "
          f"non-persistent {res[-1][2]*1e3:8.1f} us ({fl/res[-1][2]/1e9:6.0f} TF)  persistent {res[1][2]*1e3:8.1f} us ({fl/res[1][2]/1e9:6.0f} TF)", flush=True)
print("MISMATCHES", bad)
