import os, sys, torch
sys.path.insert(0, os.getcwd())
from flash_attention_annotated_amd import hopper_interface as fa3
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (d, dv, causal, chunk) in [(192,128,False,0),(192,128,True,0),(192,192,False,0),(192,192,True,0),(64,512,False,0),(128,128,True,2048),(128,128,True,0)]:
    b, s = 2, 8192
    h = 16
    q = torch.randn(b, s, h, d, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(b, s, h, d, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(b, s, h, dv, dtype=torch.bfloat16, device="cuda")
    ms = t(lambda: fa3.flash_attn_func(q, k, v, causal=causal, attention_chunk=chunk))
    fl = 2 * b * h * s * s * (d + dv) / (2 if causal else 1)
    if chunk: fl = 2 * b * h * (d + dv) * sum(min(i % chunk + 1, chunk) for i in range(s))
    print(f"d{d} dv{dv} causal={int(causal)} chunk={chunk}: {ms:7.3f} ms {fl/ms/1e9:6.0f} TF")
