#!/usr/bin/env python3
"""Forward / backward rate per feature (one line each): finds the shapes that fall off the generated loops.
b4 s4096 (16k tokens), h = 2048 / d, bf16; TFLOP/s over the visible (row, key) pairs; backward = 2.5 x forward FLOPs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

def timeit(f, n=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def visible(sq, sk, causal, window):
    if not causal and window == (-1, -1): return sq * sk
    wl, wr = window
    if causal: wr = 0
    tot = 0
    for i in range(sq):
        d = i + sk - sq
        lo = 0 if wl < 0 else max(0, d - wl)
        hi = sk - 1 if wr < 0 else min(sk - 1, d + wr)
        tot += max(0, hi - lo + 1)
    return tot

dims = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (64, 128, 256)
feats = [("plain", {}), ("causal", dict(causal=True)), ("gqa4 causal", dict(causal=True, gqa=4)), ("softcap", dict(softcap=30.0)),
         ("alibi causal", dict(causal=True, alibi=True)), ("alibi geometric causal", dict(causal=True, alibi="geom")), ("window(1024,0)", dict(window_size=(1024, 0))),
         ("window(512,512)", dict(window_size=(512, 512))), ("varlen causal", dict(causal=True, varlen=True)),
         ("sk=4000 (ragged)", dict(sk=4000)), ("sq=1024 sk=4096 causal", dict(causal=True, sq=1024))]
for d in dims:
    for name, kw in feats:
        kw = dict(kw)
        b, s = 4, 4096
        h = 2048 // d
        sq, sk = kw.pop("sq", s), kw.pop("sk", s)
        hk = h // kw.pop("gqa", 1)
        varlen = kw.pop("varlen", False)
        al = kw.pop("alibi", False)
        if al == "geom":   # the slopes ALiBi models use: 2^(-8 (i + 1) / h)
            kw["alibi_slopes"] = torch.tensor([2.0 ** (-8.0 * (i + 1) / h) for i in range(h)], device="cuda", dtype=torch.float32)
        elif al:           # the reference's tests: rand * 0.3 (tests/test_flash_attn.py:937)
            kw["alibi_slopes"] = torch.rand(h, device="cuda") * 0.3
        q = torch.randn(b, sq, h, d, dtype=torch.bfloat16, device="cuda", requires_grad=True)
        k = torch.randn(b, sk, hk, d, dtype=torch.bfloat16, device="cuda", requires_grad=True)
        v = torch.randn(b, sk, hk, d, dtype=torch.bfloat16, device="cuda", requires_grad=True)
        if varlen:
            cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device="cuda")
            q2, k2, v2 = (t.detach().reshape(b * s, -1, d).requires_grad_(True) for t in (q, k, v))
            fwd = lambda: fa.flash_attn_varlen_func(q2, k2, v2, cu, cu, s, s, **kw)
            leaves = (q2, k2, v2)
        else:
            fwd = lambda: fa.flash_attn_func(q, k, v, **kw)
            leaves = (q, k, v)
        tf = timeit(fwd)
        out = fwd()
        g = torch.randn_like(out)
        tb = timeit(lambda: torch.autograd.grad(out, leaves, g, retain_graph=True), n=4)
        pairs = visible(sq, sk, kw.get("causal", False), kw.get("window_size", (-1, -1)))
        fl = 4.0 * b * h * d * pairs
        print(f"d{d:3d} {name:26s} fwd {tf:7.3f} ms {fl / tf / 1e9:6.0f} TF   bwd {tb:7.3f} ms {2.5 * fl / tb / 1e9:6.0f} TF", flush=True)
