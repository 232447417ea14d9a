#!/usr/bin/env python3
"""Backward timing per head dim (dq + dk + dv from dout; reference FLOP convention 2.5 x forward): s = 8192, b 2, h = 2048 / d."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

DIMS = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (64, 96, 128, 160, 192, 256)
for d in DIMS:
    for causal in (False, True):
        b, s = 2, 8192
        h = 2048 // d
        q, k, v = (torch.randn(b, s, h, d, dtype=torch.bfloat16, device="cuda", requires_grad=True) for _ in range(3))
        out = fa.flash_attn_func(q, k, v, causal=causal)
        g = torch.randn_like(out)
        for _ in range(2):
            torch.autograd.grad(out, (q, k, v), g, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.autograd.grad(out, (q, k, v), g, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.5 * 4 * b * h * s * s * d / (2 if causal else 1)
        print(f"d{d:3d} causal={int(causal)} s{s} b{b} h{h}: {ms:7.3f} ms  {fl / ms / 1e9:6.0f} TF")
