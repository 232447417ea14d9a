#!/usr/bin/env python3
"""Sliding-window forward timing (Mistral-style local attention): b2 h16 d128, window (W, 0) over s = 16384."""
import sys
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

for d, W in ((128, 4096), (128, 1024), (64, 4096)):
    b, h, s = 2, 16 if d == 128 else 32, 16384
    q, k, v = (torch.randn(b, s, h, d, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    for _ in range(3):
        fa.flash_attn_func(q, k, v, window_size=(W, 0))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fa.flash_attn_func(q, k, v, window_size=(W, 0))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    # visible (row, key) pairs: sum over rows of min(row + 1, W + 1)
    pairs = sum(min(r + 1, W + 1) for r in range(s))
    print(f"d{d:3d} s{s} window ({W}, 0) b{b} h{h}: {ms:7.3f} ms  {4 * b * h * d * pairs / ms / 1e9:6.0f} TF")
