#!/usr/bin/env python3
"""Rates of inference-side shapes (one line each): chunked prefill over a cache (dense and paged), fp8 at head dims other than
128, decode at a few batch sizes.  TFLOP/s over visible pairs (prefill) or TB/s of cache bytes (decode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import hopper_interface as fa3

def timeit(f, n=8):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = "cuda"
hq, hk, d = 32, 8, 128
for b, sq, cache in ((8, 512, 16384), (4, 2048, 8192), (16, 128, 8192)):
    q = torch.randn(b, sq, hq, d, dtype=torch.bfloat16, device=dev)
    kc = torch.randn(b, cache, hk, d, dtype=torch.bfloat16, device=dev)
    vc = torch.randn(b, cache, hk, d, dtype=torch.bfloat16, device=dev)
    lens = torch.full((b,), cache, dtype=torch.int32, device=dev)
    ms = timeit(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True))
    pairs = sum(cache - sq + i + 1 for i in range(sq))
    fl = 4.0 * b * hq * d * pairs
    print(f"chunked prefill b{b} q{sq} cache{cache} dense: {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
    page = 256
    npg = cache // page
    kp = kc.reshape(b * npg, page, hk, d)
    vp = vc.reshape(b * npg, page, hk, d)
    bt = torch.arange(b * npg, dtype=torch.int32, device=dev).reshape(b, npg)
    ms = timeit(lambda: fa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=bt, causal=True))
    print(f"chunked prefill b{b} q{sq} cache{cache} paged256: {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
FP8 = torch.float8_e4m3fn
for dd in (64, 128, 256):
    b, s, h = 4, 4096, 2048 // dd
    q, k, v = (torch.randn(b, s, h, dd, dtype=torch.bfloat16, device=dev).to(FP8) for _ in range(3))
    for causal in (False, True):
        ms = timeit(lambda: fa3.flash_attn_func(q, k, v, causal=causal))
        fl = 4.0 * b * h * dd * s * s / (2 if causal else 1)
        print(f"fp8 d{dd} s{s} causal={int(causal)}: {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
for b in (1, 16, 64, 128):
    cache = 4096
    q = torch.randn(b, 1, hq, d, dtype=torch.bfloat16, device=dev)
    kc = torch.randn(b, cache, hk, d, dtype=torch.bfloat16, device=dev)
    vc = torch.randn(b, cache, hk, d, dtype=torch.bfloat16, device=dev)
    lens = torch.full((b,), cache, dtype=torch.int32, device=dev)
    ms = timeit(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens), n=20)
    by = 2.0 * b * cache * hk * d * 2
    print(f"decode b{b} cache{cache}: {ms * 1e3:7.1f} us {by / ms / 1e9:6.2f} TB/s", flush=True)
