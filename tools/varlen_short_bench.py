"""Developer aid: what the padded varlen grid costs on many short sequences.
256 sequences of 64 .. 512 tokens, hq32 / hkv8, d128, bf16, causal and not.  The grid is batch x ceil(max_seqlen / 256) m-blocks
x heads with an early exit for m-blocks past a sequence's end; a tile list (hopper/flash_prepare_scheduler.cu) would launch only
the real tiles.  Reported: time of the ragged batch, the share of workgroups that exit at once, and the time of the SAME
sequences when the launch is told max_seqlen = 256 for the short ones (two launches: lengths <= 256 with a 1-m-block grid,
the rest with 2) -- a grid without dead workgroups built from the existing entry point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa

def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]

def batch(lens, h, hk, d):
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    tot = int(cu[-1])
    q = torch.randn(tot, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(tot, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(tot, hk, d, device="cuda", dtype=torch.bfloat16)
    return q, k, v, cu

g = torch.Generator().manual_seed(0)
lens = torch.randint(64, 513, (256,), generator=g).tolist()
h, hk, d = 32, 8, 128
for causal in (False, True):
    q, k, v, cu = batch(lens, h, hk, d)
    fl = sum(4 * h * d * L * L for L in lens) / (2 if causal else 1)
    ms = t(lambda: fa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=causal))
    tiles_grid = len(lens) * ((max(lens) + 255) // 256)
    tiles_real = sum((L + 255) // 256 for L in lens)
    short = [L for L in lens if L <= 256]
    long_ = [L for L in lens if L > 256]
    qs, ks, vs, cus = batch(short, h, hk, d)
    ql, kl, vl, cul = batch(long_, h, hk, d)
    def two():
        fa.flash_attn_varlen_func(qs, ks, vs, cus, cus, max(short), max(short), causal=causal)
        fa.flash_attn_varlen_func(ql, kl, vl, cul, cul, max(long_), max(long_), causal=causal)
    ms2 = t(two)
    print(f"causal={int(causal)}: 256 sequences 64..512 (mean {sum(lens) / len(lens):.0f}), hq32/hkv8 d128: padded grid {ms:.4f} ms = {fl / ms / 1e9:.0f} TFLOP/s; "
          f"m-block slots {tiles_grid}, real {tiles_real} ({100 * (1 - tiles_real / tiles_grid):.0f} % exit at once); "
          f"two launches without dead workgroups {ms2:.4f} ms ({100 * (ms / ms2 - 1):+.1f} % for the padded grid, incl. the second launch's ~5 us)", flush=True)
