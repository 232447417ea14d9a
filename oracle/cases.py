"""ORACLE — test infrastructure only.  Seeded input recipes shared by oracle/make_golden.py
(which freezes the reference's outputs for them) and tests/ (which replay them).

Shapes follow the reference's own test matrix in miniature (tests/test_flash_attn.py:878-919:
odd (sq, sk) pairs such as (113, 203), mha/mqa/gqa, causal x local, random padding masks with
lengths in [max-20, max] :58-71; the (2,5)/(5,2) masks of the flash_attn_func docstring
flash_attn/flash_attn_interface.py:1164-1174) plus BASELINE config 1 (b2 h4 d64 s512 fp32).
"""
import torch

_DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def _case(dtype, b, sq, sk, h, hk, d, causal=False, window=(-1, -1), softcap=0.0, padding="none", seed=0,
          q_scale=1.0, store_row_stride=0, fp8=False, alibi=False):
    if store_row_stride == 0:  # keep fixtures small: store every n-th query row of the big cases
        store_row_stride = max(1, (b * sq * h * d) // 12288)
    return dict(dtype=dtype, b=b, sq=sq, sk=sk, h=h, hk=hk, d=d, causal=causal, window=tuple(window),
                softcap=softcap, padding=padding, seed=seed, q_scale=q_scale, store_row_stride=store_row_stride,
                fp8=fp8, alibi=alibi)


CASES = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "c1_fp32_b2_s512_h4_d64": _case("fp32", 2, 512, 512, 4, 4, 64, store_row_stride=16),
    "c1_fp32_b2_s512_h4_d64_causal": _case("fp32", 2, 512, 512, 4, 4, 64, causal=True, store_row_stride=16),
    # docstring mask diagrams (bottom-right alignment, fully masked rows -> 0)
    "doc_causal_sq2_sk5": _case("bf16", 1, 2, 5, 2, 2, 32, causal=True, seed=1),
    "doc_causal_sq5_sk2": _case("bf16", 1, 5, 2, 2, 2, 32, causal=True, seed=2),
    # dense, mha / gqa / mqa, odd lengths, both 16-bit types
    "bf16_dense_128_d64": _case("bf16", 2, 128, 128, 4, 4, 64, seed=3),
    "fp16_dense_128_d64": _case("fp16", 2, 128, 128, 4, 4, 64, seed=4),
    "bf16_113_203_gqa_d64": _case("bf16", 2, 113, 203, 6, 2, 64, seed=5),
    "bf16_113_203_gqa_d64_causal": _case("bf16", 2, 113, 203, 6, 2, 64, causal=True, seed=6),
    "bf16_203_113_mqa_d128_causal": _case("bf16", 1, 203, 113, 4, 1, 128, causal=True, seed=7),
    "fp16_256_384_d128": _case("fp16", 1, 256, 384, 2, 2, 128, seed=8),
    "bf16_d40": _case("bf16", 1, 108, 256, 2, 2, 40, causal=True, seed=9),
    "bf16_d96": _case("bf16", 1, 99, 130, 2, 1, 96, seed=10),
    "bf16_d256_causal": _case("bf16", 1, 130, 150, 2, 1, 256, causal=True, seed=11),
    # sliding windows
    "bf16_local_64_0": _case("bf16", 1, 200, 200, 2, 2, 64, window=(64, 0), seed=12),
    "bf16_local_16_16_sq_ne_sk": _case("bf16", 1, 113, 203, 2, 2, 64, window=(16, 16), seed=13),
    "bf16_local_0_32": _case("bf16", 1, 150, 120, 2, 1, 64, window=(0, 32), seed=14),
    # softcap (scores pushed into the tanh knee like hopper/test_flash_attn.py:139-140)
    "bf16_softcap30": _case("bf16", 1, 128, 160, 2, 2, 64, softcap=30.0, q_scale=7.5, seed=15),
    # ragged batches (key / query padding)
    "bf16_padded_causal": _case("bf16", 3, 128, 217, 4, 2, 64, causal=True, padding="random", seed=16),
    "fp16_padded": _case("fp16", 3, 97, 97, 2, 2, 128, padding="random", seed=17),
    # ALiBi: slopes rand(b, h) * 0.3 as tests/test_flash_attn.py:937
    "bf16_alibi_113_203_gqa": _case("bf16", 2, 113, 203, 4, 2, 64, seed=20, alibi=True),
    "bf16_alibi_causal_d128": _case("bf16", 2, 150, 150, 2, 2, 128, causal=True, seed=21, alibi=True),
    "fp16_alibi_padded_local": _case("fp16", 2, 128, 160, 4, 4, 64, window=(40, 8), padding="random", seed=22,
                                     alibi=True),
    # fp8 e4m3 storage (BASELINE config 5 in miniature): bf16 values rounded through e4m3 + per-(batch, kv head)
    # descales rand*2, exactly how hopper/test_flash_attn.py:135-147 builds its fp8 inputs
    "fp8_descale_gqa_d128": _case("bf16", 2, 160, 200, 4, 2, 128, seed=18, fp8=True),
    "fp8_descale_causal_d64": _case("bf16", 2, 130, 130, 4, 4, 64, causal=True, seed=19, fp8=True),
}


def _fa3_case(dtype, b, sq, sk, h, hk, d, dv=0, chunk=0, **kw):
    c = _case(dtype, b, sq, sk, h, hk, d, **kw)
    c["dv"], c["chunk"] = (dv or d), chunk
    return c


# FA3-only forward arguments (the `dv` / `attention_chunk` axes of hopper/test_flash_attn.py:120-131): pinned to the FA3
# oracle hopper/test_util.py:226-348 (construct_chunk_mask :193-223), frozen in tests/golden/attention_fa3_golden.pt.
FA3_CASES = {
    "fa3_chunk100_gqa_d64": _fa3_case("bf16", 2, 256, 256, 4, 2, 64, chunk=100, seed=60),
    "fa3_chunk37_causal_113_203": _fa3_case("bf16", 2, 113, 203, 4, 4, 64, chunk=37, causal=True, seed=61),
    "fa3_chunk64_local_40_10_d128": _fa3_case("fp16", 1, 200, 200, 2, 2, 128, chunk=64, window=(40, 10), seed=62),
    "fa3_chunk50_causal_sq_gt_sk": _fa3_case("bf16", 1, 300, 128, 2, 1, 64, chunk=50, causal=True, seed=63),
    "fa3_chunk500_no_effect": _fa3_case("bf16", 1, 128, 160, 2, 2, 64, chunk=500, seed=64),
    "fa3_chunk1": _fa3_case("bf16", 1, 70, 90, 2, 2, 64, chunk=1, seed=65),
    "fa3_chunk48_padded": _fa3_case("bf16", 3, 128, 217, 4, 2, 64, chunk=48, padding="random", seed=66),
    "fa3_chunk96_d256_causal": _fa3_case("bf16", 1, 260, 300, 2, 1, 256, chunk=96, causal=True, seed=67),
    "fa3_dv128_d192_gqa_causal": _fa3_case("bf16", 2, 130, 190, 4, 2, 192, dv=128, causal=True, seed=68),
    "fa3_dv128_d160": _fa3_case("fp16", 1, 150, 150, 2, 2, 160, dv=128, seed=69),
    "fa3_dv256_d64": _fa3_case("bf16", 1, 140, 200, 2, 1, 64, dv=256, seed=70),
    "fa3_dv512_d64_local": _fa3_case("bf16", 1, 128, 128, 2, 2, 64, dv=512, window=(50, 20), seed=71),
    "fa3_dv320_d32": _fa3_case("bf16", 1, 65, 100, 2, 2, 32, dv=320, causal=True, seed=72),
    "fa3_dv128_d192_chunk80_padded": _fa3_case("bf16", 2, 160, 160, 4, 2, 192, dv=128, chunk=80, padding="random", seed=73),
}


# Backward fixtures: small problems whose dq/dk/dv (reference oracle + autograd, the way tests/test_flash_attn.py:1071-1105
# obtains dq_ref / dq_pt) are frozen in tests/golden/attention_grad_golden.pt.  Stored in full (store_row_stride = 1).
GRAD_CASES = {
    "grad_bf16_gqa_d32": _case("bf16", 2, 64, 96, 4, 2, 32, seed=40, store_row_stride=1),
    "grad_fp16_causal_sq_ne_sk_d64": _case("fp16", 1, 100, 130, 2, 2, 64, causal=True, seed=41, store_row_stride=1),
    "grad_bf16_causal_sq_gt_sk": _case("bf16", 1, 96, 40, 2, 1, 64, causal=True, seed=42, store_row_stride=1),
    "grad_bf16_local_softcap": _case("bf16", 1, 128, 128, 2, 2, 64, window=(30, 10), softcap=20.0, q_scale=4.0,
                                     seed=43, store_row_stride=1),
    "grad_bf16_alibi_mqa": _case("bf16", 2, 80, 112, 2, 1, 64, seed=44, alibi=True, store_row_stride=1),
    "grad_fp16_padded_causal_d128": _case("fp16", 2, 70, 90, 2, 2, 128, causal=True, padding="random", seed=45,
                                          store_row_stride=1),
}


def make_grad_output(c):
    """dO ~ N(0,1) in the case's dtype, (b, sq, h, d)."""
    g = torch.Generator().manual_seed(5000 + c["seed"])
    return torch.randn(c["b"], c["sq"], c["h"], c["d"], generator=g, dtype=torch.float32).to(_DT[c["dtype"]])


def make_inputs(c):
    """q, k, v in that order from torch.manual_seed(seed) (CPU generator: deterministic across hosts)."""
    g = torch.Generator().manual_seed(1000 + c["seed"])
    dt = _DT[c["dtype"]]
    q = torch.randn(c["b"], c["sq"], c["h"], c["d"], generator=g, dtype=torch.float32)
    k = torch.randn(c["b"], c["sk"], c["hk"], c["d"], generator=g, dtype=torch.float32)
    v = torch.randn(c["b"], c["sk"], c["hk"], c.get("dv", c["d"]), generator=g, dtype=torch.float32)
    q = q * c["q_scale"]
    q, k, v = q.to(dt), k.to(dt), v.to(dt)
    if c.get("fp8"):  # values an e4m3 tensor can hold, kept in bf16 for the oracle (the kernel gets .to(float8_e4m3fn))
        q, k, v = (t.to(torch.float8_e4m3fn).to(dt) for t in (q, k, v))
    return q, k, v


def make_descales(c):
    """(q_descale, k_descale, v_descale), each (b, hk) fp32 = rand * 2, or (None, None, None)."""
    if not c.get("fp8"):
        return None, None, None
    g = torch.Generator().manual_seed(3000 + c["seed"])
    return tuple(torch.rand(c["b"], c["hk"], generator=g, dtype=torch.float32) * 2 for _ in range(3))


def make_alibi_slopes(c):
    """(b, h) fp32 slopes = rand * 0.3, or None."""
    if not c.get("alibi"):
        return None
    g = torch.Generator().manual_seed(4000 + c["seed"])
    return torch.rand(c["b"], c["h"], generator=g, dtype=torch.float32) * 0.3


def padding_masks(c):
    """(query_padding_mask, key_padding_mask) bool (b, s) or (None, None).  Lengths in [max-20, max]."""
    if c["padding"] == "none":
        return None, None
    g = torch.Generator().manual_seed(2000 + c["seed"])

    def one(smax):
        lens = torch.randint(max(1, smax - 20), smax + 1, (c["b"], 1), generator=g)
        return torch.arange(smax).view(1, -1) < lens
    return one(c["sq"]), one(c["sk"])


def checksum(t):
    t = t.double()
    w = torch.arange(1, t.numel() + 1, dtype=torch.float64).reshape(t.shape) % 7 + 1
    return float((t * w).sum().item())
