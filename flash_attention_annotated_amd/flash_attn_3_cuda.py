"""FA3 operator surface: `flash_attn_3::fwd` (hopper/flash_api.cpp:672-1198; schema :1672-1707), as the Python
callable `hopper/flash_attn_interface.py:66` invokes — 34 positional arguments, returns
(out, softmax_lse, out_accum, softmax_lse_accum).

Built: fp16 / bf16 / fp8 e4m3 inputs (fp8 -> bf16 output, :859), per-(batch, kv head) q/k/v descales (:1115-1146),
dense and varlen (`cu_seqlens_*`, `seqused_*`), causal / sliding window / softcap, GQA.
KV-cache arguments (dense q, 16-bit): k_new/v_new (in-place append at seqused_k), page_table (any page size),
kv_batch_idx, leftpad_k, rotary_cos/sin (+ interleaved), num_splits -- served by the same routines as the FA2
`fwd_kvcache` surface (flash_attn_2_cuda.fwd_kvcache).
Accepted and rejected by message, like the reference does for compiled-out features (:1148-1165): qv,
attention_chunk, cu_seqlens_k_new, seqlens_rotary, KV-cache arguments together with cu_seqlens_q or fp8.
`scheduler_metadata`, `pack_gqa`, `sm_margin` are performance hints and do not change results: ignored.
"""
import math
from typing import Optional

import torch

from . import _dispatch, _lib

_FP8 = getattr(torch, "float8_e4m3fn", None)


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def fwd(q, k, v, k_new, v_new, qv, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new, seqused_q, seqused_k,
        max_seqlen_q, max_seqlen_k, page_table, kv_batch_idx, leftpad_k, rotary_cos, rotary_sin, seqlens_rotary,
        q_descale, k_descale, v_descale, softmax_scale, is_causal, window_size_left, window_size_right,
        attention_chunk, softcap, is_rotary_interleaved, scheduler_metadata, num_splits, pack_gqa, sm_margin):
    _lib.load()
    _check(q.dtype in (torch.float16, torch.bfloat16) or (_FP8 is not None and q.dtype == _FP8),
           "FlashAttention only supports fp16, bf16, and fp8_e4m3 data type")  # hopper/flash_api.cpp:714-722
    _check(k.dtype == q.dtype, "query and key must have the same dtype")
    _check(v.dtype == q.dtype, "query and value must have the same dtype")
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _check(t.is_cuda, f"{n} must be on CUDA")
        _check(t.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    for x, n in ((qv, "qv"), (cu_seqlens_k_new, "cu_seqlens_k_new")):
        _check(x is None, f"This flash attention build does not support {n}.")
    attention_chunk = int(attention_chunk or 0)
    _check(attention_chunk >= 0, "attention_chunk must be non-negative")
    is_fp8 = _FP8 is not None and q.dtype == _FP8
    head_size_v = v.shape[-1]  # hopper/flash_api.cpp:764
    if head_size_v != q.shape[-1]:  # :782-792 (the "Only Hopper" line is the one check that does not carry over)
        _check((128 < q.shape[-1] <= 192 and 96 < head_size_v <= 128) or (q.shape[-1] <= 64 and head_size_v <= 512),
               "If V headdim is different from Q/K dim, we only support Q/K headdim in (128, 192] and V headdim in (96, 128], "
               "or (Q/K <= 64 and V <= 512).")
        _check(not is_fp8, "This flash attention build does not support a V headdim of its own with fp8 inputs.")
        _check(head_size_v % 8 == 0, "head_size_v should be a multiple of 8")  # :856
    if seqlens_rotary is not None:  # hopper/flash_api.cpp:1074-1079; only read together with k_new + rotary (hopper/seqlen.h:89)
        _check(seqlens_rotary.is_cuda and seqlens_rotary.is_contiguous(), "seqlens_rotary must be a contiguous CUDA tensor")
        _check(seqlens_rotary.dtype == torch.int32, "seqlens_rotary must have dtype torch.int32")
        _check(tuple(seqlens_rotary.shape) == (q.shape[0],), "seqlens_rotary must have shape (batch_size,)")
        if k_new is None or rotary_cos is None:
            seqlens_rotary = None
    if any(x is not None for x in (k_new, v_new, page_table, kv_batch_idx, leftpad_k, rotary_cos, rotary_sin)):
        # KV-cache step (hopper/flash_api.cpp:736-760, 935-1060): k / v are the cache, seqused_k its fill levels
        _check(cu_seqlens_q is None and cu_seqlens_k is None and seqused_q is None,
               "This flash attention build does not support KV-cache arguments together with cu_seqlens / seqused_q.")
        _check(not is_fp8, "This flash attention build does not support KV-cache arguments with fp8 inputs.")
        _check(not attention_chunk and head_size_v == q.shape[-1],
               "This flash attention build does not support attention_chunk or a V headdim of its own with KV-cache arguments.")
        _check((k_new is None) == (v_new is None), "k_new and v_new must be passed together")
        _check((rotary_cos is None) == (rotary_sin is None), "rotary_cos and rotary_sin must be passed together")
        if k_new is not None or leftpad_k is not None:
            _check(seqused_k is not None, "seqused_k must be provided with k_new / leftpad_k")
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        from . import flash_attn_2_cuda
        o, lse = flash_attn_2_cuda._fwd_kvcache_impl(q, k, v, k_new, v_new, seqused_k, rotary_cos, rotary_sin, kv_batch_idx,
                                               leftpad_k, page_table, None, out, softmax_scale, bool(is_causal),
                                               int(window_size_left), int(window_size_right), float(softcap),
                                               bool(is_rotary_interleaved), int(num_splits), 1, seqlens_rotary)
        return o, lse, None, None
    if (cu_seqlens_q is None and cu_seqlens_k is None and seqused_q is None and seqused_k is not None and not is_fp8
            and q.dim() == 4 and q.shape[1] <= 128 and window_size_left < 0 and (window_size_right < 0 or is_causal)
            and out is None and not attention_chunk and head_size_v == q.shape[-1]):
        # plain decode over a cache (flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens=...)): the same routine as
        # the append / paged calls, which brings the split-KV heuristic (num_splits = 0) and the (b, 1, h) -> (b, ngroups, h_k)
        # GQA swap (hopper/flash_api.cpp:935-1060 runs them for every call with seqused_k)
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        from . import flash_attn_2_cuda
        o, lse = flash_attn_2_cuda._fwd_kvcache_impl(q, k, v, None, None, seqused_k, None, None, None, None, None, None, None,
                                               softmax_scale, bool(is_causal), -1, -1, float(softcap), False,
                                               int(num_splits), 1)
        return o, lse, None, None
    varlen_q = cu_seqlens_q is not None
    varlen_k = cu_seqlens_k is not None
    _check(varlen_q == varlen_k, "This flash attention build needs cu_seqlens_q and cu_seqlens_k together.")
    if varlen_q:
        _check(cu_seqlens_q.dtype == torch.int32 and cu_seqlens_k.dtype == torch.int32, "cu_seqlens must have dtype torch.int32")
        _check(cu_seqlens_q.is_contiguous() and cu_seqlens_k.is_contiguous(), "cu_seqlens must be contiguous")
        _check(max_seqlen_q is not None and max_seqlen_k is not None, "max_seqlen_q/k must be provided with cu_seqlens")
        total_q, num_heads, head_size = q.shape
        num_heads_k = k.shape[1]
        batch_size = cu_seqlens_q.numel() - 1
        seqlen_q, seqlen_k = int(max_seqlen_q), int(max_seqlen_k)
    else:
        batch_size, seqlen_q, num_heads, head_size = q.shape
        seqlen_k, num_heads_k = k.shape[1], k.shape[2]
        total_q = batch_size * seqlen_q
    _check(batch_size > 0, "batch size must be positive")
    # CHECK_SHAPE(k, ..., num_heads_k, head_size) / CHECK_SHAPE(v, ..., num_heads_k, head_size_v), hopper/flash_api.cpp:813-819
    _check(k.shape[-1] == head_size, f"k must have shape (..., {num_heads_k}, {head_size})")
    _check(tuple(v.shape[:-1]) == tuple(k.shape[:-1]), f"v must have shape {tuple(k.shape[:-1]) + (head_size_v,)}")
    _check(head_size <= 256, "FlashAttention forward only supports head dimension at most 256")
    _check(head_size % (16 if is_fp8 else 8) == 0,
           f"head_size should be a multiple of {16 if is_fp8 else 8}")  # :854-856
    _check(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query")
    for t, n in ((seqused_q, "seqused_q"), (seqused_k, "seqused_k")):
        if t is not None:
            _check(t.dtype == torch.int32 and t.is_contiguous() and t.numel() == batch_size, f"{n} must be int32 of shape (batch_size,)")
    for t, n in ((q_descale, "q_descale"), (k_descale, "k_descale"), (v_descale, "v_descale")):
        if t is not None:
            _check(is_fp8, f"{n} is only supported with fp8 inputs")
            _check(t.dtype == torch.float32 and tuple(t.shape) == (batch_size, num_heads_k), f"{n} must be fp32 (batch_size, num_heads_k)")
    if softmax_scale is None:
        softmax_scale = head_size ** (-0.5)
    # causal/local normalisation, hopper/flash_api.cpp:796-805
    if window_size_left >= seqlen_k - 1:
        window_size_left = -1
    if window_size_right >= seqlen_q - 1:
        window_size_right = -1
    if seqlen_q == 1 and window_size_left == -1 and window_size_right == -1 and attention_chunk == 0:
        is_causal = False  # causal=true is the same as causal=false in this case
    if is_causal:
        window_size_right = 0
    out_dtype = torch.bfloat16 if is_fp8 else q.dtype  # :859
    if out is not None:
        _check(out.dtype == out_dtype, "For FP8 input, output must have dtype BF16" if is_fp8 else "Output must have the same dtype as inputs")
        _check(out.is_cuda and out.stride(-1) == 1 and tuple(out.shape) == tuple(q.shape[:-1]) + (head_size_v,),
               "out must have shape (..., num_heads, head_size_v)")  # :866-870
    else:
        out = torch.empty(tuple(q.shape[:-1]) + (head_size_v,), dtype=out_dtype, device=q.device)  # :872-874
    with torch.cuda.device(q.device):
        lse_shape = (num_heads, total_q) if varlen_q else (batch_size, num_heads, seqlen_q)
        softmax_lse = torch.empty(lse_shape, dtype=torch.float32, device=q.device)
        if seqlen_k > 0 and total_q > 0 and seqlen_q > 0:
            qc, kc, vc = (x if _dispatch.aligned(x) else x.contiguous() for x in (q, k, v))
            oc = out if _dispatch.aligned(out) else torch.empty_like(out)
            _dispatch.launch(qc, kc, vc, oc, softmax_lse, varlen=varlen_q, batch=batch_size, max_seqlen_q=seqlen_q,
                             max_seqlen_k=seqlen_k, softmax_scale=softmax_scale, causal=is_causal,
                             window_left=window_size_left, window_right=window_size_right, softcap=softcap,
                             cu_seqlens_q=cu_seqlens_q, cu_seqlens_k=cu_seqlens_k, seqused_q=seqused_q,
                             seqused_k=seqused_k, q_descale=q_descale, k_descale=k_descale, v_descale=v_descale,
                             attention_chunk=attention_chunk,
                             fa3_window=True)  # a missing window side is unbounded (hopper/flash_api.cpp:152-153)
            if oc is not out:
                out.copy_(oc)
        elif total_q > 0:
            out.zero_()                    # hopper/flash_api.cpp:1190-1194
            softmax_lse.fill_(math.inf)
    return out, softmax_lse, None, None


def bwd(*args, **kwargs):
    """flash_attn_3::bwd (hopper/flash_api.cpp:1259-1570): see flash_attn_3_ops._bwd."""
    from . import flash_attn_3_ops  # noqa: F401
    return torch.ops.flash_attn_3.bwd(*args, **kwargs)


def fwd_combine(out_partial, lse_partial, out=None, out_dtype=None):
    """flash_attn_3::fwd_combine (hopper/flash_api.cpp:1569-1670): see flash_attn_3_ops._fwd_combine."""
    from . import flash_attn_3_ops  # noqa: F401
    return torch.ops.flash_attn_3.fwd_combine(out_partial, lse_partial, out, out_dtype)


def get_scheduler_metadata(*args, **kwargs):
    """flash_attn_3::get_scheduler_metadata: opaque and empty in this build (tiles are scheduled inside the kernel)."""
    from . import flash_attn_3_ops  # noqa: F401
    return torch.ops.flash_attn_3.get_scheduler_metadata(*args, **kwargs)
