"""FA3-flavoured public API — names, argument order and defaults of `hopper/flash_attn_interface.py`
(`flash_attn_func` :507-585, `flash_attn_varlen_func` :588-633) on top of `torch.ops.flash_attn_3.fwd`
(flash_attn_3_ops.py registers the library with the reference's schema; :10-14 of the reference does the same lookup).
This is the surface that carries fp8 e4m3 inputs with per-(batch, kv head) descales (BASELINE config 5)."""
import torch

from . import flash_attn_3_ops  # noqa: F401  (defines torch.ops.flash_attn_3)

flash_attn_3_gpu = torch.ops.flash_attn_3


def maybe_contiguous(x):
    return x.contiguous() if x is not None and x.stride(-1) != 1 else x


def _flash_attn_forward(q, k, v, k_new, v_new, qv, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new, seqused_q, seqused_k,
                        max_seqlen_q, max_seqlen_k, page_table, kv_batch_idx, leftpad_k, rotary_cos, rotary_sin,
                        seqlens_rotary, q_descale, k_descale, v_descale, softmax_scale, causal, window_size=(-1, -1),
                        attention_chunk=0, softcap=0.0, rotary_interleaved=True, scheduler_metadata=None, num_splits=1,
                        pack_gqa=None, sm_margin=0):
    """hopper/flash_attn_interface.py:20-102"""
    q, k = [maybe_contiguous(x) for x in (q, k)]
    v = v.contiguous() if v.stride(-1) != 1 and v.stride(-3) != 1 else v
    out, softmax_lse, *rest = flash_attn_3_gpu.fwd(
        q, k, v, k_new, v_new, qv, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new, seqused_q, seqused_k,
        max_seqlen_q, max_seqlen_k, page_table, kv_batch_idx, leftpad_k, rotary_cos, rotary_sin, seqlens_rotary,
        q_descale, k_descale, v_descale, softmax_scale, causal, window_size[0], window_size[1], attention_chunk, softcap,
        rotary_interleaved, scheduler_metadata, num_splits, pack_gqa, sm_margin)
    return out, softmax_lse, *rest


def flash_attn_func(q, k, v, softmax_scale=None, causal=False, qv=None, q_descale=None, k_descale=None, v_descale=None,
                    window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1, pack_gqa=None,
                    deterministic=False, sm_margin=0, return_attn_probs=False):
    """q: (batch, seqlen, nheads, headdim); k, v: (batch, seqlen_k, nheads_k, headdim); fp16 / bf16 / fp8 e4m3.
    Returns out (bf16 for fp8 inputs), or (out, softmax_lse (batch, nheads, seqlen)) when return_attn_probs."""
    if softmax_scale is None:
        softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
    out, softmax_lse, *_ = _flash_attn_forward(
        q, k, v, None, None, qv, None, None, None, None, None, None, None, None, None, None, None, None, None, None,
        q_descale, k_descale, v_descale, softmax_scale, causal=causal, window_size=window_size,
        attention_chunk=attention_chunk, softcap=softcap, num_splits=num_splits, pack_gqa=pack_gqa, sm_margin=sm_margin)
    return (out, softmax_lse) if return_attn_probs else out


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, seqused_q=None,
                           seqused_k=None, softmax_scale=None, causal=False, qv=None, q_descale=None, k_descale=None,
                           v_descale=None, window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1,
                           pack_gqa=None, deterministic=False, sm_margin=0, return_attn_probs=False):
    """q: (total_q, nheads, headdim); k, v: (total_k, nheads_k, headdim); cu_seqlens_*: (batch+1,) int32;
    seqused_*: (batch,) int32, the part of each sequence that is actually used."""
    if softmax_scale is None:
        softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
    out, softmax_lse, *_ = _flash_attn_forward(
        q, k, v, None, None, qv, None, cu_seqlens_q, cu_seqlens_k, None, seqused_q, seqused_k, max_seqlen_q, max_seqlen_k,
        None, None, None, None, None, None, q_descale, k_descale, v_descale, softmax_scale, causal=causal,
        window_size=window_size, attention_chunk=attention_chunk, softcap=softcap, num_splits=num_splits,
        pack_gqa=pack_gqa, sm_margin=sm_margin)
    return (out, softmax_lse) if return_attn_probs else out


def flash_attn_combine(out_partial, lse_partial, out=None, out_dtype=None):
    """reference hopper/flash_attn_interface.py:636-637"""
    return torch.ops.flash_attn_3.fwd_combine(out_partial, lse_partial, out, out_dtype)


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, qv=None, rotary_cos=None, rotary_sin=None,
                            cache_seqlens=None, cache_batch_idx=None, cache_leftpad=None, page_table=None,
                            cu_seqlens_q=None, cu_seqlens_k_new=None, max_seqlen_q=None, rotary_seqlens=None,
                            q_descale=None, k_descale=None, v_descale=None, softmax_scale=None, causal=False,
                            window_size=(-1, -1), attention_chunk=0, softcap=0.0, rotary_interleaved=True,
                            scheduler_metadata=None, num_splits=0, pack_gqa=None, sm_margin=0, return_softmax_lse=False):
    """reference hopper/flash_attn_interface.py:640-800: attention over a KV cache, optionally appending k / v in place
    (rotated by rotary_cos / rotary_sin) first.  Paged caches: any page size."""
    assert k_cache.stride(-1) == 1, "k_cache must have contiguous last dimension"
    assert v_cache.stride(-1) == 1, "v_cache must have contiguous last dimension"
    if softmax_scale is None:
        softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
    if cache_seqlens is not None and isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((q.shape[0],), cache_seqlens, dtype=torch.int32, device=k_cache.device)
    out, softmax_lse, *rest = _flash_attn_forward(
        q, k_cache, v_cache, k, v, qv, None, cu_seqlens_q, None, cu_seqlens_k_new, None, cache_seqlens, max_seqlen_q, None,
        page_table, cache_batch_idx, cache_leftpad, rotary_cos, rotary_sin, rotary_seqlens, q_descale, k_descale, v_descale,
        softmax_scale, causal=causal, window_size=window_size, attention_chunk=attention_chunk, softcap=softcap,
        rotary_interleaved=rotary_interleaved, scheduler_metadata=scheduler_metadata, num_splits=num_splits,
        pack_gqa=pack_gqa, sm_margin=sm_margin)
    return (out, softmax_lse, *rest) if return_softmax_lse else out
