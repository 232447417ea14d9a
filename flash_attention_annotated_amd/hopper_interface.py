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


def _flash_attn_backward(dout, q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k, max_seqlen_q,
                         max_seqlen_k, dq, dk, dv, softmax_scale, causal, window_size=(-1, -1), softcap=0.0,
                         deterministic=False, sm_margin=0):
    """hopper/flash_attn_interface.py:105-154"""
    dout, q, k, v, out = [maybe_contiguous(x) for x in (dout, q, k, v, out)]
    dq, dk, dv, softmax_d, *rest = torch.ops.flash_attn_3.bwd(
        dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k, max_seqlen_q,
        max_seqlen_k, softmax_scale, causal, window_size[0], window_size[1], softcap, deterministic, sm_margin)
    return dq, dk, dv, softmax_d


class FlashAttnFunc(torch.autograd.Function):
    """hopper/flash_attn_interface.py:254-339"""

    @staticmethod
    def forward(ctx, q, k, v, softmax_scale, causal, qv=None, q_descale=None, k_descale=None, v_descale=None,
                window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1, pack_gqa=None, deterministic=False,
                sm_margin=0, return_softmax=False):
        if softmax_scale is None:
            softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
        out, softmax_lse, *_ = _flash_attn_forward(
            q, k, v, None, None, qv, None, None, None, None, None, None, None, None, None, None, None, None, None, None,
            q_descale, k_descale, v_descale, softmax_scale, causal=causal, window_size=window_size,
            attention_chunk=attention_chunk, softcap=softcap, num_splits=num_splits, pack_gqa=pack_gqa, sm_margin=sm_margin)
        ctx.save_for_backward(q, k, v, out, softmax_lse)
        ctx.softmax_scale, ctx.causal, ctx.window_size = softmax_scale, causal, window_size
        ctx.attention_chunk, ctx.softcap, ctx.deterministic, ctx.sm_margin = attention_chunk, softcap, deterministic, sm_margin
        if return_softmax:
            ctx.mark_non_differentiable(softmax_lse)
        return (out, softmax_lse) if return_softmax else out

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v, out, softmax_lse = ctx.saved_tensors
        assert ctx.attention_chunk == 0, "FA3 backward does not support attention_chunk"
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _flash_attn_backward(dout, q, k, v, out, softmax_lse, None, None, None, None, None, None, dq, dk, dv,
                             ctx.softmax_scale, ctx.causal, ctx.window_size, ctx.softcap, ctx.deterministic, ctx.sm_margin)
        return (dq, dk, dv) + (None,) * 14


class FlashAttnVarlenFunc(torch.autograd.Function):
    """hopper/flash_attn_interface.py:342-442"""

    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k, max_seqlen_q, max_seqlen_k, softmax_scale,
                causal, qv=None, q_descale=None, k_descale=None, v_descale=None, window_size=(-1, -1), attention_chunk=0,
                softcap=0.0, num_splits=1, pack_gqa=None, deterministic=False, sm_margin=0, return_softmax=False):
        if softmax_scale is None:
            softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
        out, softmax_lse, *_ = _flash_attn_forward(
            q, k, v, None, None, qv, None, cu_seqlens_q, cu_seqlens_k, None, seqused_q, seqused_k, max_seqlen_q,
            max_seqlen_k, None, None, None, None, None, None, q_descale, k_descale, v_descale, softmax_scale, causal=causal,
            window_size=window_size, attention_chunk=attention_chunk, softcap=softcap, num_splits=num_splits,
            pack_gqa=pack_gqa, sm_margin=sm_margin)
        ctx.save_for_backward(q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k)
        ctx.max_seqlen_q, ctx.max_seqlen_k = max_seqlen_q, max_seqlen_k
        ctx.softmax_scale, ctx.causal, ctx.window_size = softmax_scale, causal, window_size
        ctx.attention_chunk, ctx.softcap, ctx.deterministic, ctx.sm_margin = attention_chunk, softcap, deterministic, sm_margin
        if return_softmax:
            ctx.mark_non_differentiable(softmax_lse)
        return (out, softmax_lse) if return_softmax else out

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k = ctx.saved_tensors
        assert ctx.attention_chunk == 0, "FA3 backward does not support attention_chunk"
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _flash_attn_backward(dout, q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k,
                             ctx.max_seqlen_q, ctx.max_seqlen_k, dq, dk, dv, ctx.softmax_scale, ctx.causal, ctx.window_size,
                             ctx.softcap, ctx.deterministic, ctx.sm_margin)
        return (dq, dk, dv) + (None,) * 20


class FlashAttnQKVPackedFunc(torch.autograd.Function):
    """hopper/flash_attn_interface.py:157-251: qkv (b, s, 3, h, d), or (b, s, h_q + 2 h_k, d) with num_heads_q."""

    @staticmethod
    def forward(ctx, qkv, softmax_scale, causal, q_descale=None, k_descale=None, v_descale=None, window_size=(-1, -1),
                attention_chunk=0, softcap=0.0, deterministic=False, num_heads_q=None, sm_margin=0, return_softmax=False):
        if softmax_scale is None:
            softmax_scale = qkv.shape[-1] ** (-0.5)
        if qkv.dim() == 5:
            assert qkv.shape[-3] == 3
            q, k, v = qkv.unbind(dim=-3)
        else:
            assert qkv.dim() == 4
            assert num_heads_q is not None
            num_heads_k = (qkv.shape[2] - num_heads_q) // 2
            assert num_heads_k * 2 + num_heads_q == qkv.shape[2]
            q, k, v = qkv.split([num_heads_q, num_heads_k, num_heads_k], dim=-2)
        out, softmax_lse, *_ = _flash_attn_forward(
            q, k, v, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None,
            q_descale, k_descale, v_descale, softmax_scale, causal=causal, window_size=window_size,
            attention_chunk=attention_chunk, softcap=softcap, sm_margin=sm_margin)
        ctx.save_for_backward(q, k, v, out, softmax_lse)
        ctx.softmax_scale, ctx.causal, ctx.window_size = softmax_scale, causal, window_size
        ctx.attention_chunk, ctx.softcap, ctx.deterministic, ctx.sm_margin = attention_chunk, softcap, deterministic, sm_margin
        ctx.ndim = qkv.dim()
        if return_softmax:
            ctx.mark_non_differentiable(softmax_lse)
        return (out, softmax_lse) if return_softmax else out

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v, out, softmax_lse = ctx.saved_tensors
        assert ctx.attention_chunk == 0, "FA3 backward does not support attention_chunk"
        if ctx.ndim == 5:
            dqkv = torch.empty(q.shape[:-2] + (3, *q.shape[-2:]), dtype=q.dtype, device=q.device)
            dq, dk, dv = dqkv.unbind(dim=-3)
        else:
            hq, hk = q.shape[2], k.shape[2]
            dqkv = torch.empty(q.shape[:-2] + (hq + 2 * hk, q.shape[-1]), dtype=q.dtype, device=q.device)
            dq, dk, dv = dqkv.split([hq, hk, hk], dim=-2)
        _flash_attn_backward(dout, q, k, v, out, softmax_lse, None, None, None, None, None, None, dq, dk, dv,
                             ctx.softmax_scale, ctx.causal, ctx.window_size, ctx.softcap, ctx.deterministic, ctx.sm_margin)
        return (dqkv,) + (None,) * 12


def flash_attn_qkvpacked_func(qkv, softmax_scale=None, causal=False, q_descale=None, k_descale=None, v_descale=None,
                              window_size=(-1, -1), attention_chunk=0, softcap=0.0, deterministic=False, num_heads_q=None,
                              sm_margin=0, return_attn_probs=False):
    """hopper/flash_attn_interface.py:445-504.  qkv: (batch, seqlen, 3, nheads, headdim) -- or (batch, seqlen,
    nheads_q + 2 nheads_k, headdim) with num_heads_q for GQA."""
    return FlashAttnQKVPackedFunc.apply(qkv, softmax_scale, causal, q_descale, k_descale, v_descale, window_size,
                                        attention_chunk, softcap, deterministic, num_heads_q, sm_margin, return_attn_probs)


def flash_attn_func(q, k, v, softmax_scale=None, causal=False, qv=None, q_descale=None, k_descale=None, v_descale=None,
                    window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1, pack_gqa=None,
                    deterministic=False, sm_margin=0, return_attn_probs=False):
    """hopper/flash_attn_interface.py:507-585.  q: (batch, seqlen, nheads, headdim); k, v: (batch, seqlen_k, nheads_k,
    headdim); fp16 / bf16 (differentiable) / fp8 e4m3 (forward only, bf16 output).
    Returns out, or (out, softmax_lse (batch, nheads, seqlen)) when return_attn_probs."""
    return FlashAttnFunc.apply(q, k, v, softmax_scale, causal, qv, q_descale, k_descale, v_descale, window_size,
                               attention_chunk, softcap, num_splits, pack_gqa, deterministic, sm_margin, return_attn_probs)


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, seqused_q=None,
                           seqused_k=None, softmax_scale=None, causal=False, qv=None, q_descale=None, k_descale=None,
                           v_descale=None, window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1,
                           pack_gqa=None, deterministic=False, sm_margin=0, return_attn_probs=False):
    """hopper/flash_attn_interface.py:588-633.  q: (total_q, nheads, headdim); k, v: (total_k, nheads_k, headdim);
    cu_seqlens_*: (batch+1,) int32; seqused_*: (batch,) int32, the part of each sequence that is actually used."""
    return FlashAttnVarlenFunc.apply(q, k, v, cu_seqlens_q, cu_seqlens_k, seqused_q, seqused_k, max_seqlen_q, max_seqlen_k,
                                     softmax_scale, causal, qv, q_descale, k_descale, v_descale, window_size,
                                     attention_chunk, softcap, num_splits, pack_gqa, deterministic, sm_margin,
                                     return_attn_probs)


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


def get_scheduler_metadata(batch_size, max_seqlen_q, max_seqlen_k, num_heads_q, num_heads_kv, headdim, cache_seqlens,
                           qkv_dtype=torch.bfloat16, headdim_v=None, cu_seqlens_q=None, cu_seqlens_k_new=None,
                           cache_leftpad=None, page_size=None, max_seqlen_k_new=0, causal=False, window_size=(-1, -1),
                           attention_chunk=0, has_softcap=False, num_splits=0, pack_gqa=None, sm_margin=0):
    """hopper/flash_attn_interface.py:803-845.  The tile schedule of this build is computed inside the kernel: the
    returned tensor is opaque, and `scheduler_metadata=` is accepted and ignored by the forward."""
    if headdim_v is None:
        headdim_v = headdim
    return torch.ops.flash_attn_3.get_scheduler_metadata(
        batch_size, max_seqlen_q, max_seqlen_k, num_heads_q, num_heads_kv, headdim, headdim_v, qkv_dtype,
        maybe_contiguous(cache_seqlens), cu_seqlens_q, None, cu_seqlens_k_new, None, cache_leftpad, page_size,
        max_seqlen_k_new, causal, window_size[0], window_size[1], attention_chunk, has_softcap, num_splits, pack_gqa,
        sm_margin)
