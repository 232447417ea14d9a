"""Public Python API of the attention op (forward hot path + its backward) — same names, arguments, defaults and return
conventions as the reference's `flash_attn/flash_attn_interface.py`:

    flash_attn_func                     reference :1145-1219
    flash_attn_varlen_func              reference :1380-1471
    flash_attn_qkvpacked_func           reference :1008-1062
    flash_attn_kvpacked_func            reference :1065-1142
    flash_attn_varlen_qkvpacked_func    reference :1222-1287
    flash_attn_varlen_kvpacked_func     reference :1290-1377

Host glue only (reference FlashAttnFunc.forward :817-867 / FlashAttnVarlenFunc.forward :903-968):
default scale d**-0.5, head dim padded to a multiple of 8 and un-padded on return,
`return_attn_probs` returning (out, softmax_lse, S_dmask).  The compute goes through the
`flash_attn_2_cuda`-shaped module (this package's flash_attn_2_cuda.py) to the HIP kernel.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import flash_attn_2_cuda as flash_attn_gpu


def maybe_contiguous(x):
    return x.contiguous() if x is not None and x.stride(-1) != 1 else x


def _precheck(q, k, v):
    """The first two checks of mha_fwd (csrc/flash_attn/flash_api.cpp:370-377), made before the dispatcher sees the
    tensors: the custom ops only exist for the GPU backend, and its own 'could not run ... CPU backend' error would
    hide the reference's messages."""
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("FlashAttention only support fp16 and bf16 data type")
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be on CUDA")


def _pad_head_dim(*tensors):
    d = tensors[0].shape[-1]
    if d % 8 == 0:
        return tensors
    pad = 8 - d % 8
    return tuple(F.pad(t, [0, pad]) for t in tensors)


def round_multiple(x, m):
    return (x + m - 1) // m * m


# torch.compile surface (reference :56-73, 76, 109, 139-142): the four host entry points are torch.library custom
# ops with fake (shape-only) implementations.  Namespace `flash_attn_amd` so that the reference package, which
# registers `flash_attn::*` itself, can live in the same process (INTEGRATION.md option A).
_custom_op = torch.library.custom_op
_register_fake = torch.library.register_fake


@_custom_op("flash_attn_amd::_flash_attn_forward", mutates_args=(), device_types="cuda")
def _flash_attn_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dropout_p: float, softmax_scale: float,
                        causal: bool, window_size_left: int, window_size_right: int, softcap: float,
                        alibi_slopes: Optional[torch.Tensor], return_softmax: bool
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """reference :76-106"""
    q, k, v = [maybe_contiguous(x) for x in (q, k, v)]
    out, softmax_lse, S_dmask, rng_state = flash_attn_gpu.fwd(
        q, k, v, None, alibi_slopes, dropout_p, softmax_scale, causal,
        window_size_left, window_size_right, softcap, return_softmax, None)
    return out, softmax_lse, S_dmask, rng_state


@_register_fake("flash_attn_amd::_flash_attn_forward")
def _flash_attn_forward_fake(q, k, v, dropout_p, softmax_scale, causal, window_size_left, window_size_right, softcap,
                             alibi_slopes, return_softmax):
    """reference :109-136 (the CUDA branch: `p` is (b, h, seqlen_q rounded to 128, seqlen_k rounded to 128), :132-135)"""
    batch_size, seqlen_q, num_heads, _ = q.shape
    seqlen_k = k.shape[1]
    out = torch.empty_like(q)  # (same strides as the real op: flash_attn_2_cuda.fwd allocates empty_like(q))
    softmax_lse = torch.empty((batch_size, num_heads, seqlen_q), dtype=torch.float32, device=q.device)
    p = torch.empty((0,), dtype=q.dtype, device=q.device)
    if return_softmax:
        p = torch.empty((batch_size, num_heads, (seqlen_q + 127) // 128 * 128, (seqlen_k + 127) // 128 * 128), dtype=q.dtype,
                        device=q.device)
    rng_state = torch.empty((2,), dtype=torch.int64, device=q.device)
    return out, softmax_lse, p, rng_state


@_custom_op("flash_attn_amd::_flash_attn_backward", mutates_args=("dq", "dk", "dv"), device_types="cuda")
def _flash_attn_backward(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor,
                         softmax_lse: torch.Tensor, dq: Optional[torch.Tensor], dk: Optional[torch.Tensor],
                         dv: Optional[torch.Tensor], dropout_p: float, softmax_scale: float, causal: bool,
                         window_size_left: int, window_size_right: int, softcap: float,
                         alibi_slopes: Optional[torch.Tensor], deterministic: bool,
                         rng_state: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference :241-289"""
    dout, q, k, v, out = [maybe_contiguous(x) for x in (dout, q, k, v, out)]
    dq, dk, dv, softmax_d = flash_attn_gpu.bwd(
        dout, q, k, v, out, softmax_lse, dq, dk, dv, alibi_slopes, dropout_p, softmax_scale, causal,
        window_size_left, window_size_right, softcap, deterministic, None, rng_state)
    return softmax_d


@_register_fake("flash_attn_amd::_flash_attn_backward")
def _flash_attn_backward_fake(dout, q, k, v, out, softmax_lse, dq, dk, dv, dropout_p, softmax_scale, causal,
                              window_size_left, window_size_right, softcap, alibi_slopes, deterministic,
                              rng_state=None):
    """reference :292-325"""
    batch_size, seqlen_q, num_heads, _ = q.shape
    return torch.empty((batch_size, num_heads, round_multiple(seqlen_q, 128)), device=q.device, dtype=torch.float32)


@_custom_op("flash_attn_amd::_flash_attn_varlen_backward", mutates_args=("dq", "dk", "dv"), device_types="cuda")
def _flash_attn_varlen_backward(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                                out: torch.Tensor, softmax_lse: torch.Tensor, dq: Optional[torch.Tensor],
                                dk: Optional[torch.Tensor], dv: Optional[torch.Tensor], cu_seqlens_q: torch.Tensor,
                                cu_seqlens_k: torch.Tensor, max_seqlen_q: int, max_seqlen_k: int, dropout_p: float,
                                softmax_scale: float, causal: bool, window_size_left: int, window_size_right: int,
                                softcap: float, alibi_slopes: Optional[torch.Tensor], deterministic: bool,
                                rng_state: Optional[torch.Tensor] = None, zero_tensors: bool = False) -> torch.Tensor:
    """reference :337-392"""
    dout, q, k, v, out = [maybe_contiguous(x) for x in (dout, q, k, v, out)]
    dq, dk, dv, softmax_d = flash_attn_gpu.varlen_bwd(
        dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, alibi_slopes, max_seqlen_q,
        max_seqlen_k, dropout_p, softmax_scale, zero_tensors, causal, window_size_left, window_size_right, softcap,
        deterministic, None, rng_state)
    return softmax_d


@_register_fake("flash_attn_amd::_flash_attn_varlen_backward")
def _flash_attn_varlen_backward_fake(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k,
                                     max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal, window_size_left,
                                     window_size_right, softcap, alibi_slopes, deterministic, rng_state=None,
                                     zero_tensors=False):
    """reference :395-430"""
    batch_size = cu_seqlens_q.numel() - 1
    total_q, num_heads, _ = q.shape
    return torch.empty((num_heads, total_q + 128 * batch_size), device=q.device, dtype=torch.float32)


@_custom_op("flash_attn_amd::_flash_attn_varlen_forward", mutates_args=(), device_types="cuda")
def _flash_attn_varlen_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor,
                               cu_seqlens_k: torch.Tensor, max_seqlen_q: int, max_seqlen_k: int, dropout_p: float,
                               softmax_scale: float, causal: bool, window_size_left: int = -1,
                               window_size_right: int = -1, softcap: float = 0.0,
                               alibi_slopes: Optional[torch.Tensor] = None, return_softmax: bool = False,
                               block_table: Optional[torch.Tensor] = None, leftpad_k: Optional[torch.Tensor] = None,
                               seqused_k: Optional[torch.Tensor] = None, zero_tensors: bool = False
                               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """reference :145-193"""
    q, k, v = [maybe_contiguous(x) for x in (q, k, v)]
    out, softmax_lse, S_dmask, rng_state = flash_attn_gpu.varlen_fwd(
        q, k, v, None, cu_seqlens_q, cu_seqlens_k, seqused_k, leftpad_k, block_table, alibi_slopes,
        max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, zero_tensors, causal,
        window_size_left, window_size_right, softcap, return_softmax, None)
    return out, softmax_lse, S_dmask, rng_state


@_register_fake("flash_attn_amd::_flash_attn_varlen_forward")
def _flash_attn_varlen_forward_fake(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                                    softmax_scale, causal, window_size_left=-1, window_size_right=-1, softcap=0.0,
                                    alibi_slopes=None, return_softmax=False, block_table=None, leftpad_k=None,
                                    seqused_k=None, zero_tensors=False):
    """reference :196-233"""
    batch_size = cu_seqlens_q.numel() - 1
    total_q, num_heads, _ = q.shape
    out = torch.empty_like(q)  # (same strides as the real op: flash_attn_2_cuda.fwd allocates empty_like(q))
    softmax_lse = torch.empty((num_heads, total_q), dtype=torch.float32, device=q.device)
    p = torch.empty((0,), dtype=q.dtype, device=q.device)
    if return_softmax:
        p = torch.empty((batch_size, num_heads, (max_seqlen_q + 127) // 128 * 128, (max_seqlen_k + 127) // 128 * 128),
                        dtype=q.dtype, device=q.device)
    rng_state = torch.empty((2,), dtype=torch.int64, device=q.device)
    return out, softmax_lse, p, rng_state


class FlashAttnFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                deterministic, return_softmax, is_grad_enabled):
        _precheck(q, k, v)
        is_grad = is_grad_enabled and any(x.requires_grad for x in [q, k, v])
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        head_size_og = q.size(-1)
        qp, kp, vp = _pad_head_dim(q, k, v)
        out_padded, softmax_lse, S_dmask, rng_state = _flash_attn_forward(
            qp, kp, vp, dropout_p, softmax_scale, causal=causal, window_size_left=window_size[0],
            window_size_right=window_size[1], softcap=softcap, alibi_slopes=alibi_slopes,
            return_softmax=return_softmax and dropout_p > 0)
        if is_grad:
            ctx.save_for_backward(qp, kp, vp, out_padded, softmax_lse, rng_state)
            ctx.dropout_p = dropout_p
            ctx.softmax_scale = softmax_scale
            ctx.causal = causal
            ctx.window_size = window_size
            ctx.softcap = softcap
            ctx.alibi_slopes = alibi_slopes
            ctx.deterministic = deterministic
        out = out_padded[..., :head_size_og]
        return out if not return_softmax else (out, softmax_lse, S_dmask)

    @staticmethod
    def backward(ctx, dout, *args):
        """reference :869-900"""
        q, k, v, out, softmax_lse, rng_state = ctx.saved_tensors
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        (dout_padded,) = _pad_head_dim(dout)
        _flash_attn_backward(dout_padded, q, k, v, out, softmax_lse, dq, dk, dv, ctx.dropout_p, ctx.softmax_scale,
                             ctx.causal, ctx.window_size[0], ctx.window_size[1], ctx.softcap, ctx.alibi_slopes,
                             ctx.deterministic, rng_state=rng_state)
        d = dout.shape[-1]  # the head dimension may have been padded
        return dq[..., :d], dk[..., :d], dv[..., :d], None, None, None, None, None, None, None, None, None


class FlashAttnVarlenFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale,
                causal, window_size, softcap, alibi_slopes, deterministic, return_softmax, block_table,
                is_grad_enabled):
        _precheck(q, k, v)
        is_grad = is_grad_enabled and any(x.requires_grad for x in [q, k, v])
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        head_size_og = q.size(-1)
        qp, kp, vp = _pad_head_dim(q, k, v)
        out_padded, softmax_lse, S_dmask, rng_state = _flash_attn_varlen_forward(
            qp, kp, vp, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale,
            causal=causal, window_size_left=window_size[0], window_size_right=window_size[1], softcap=softcap,
            alibi_slopes=alibi_slopes, return_softmax=return_softmax and dropout_p > 0, block_table=block_table)
        if is_grad:
            ctx.save_for_backward(qp, kp, vp, out_padded, softmax_lse, cu_seqlens_q, cu_seqlens_k, rng_state)
            ctx.dropout_p = dropout_p
            ctx.max_seqlen_q = max_seqlen_q
            ctx.max_seqlen_k = max_seqlen_k
            ctx.softmax_scale = softmax_scale
            ctx.causal = causal
            ctx.window_size = window_size
            ctx.softcap = softcap
            ctx.alibi_slopes = alibi_slopes
            ctx.deterministic = deterministic
        out = out_padded[..., :head_size_og]
        return out if not return_softmax else (out, softmax_lse, S_dmask)

    @staticmethod
    def backward(ctx, dout, *args):
        """reference :970-1005"""
        q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, rng_state = ctx.saved_tensors
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        (dout_padded,) = _pad_head_dim(dout)
        _flash_attn_varlen_backward(dout_padded, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k,
                                    ctx.max_seqlen_q, ctx.max_seqlen_k, ctx.dropout_p, ctx.softmax_scale, ctx.causal,
                                    ctx.window_size[0], ctx.window_size[1], ctx.softcap, ctx.alibi_slopes,
                                    ctx.deterministic, rng_state=rng_state)
        d = dout.shape[-1]
        return (dq[..., :d], dk[..., :d], dv[..., :d], None, None, None, None, None, None, None, None, None, None,
                None, None, None, None)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                    softcap=0.0, alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q: (batch, seqlen_q, nheads, headdim); k, v: (batch, seqlen_k, nheads_k, headdim).

    MQA/GQA: nheads % nheads_k == 0, query head i reads kv head i // (nheads / nheads_k).
    causal masks are aligned to the BOTTOM-RIGHT corner when seqlen_q != seqlen_k; a query row with
    no visible key produces a zero output row.  window_size=(left, right): query i sees keys in
    [i + seqlen_k - seqlen_q - left, i + seqlen_k - seqlen_q + right].
    Returns out (batch, seqlen_q, nheads, headdim), or (out, softmax_lse (batch, nheads, seqlen_q),
    S_dmask) when return_attn_probs.
    """
    return FlashAttnFunc.apply(q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                               deterministic, return_attn_probs, torch.is_grad_enabled())


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                           alibi_slopes=None, deterministic=False, return_attn_probs=False, block_table=None):
    """q: (total_q, nheads, headdim); k, v: (total_k, nheads_k, headdim); cu_seqlens_*: (batch+1,) int32.

    Returns out (total_q, nheads, headdim), or (out, softmax_lse (nheads, total_q), S_dmask).
    """
    return FlashAttnVarlenFunc.apply(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                                     softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                                     return_attn_probs, block_table, torch.is_grad_enabled())


def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                              softcap=0.0, alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """qkv: (batch, seqlen, 3, nheads, headdim).  The three views are read in place (row stride 3*h*d)."""
    return flash_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p, softmax_scale, causal,
                           window_size, softcap, alibi_slopes, deterministic, return_attn_probs)


def flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                             softcap=0.0, alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q: (batch, seqlen_q, nheads, headdim); kv: (batch, seqlen_k, 2, nheads_k, headdim)."""
    return flash_attn_func(q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale, causal, window_size, softcap,
                           alibi_slopes, deterministic, return_attn_probs)


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                                     return_attn_probs=False):
    """qkv: (total, 3, nheads, headdim)."""
    return flash_attn_varlen_func(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, cu_seqlens, max_seqlen, max_seqlen,
                                  dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                                  deterministic, return_attn_probs)


def flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                                    softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q: (total_q, nheads, headdim); kv: (total_k, 2, nheads_k, headdim)."""
    return flash_attn_varlen_func(q, kv[:, 0], kv[:, 1], cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                  dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                                  deterministic, return_attn_probs)


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None, cache_seqlens=None,
                            cache_batch_idx=None, cache_leftpad=None, block_table=None, softmax_scale=None,
                            causal=False, window_size=(-1, -1), softcap=0.0, rotary_interleaved=True,
                            alibi_slopes=None, num_splits=0, return_softmax_lse=False):
    """reference :1474-1616.  q: (batch, seqlen_q, nheads, headdim); k_cache, v_cache: (batch_cache, seqlen_cache,
    nheads_k, headdim).  If k / v are given they are written IN PLACE into the caches at rows
    [cache_seqlens, cache_seqlens + seqlen_new) and attention runs over the updated cache (incremental decoding).
    cache_seqlens: int or (batch,) int32; cache_batch_idx: (batch,) int32 indices into the cache.
    Causal / window masks are aligned to the bottom-right corner of each (seqlen_q, cache_seqlens + seqlen_new) block.
    block_table: (batch, max_blocks_per_seq) int32 for a paged cache (k_cache, v_cache: (num_blocks, page_block_size,
    nheads_k, headdim), page_block_size % 256 == 0).
    rotary_cos / rotary_sin: (seqlen_ro, rotary_dim / 2): rotary embedding of the appended keys (position
    cache_seqlens + i) and of q (the same positions when causal / local, otherwise all rows at cache_seqlens);
    rotary_interleaved: pairs (2j, 2j+1) instead of (j, j + rotary_dim/2).
    num_splits: 0 = heuristic (splits the key range when the tiles would leave most CUs idle), 1 = no split, N = N
    splits.  cache_leftpad: (batch,) int32, rows of padding in front of each cache entry's keys.
    Returns out (batch, seqlen_q, nheads, headdim) [, softmax_lse (batch, nheads, seqlen_q)]."""
    assert k_cache.stride(-1) == 1, "k_cache must have contiguous last dimension"
    assert v_cache.stride(-1) == 1, "v_cache must have contiguous last dimension"
    q, k, v = [maybe_contiguous(x) for x in (q, k, v)]
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    if cache_seqlens is not None and isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((q.shape[0],), cache_seqlens, dtype=torch.int32, device=k_cache.device)
        cache_seqlens = maybe_contiguous(cache_seqlens)
    cache_batch_idx = maybe_contiguous(cache_batch_idx)
    block_table = maybe_contiguous(block_table)
    out, softmax_lse = flash_attn_gpu.fwd_kvcache(
        q, k_cache, v_cache, k, v, cache_seqlens, rotary_cos, rotary_sin, cache_batch_idx, cache_leftpad, block_table,
        alibi_slopes, None, softmax_scale, causal, window_size[0], window_size[1], softcap, rotary_interleaved,
        num_splits)
    return (out, softmax_lse) if return_softmax_lse else out
