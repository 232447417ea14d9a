"""Extension-module surface the reference's Python API binds to: `flash_attn_2_cuda`.

`flash_attn/flash_attn_interface.py:15` does `import flash_attn_2_cuda as flash_attn_gpu`
and calls `.fwd` (:91), `.varlen_fwd` (:168), `.bwd` (:269), `.varlen_bwd` (:369) and
`.fwd_kvcache` (:1594).  This module exports the same five names with the same positional
argument lists as the pybind module of `csrc/flash_attn/flash_api.cpp:1478-1485`; `fwd`
and `varlen_fwd` do the host work of `mha_fwd` (:350-512) / `mha_varlen_fwd` (:514-755) —
checks, output allocation, params — and enqueue the gfx950 kernel through the C-ABI
(`include/fa_fwd.h`) on torch's current stream.  Error texts are the reference's
`TORCH_CHECK` messages, raised as RuntimeError like c10::Error is.

Only the forward hot path is built: `bwd`, `varlen_bwd`, `fwd_kvcache` raise.
"""
import math
from typing import List, Optional

import torch

from . import _dispatch, _lib
from ._dispatch import aligned as _aligned

__all__ = ["fwd", "varlen_fwd", "bwd", "varlen_bwd", "fwd_kvcache"]


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _dtype_code(t):
    if t.dtype == torch.float16:
        return _lib.FA_DTYPE_FP16
    if t.dtype == torch.bfloat16:
        return _lib.FA_DTYPE_BF16
    raise RuntimeError("FlashAttention only support fp16 and bf16 data type")


def _check_device(x, name):
    _check(x.is_cuda, f"{name} must be on CUDA")


def _check_shape(x, name, *shape):
    _check(tuple(x.shape) == tuple(shape), f"{name} must have shape ({', '.join(str(s) for s in shape)})")


def _reject_unbuilt(alibi_slopes_, p_dropout, return_softmax):
    # accepted positionally like the reference; rejected by message like the reference does for
    # compiled-out features (hopper/flash_api.cpp:1148-1165)
    _check(alibi_slopes_ is None, "This flash attention build does not support alibi.")
    _check(p_dropout == 0.0, "This flash attention build does not support dropout.")
    _check(not return_softmax, "return_softmax is only supported when p_dropout > 0.0")


def fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out_: Optional[torch.Tensor],
        alibi_slopes_: Optional[torch.Tensor], p_dropout: float, softmax_scale: float, is_causal: bool,
        window_size_left: int, window_size_right: int, softcap: float, return_softmax: bool,
        gen_: Optional[torch.Generator]) -> List[torch.Tensor]:
    """mha_fwd, csrc/flash_attn/flash_api.cpp:350-512.  Returns [out, softmax_lse, p, rng_state]."""
    _lib.load()  # fail loudly before anything else if the HIP library is missing
    q_dtype = q.dtype
    _check(q_dtype in (torch.float16, torch.bfloat16), "FlashAttention only support fp16 and bf16 data type")
    _check(k.dtype == q_dtype, "query and key must have the same dtype")
    _check(v.dtype == q_dtype, "query and value must have the same dtype")
    _check_device(q, "q"); _check_device(k, "k"); _check_device(v, "v")
    _check(q.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(k.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(v.stride(-1) == 1, "Input tensor must have contiguous last dimension")

    batch_size, seqlen_q, num_heads, head_size = q.shape
    seqlen_k, num_heads_k = k.shape[1], k.shape[2]
    _check(batch_size > 0, "batch size must be positive")
    _check(head_size <= 256, "FlashAttention forward only supports head dimension at most 256")
    _check(head_size % 8 == 0, "query, key, value, and out_ must have a head_size that is a multiple of 8")
    _check(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query")
    if softcap > 0.0:
        _check(p_dropout == 0.0, "Softcapping does not support dropout for now")
    _reject_unbuilt(alibi_slopes_, p_dropout, return_softmax)

    # causal=true is the same as causal=false in this case (:402)
    if seqlen_q == 1 and alibi_slopes_ is None:
        is_causal = False

    _check_shape(q, "q", batch_size, seqlen_q, num_heads, head_size)
    _check_shape(k, "k", batch_size, seqlen_k, num_heads_k, head_size)
    _check_shape(v, "v", batch_size, seqlen_k, num_heads_k, head_size)

    if out_ is not None:
        out = out_
        _check(out.dtype == q_dtype, "Output must have the same dtype as inputs")
        _check_device(out, "out")
        _check(out.stride(-1) == 1, "Output tensor must have contiguous last dimension")
        _check_shape(out, "out", batch_size, seqlen_q, num_heads, head_size)
    else:
        out = torch.empty_like(q)

    with torch.cuda.device(q.device):
        softmax_lse = torch.empty((batch_size, num_heads, seqlen_q), dtype=torch.float32, device=q.device)
        p = torch.empty((0,), dtype=q_dtype, device=q.device)
        rng_state = torch.zeros((2,), dtype=torch.int64, device=q.device)

        if seqlen_k > 0 and seqlen_q > 0:
            qc, kc, vc = (x if _aligned(x) else x.contiguous() for x in (q, k, v))
            oc = out if _aligned(out) else torch.empty_like(qc)
            _dispatch.launch(qc, kc, vc, oc, softmax_lse, varlen=False, batch=batch_size, max_seqlen_q=seqlen_q,
                             max_seqlen_k=seqlen_k, softmax_scale=softmax_scale, causal=is_causal,
                             window_left=window_size_left, window_right=window_size_right, softcap=softcap)
            if oc is not out:
                out.copy_(oc)
        elif seqlen_q > 0:
            # If seqlen_k == 0, then we have an empty tensor. We need to set the output to 0. (:499-504)
            out.zero_()
            softmax_lse.fill_(math.inf)
    return [out, softmax_lse, p, rng_state]


def varlen_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out_: Optional[torch.Tensor],
               cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor, seqused_k: Optional[torch.Tensor],
               leftpad_k_: Optional[torch.Tensor], block_table_: Optional[torch.Tensor],
               alibi_slopes_: Optional[torch.Tensor], max_seqlen_q: int, max_seqlen_k: int, p_dropout: float,
               softmax_scale: float, zero_tensors: bool, is_causal: bool, window_size_left: int,
               window_size_right: int, softcap: float, return_softmax: bool,
               gen_: Optional[torch.Generator]) -> List[torch.Tensor]:
    """mha_varlen_fwd, csrc/flash_attn/flash_api.cpp:514-755.  Returns [out, softmax_lse (h,total_q), p, rng_state]."""
    _lib.load()
    q_dtype = q.dtype
    _check(q_dtype in (torch.float16, torch.bfloat16), "FlashAttention only support fp16 and bf16 data type")
    _check(k.dtype == q_dtype, "query and key must have the same dtype")
    _check(v.dtype == q_dtype, "query and value must have the same dtype")
    _check(cu_seqlens_q.dtype == torch.int32, "cu_seqlens_q must have dtype int32")
    _check(cu_seqlens_k.dtype == torch.int32, "cu_seqlens_k must have dtype int32")
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (cu_seqlens_q, "cu_seqlens_q"), (cu_seqlens_k, "cu_seqlens_k")):
        _check_device(t, n)
    _check(block_table_ is None, "This flash attention build does not support paged KV.")
    _check(leftpad_k_ is None, "This flash attention build does not support leftpad_k.")
    _check(q.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(k.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(v.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(cu_seqlens_q.is_contiguous(), "cu_seqlens_q must be contiguous")
    _check(cu_seqlens_k.is_contiguous(), "cu_seqlens_k must be contiguous")

    total_q, num_heads, head_size = q.shape
    batch_size = cu_seqlens_q.numel() - 1
    total_k, num_heads_k = k.shape[0], k.shape[1]
    _check(batch_size > 0, "batch size must be positive")
    _check(head_size <= 256, "FlashAttention forward only supports head dimension at most 256")
    _check(head_size % 8 == 0, "query, key, value, and out_ must have a head_size that is a multiple of 8")
    _check(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query")
    if softcap > 0.0:
        _check(p_dropout == 0.0, "Softcapping does not support dropout for now")
    _reject_unbuilt(alibi_slopes_, p_dropout, return_softmax)

    if max_seqlen_q == 1 and alibi_slopes_ is None:
        is_causal = False  # (:590)

    _check_shape(q, "q", total_q, num_heads, head_size)
    _check_shape(k, "k", total_k, num_heads_k, head_size)
    _check_shape(v, "v", total_k, num_heads_k, head_size)
    _check_shape(cu_seqlens_q, "cu_seqlens_q", batch_size + 1)
    _check_shape(cu_seqlens_k, "cu_seqlens_k", batch_size + 1)
    if seqused_k is not None:
        _check(seqused_k.dtype == torch.int32, "seqused_k must have dtype int32")
        _check_device(seqused_k, "seqused_k")
        _check(seqused_k.is_contiguous(), "seqused_k must be contiguous")
        _check_shape(seqused_k, "seqused_k", batch_size)

    if out_ is not None:
        out = out_
        _check(out.dtype == q_dtype, "Output must have the same dtype as inputs")
        _check_device(out, "out")
        _check(out.stride(-1) == 1, "Output tensor must have contiguous last dimension")
        _check_shape(out, "out", total_q, num_heads, head_size)
    else:
        out = torch.empty_like(q)

    with torch.cuda.device(q.device):
        softmax_lse = torch.empty((num_heads, total_q), dtype=torch.float32, device=q.device)
        p = torch.empty((0,), dtype=q_dtype, device=q.device)
        rng_state = torch.zeros((2,), dtype=torch.int64, device=q.device)
        if zero_tensors:
            out.zero_()
            softmax_lse.fill_(-math.inf)

        if max_seqlen_k > 0 and total_q > 0 and max_seqlen_q > 0:
            qc, kc, vc = (x if _aligned(x) else x.contiguous() for x in (q, k, v))
            oc = out if _aligned(out) else torch.empty_like(qc)
            _dispatch.launch(qc, kc, vc, oc, softmax_lse, varlen=True, batch=batch_size, max_seqlen_q=max_seqlen_q,
                             max_seqlen_k=max_seqlen_k, softmax_scale=softmax_scale, causal=is_causal,
                             window_left=window_size_left, window_right=window_size_right, softcap=softcap,
                             cu_seqlens_q=cu_seqlens_q, cu_seqlens_k=cu_seqlens_k, seqused_k=seqused_k)
            if oc is not out:
                out.copy_(oc)
        elif total_q > 0:
            out.zero_()
            softmax_lse.fill_(math.inf)
    return [out, softmax_lse, p, rng_state]


def bwd(*args, **kwargs):
    """mha_bwd, csrc/flash_attn/flash_api.cpp:767 — outside the forward hot path."""
    raise RuntimeError("flash_attn_2_cuda.bwd: the backward pass is not built in this forward-only back-end")


def varlen_bwd(*args, **kwargs):
    """mha_varlen_bwd, csrc/flash_attn/flash_api.cpp:973 — not built."""
    raise RuntimeError("flash_attn_2_cuda.varlen_bwd: the backward pass is not built in this forward-only back-end")


def fwd_kvcache(*args, **kwargs):
    """mha_fwd_kvcache, csrc/flash_attn/flash_api.cpp:1202 — decode path, not built."""
    raise RuntimeError("flash_attn_2_cuda.fwd_kvcache: the KV-cache decode path is not built in this back-end")
