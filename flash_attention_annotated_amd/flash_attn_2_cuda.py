"""Extension-module surface the reference's Python API binds to: `flash_attn_2_cuda`.

`flash_attn/flash_attn_interface.py:15` does `import flash_attn_2_cuda as flash_attn_gpu`
and calls `.fwd` (:91), `.varlen_fwd` (:168), `.bwd` (:269), `.varlen_bwd` (:369) and
`.fwd_kvcache` (:1594).  This module exports the same five names with the same positional
argument lists as the pybind module of `csrc/flash_attn/flash_api.cpp:1478-1485`; `fwd`
and `varlen_fwd` do the host work of `mha_fwd` (:350-512) / `mha_varlen_fwd` (:514-755) —
checks, output allocation, params — and enqueue the gfx950 kernel through the C-ABI
(`include/fa_fwd.h`) on torch's current stream.  Error texts are the reference's
`TORCH_CHECK` messages, raised as RuntimeError like c10::Error is.

`bwd` / `varlen_bwd` do the same for mha_bwd (:767-971) / mha_varlen_bwd (:973-1200) through include/fa_bwd.h.
`fwd_kvcache` covers the decode path (mha_fwd_kvcache :1202-1476): in-place append, rotary, cache_batch_idx, paged
and left-padded caches, split-KV.

Two bindings of the same host logic (round 3): the COMPILED module `flash_attn_2_cuda_C` (csrc/torch_binding.cpp: the pybind
module of flash_api.cpp:1478-1485, host-only C++ against the torch headers, built by `_lib.build()` with plain g++) is what
the five names resolve to when it is built; the Python statements below (ctypes onto the same C-ABI) are the fallback and
what `FA_BINDING=python` selects.  Both enqueue the same kernels of libfa_fwd_gfx950.so.
"""
import contextlib
import math
import os
import threading
from typing import List, Optional

import torch

from . import _dispatch, _lib
from ._dispatch import aligned as _aligned

__all__ = ["fwd", "varlen_fwd", "bwd", "varlen_bwd", "fwd_kvcache"]


# The FA3 operator surface (flash_attn_3_ops._bwd) runs its backward through bwd / varlen_bwd below with ITS window rule (a
# missing side is unbounded, include/fa_fwd.h FA_FLAG_FA3_WINDOW); the reference signatures have no room for that switch.
_window_rule = threading.local()


@contextlib.contextmanager
def fa3_window_rule():
    prev = getattr(_window_rule, "fa3", False)
    _window_rule.fa3 = True
    try:
        yield
    finally:
        _window_rule.fa3 = prev   # (nested use keeps the outer rule)


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


def _sdmask_block_n(head_dim, is_dropout, is_causal):
    """kBlockN of the reference's forward for this head dim (flash_attn/flash_attn_interface.py:23-46, the row of a device
    that is neither sm8x nor sm90 -- what `_get_block_size_n` answers on this GPU): the key-block width behind the running
    maxima of the returned S_dmask, which tests/test_flash_attn.py:479-526 undoes with the same table."""
    if head_dim <= 32:
        return 128
    if head_dim <= 64:
        return 128 if not is_dropout else 64
    if head_dim <= 96:
        return 64
    if head_dim <= 128:
        return 64 if not is_dropout else 32
    return 64


def _round128(x):
    return (x + 127) // 128 * 128


def _check_dropout(p_dropout, return_softmax):
    """csrc/flash_attn/flash_api.cpp:131 (p_dropout < 1) and :428-431 (return_softmax needs dropout)."""
    _check(0.0 <= p_dropout < 1.0, "p_dropout must be in [0, 1)")
    if return_softmax:
        _check(p_dropout > 0.0, "return_softmax is only supported when p_dropout > 0.0")


def _dropout_state(p_dropout, gen_, device):
    """The (seed, offset) pair of this call (role of philox_cuda_state, csrc/flash_attn/flash_api.cpp:486-493): two
    int64 drawn ON THE DEVICE from gen_ / torch's default generator of `device` -- reproducible under
    torch.manual_seed, advances the generator, never syncs the host.  Zeros when dropout is off, like the reference."""
    if p_dropout <= 0.0:
        return torch.zeros((2,), dtype=torch.int64, device=device)
    return torch.randint(-(1 << 62), 1 << 62, (2,), dtype=torch.int64, device=device, generator=gen_)


def _check_alibi(alibi_slopes_, batch_size, num_heads):
    """set_params_alibi, csrc/flash_attn/flash_api.cpp:331-349."""
    if alibi_slopes_ is None:
        return None
    _check(alibi_slopes_.dtype == torch.float32, "ALiBi slopes must have dtype fp32")
    _check_device(alibi_slopes_, "alibi_slopes")
    _check(alibi_slopes_.stride(-1) == 1, "ALiBi slopes tensor must have contiguous last dimension")
    _check(tuple(alibi_slopes_.shape) in ((num_heads,), (batch_size, num_heads)),
           "alibi_slopes must have shape (num_heads) or (batch_size, num_heads)")
    return alibi_slopes_


def _check_leftpad(leftpad_k_, batch_size, paged):
    """csrc/flash_attn/flash_api.cpp:1395-1403 (and :683-691 in mha_varlen_fwd)"""
    if leftpad_k_ is None:
        return
    _check(not paged, "We don't support Paged KV and leftpad_k running at the same time yet")
    _check(leftpad_k_.dtype == torch.int32, "leftpad_k must have dtype int32")
    _check_device(leftpad_k_, "leftpad_k")
    _check(leftpad_k_.is_contiguous(), "leftpad_k must be contiguous")
    _check_shape(leftpad_k_, "leftpad_k", batch_size)


def _check_block_table(block_table_, kcache, batch_size, page_multiple=256):
    """Paged KV (csrc/flash_attn/flash_api.cpp:554-560, 1245-1266): returns (page_block_size, max_num_blocks_per_seq)."""
    _check_device(block_table_, "block_table")
    _check(block_table_.dtype == torch.int32, "block_table must have dtype torch.int32")
    _check(block_table_.stride(-1) == 1, "block_table must have contiguous last dimension")
    _check(kcache.dim() == 4, "paged k/v must have shape (num_blocks, page_block_size, num_heads_k, head_size)")
    page_block_size = kcache.shape[1]
    # the FA2 entry points keep the reference's rule (:1265); the FA3 surface ("page_block_size can be arbitrary",
    # hopper/flash_attn_interface.py:712) passes page_multiple=1 -- the kernel reads any page size
    _check(page_block_size % page_multiple == 0, f"Paged KV cache block size must be divisible by {page_multiple}")
    _check(block_table_.dim() == 2 and block_table_.shape[0] == batch_size,
           "block_table must have shape (batch_size, max_num_blocks_per_seq)")
    return page_block_size, block_table_.shape[1]


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
    _check_dropout(p_dropout, return_softmax)
    alibi = _check_alibi(alibi_slopes_, batch_size, num_heads)

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
        # return_softmax: S_dmask as the reference's CUDA forward returns it (csrc/flash_attn/flash_api.cpp:436-449):
        # (b, h, seqlen_q rounded to 128, seqlen_k rounded to 128) in the input dtype, the probabilities relative to the
        # running maximum of their key block, negative where dropout discards them (include/fa_fwd.h FA_FLAG_SDMASK_SIGNED)
        p = (torch.zeros((batch_size, num_heads, _round128(seqlen_q), _round128(seqlen_k)), dtype=q_dtype, device=q.device)
             if return_softmax else torch.empty((0,), dtype=q_dtype, device=q.device))
        rng_state = _dropout_state(p_dropout, gen_, q.device)

        if seqlen_k > 0 and seqlen_q > 0:
            qc, kc, vc = (x if _aligned(x) else x.contiguous() for x in (q, k, v))
            oc = out if _aligned(out) else torch.empty_like(qc)
            _dispatch.launch(qc, kc, vc, oc, softmax_lse, varlen=False, batch=batch_size, max_seqlen_q=seqlen_q,
                             max_seqlen_k=seqlen_k, softmax_scale=softmax_scale, causal=is_causal,
                             window_left=window_size_left, window_right=window_size_right, softcap=softcap,
                             alibi_slopes=alibi, p_dropout=p_dropout, rng_state=rng_state if p_dropout > 0 else None,
                             s_dmask=p if return_softmax else None,
                             s_dmask_block_n=_sdmask_block_n(head_size, p_dropout > 0, is_causal),
                             # the split heuristic runs whenever there is no dropout, as in mha_fwd (flash_api.cpp:453-456);
                             # it only splits problems whose tiles leave most CUs idle
                             num_splits=0 if p_dropout == 0 else 1)
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
    paged = block_table_ is not None
    _check(q.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(k.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(v.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(cu_seqlens_q.is_contiguous(), "cu_seqlens_q must be contiguous")
    _check(cu_seqlens_k.is_contiguous(), "cu_seqlens_k must be contiguous")

    total_q, num_heads, head_size = q.shape
    batch_size = cu_seqlens_q.numel() - 1
    if paged:  # k, v: (num_blocks, page_block_size, h_k, d), rows found through block_table (:554-560, :608-612)
        _check_block_table(block_table_, k, batch_size)
        total_k, num_heads_k = 0, k.shape[2]
    else:
        total_k, num_heads_k = k.shape[0], k.shape[1]
    _check(batch_size > 0, "batch size must be positive")
    _check(head_size <= 256, "FlashAttention forward only supports head dimension at most 256")
    _check(head_size % 8 == 0, "query, key, value, and out_ must have a head_size that is a multiple of 8")
    _check(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query")
    if softcap > 0.0:
        _check(p_dropout == 0.0, "Softcapping does not support dropout for now")
    _check_dropout(p_dropout, return_softmax)
    if p_dropout > 0.0:
        _check(not paged and leftpad_k_ is None, "dropout is not supported with a paged or left-padded KV cache")
    alibi = _check_alibi(alibi_slopes_, batch_size, num_heads)

    if max_seqlen_q == 1 and alibi_slopes_ is None:
        is_causal = False  # (:590)

    _check_shape(q, "q", total_q, num_heads, head_size)
    if paged:
        _check_shape(k, "k", k.shape[0], k.shape[1], num_heads_k, head_size)
        _check_shape(v, "v", k.shape[0], k.shape[1], num_heads_k, head_size)
    else:
        _check_shape(k, "k", total_k, num_heads_k, head_size)
        _check_shape(v, "v", total_k, num_heads_k, head_size)
    _check_shape(cu_seqlens_q, "cu_seqlens_q", batch_size + 1)
    _check_shape(cu_seqlens_k, "cu_seqlens_k", batch_size + 1)
    _check_leftpad(leftpad_k_, batch_size, paged)
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
        # return_softmax: S_dmask (b, h, max_seqlen_q rounded to 128, max_seqlen_k rounded to 128), each sequence's block
        # at [i, :, :seqlen_q_i, :seqlen_k_i] (csrc/flash_attn/flash_api.cpp:648-660)
        p = (torch.zeros((batch_size, num_heads, _round128(max_seqlen_q), _round128(max_seqlen_k)), dtype=q_dtype,
                         device=q.device)
             if return_softmax else torch.empty((0,), dtype=q_dtype, device=q.device))
        rng_state = _dropout_state(p_dropout, gen_, q.device)
        if zero_tensors:
            out.zero_()
            softmax_lse.fill_(-math.inf)

        if max_seqlen_k > 0 and total_q > 0 and max_seqlen_q > 0:
            qc, kc, vc = (x if _aligned(x) else x.contiguous() for x in (q, k, v))
            oc = out if _aligned(out) else torch.empty_like(qc)
            _dispatch.launch(qc, kc, vc, oc, softmax_lse, varlen=True, batch=batch_size, max_seqlen_q=max_seqlen_q,
                             max_seqlen_k=max_seqlen_k, softmax_scale=softmax_scale, causal=is_causal,
                             window_left=window_size_left, window_right=window_size_right, softcap=softcap,
                             cu_seqlens_q=cu_seqlens_q, cu_seqlens_k=cu_seqlens_k, seqused_k=seqused_k,
                             alibi_slopes=alibi, block_table=block_table_, leftpad_k=leftpad_k_, p_dropout=p_dropout,
                             rng_state=rng_state if p_dropout > 0 else None, s_dmask=p if return_softmax else None,
                             s_dmask_block_n=_sdmask_block_n(head_size, p_dropout > 0, is_causal))
            if oc is not out:
                out.copy_(oc)
        elif total_q > 0:
            out.zero_()
            softmax_lse.fill_(math.inf)
    return [out, softmax_lse, p, rng_state]


def _grad_out(given, like, name, shape):
    if given is None:
        return torch.empty_like(like)
    _check(given.dtype == like.dtype, f"{name} must have the same dtype as q")
    _check_device(given, name)
    _check(given.stride(-1) == 1, f"{name} must have contiguous last dimension")
    _check_shape(given, name, *shape)
    return given


def _round_multiple(x, m):
    return (x + m - 1) // m * m


def bwd(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor,
        softmax_lse: torch.Tensor, dq_: Optional[torch.Tensor], dk_: Optional[torch.Tensor],
        dv_: Optional[torch.Tensor], alibi_slopes_: Optional[torch.Tensor], p_dropout: float, softmax_scale: float,
        is_causal: bool, window_size_left: int, window_size_right: int, softcap: float, deterministic: bool,
        gen_: Optional[torch.Generator], rng_state: Optional[torch.Tensor]) -> List[torch.Tensor]:
    """mha_bwd, csrc/flash_attn/flash_api.cpp:767-971.  Returns [dq, dk, dv, softmax_d]."""
    _lib.load()
    q_dtype = q.dtype
    _check(q_dtype in (torch.float16, torch.bfloat16), "FlashAttention only support fp16 and bf16 data type")
    _check(k.dtype == q_dtype, "query and key must have the same dtype")
    _check(v.dtype == q_dtype, "query and value must have the same dtype")
    _check(out.dtype == q_dtype, "query and out must have the same dtype")
    _check(dout.dtype == q_dtype, "query and dout must have the same dtype")
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout"), (softmax_lse, "softmax_lse")):
        _check_device(t, n)
    for t in (q, k, v):
        _check(t.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(out.stride(-1) == 1, "out tensor must have contiguous last dimension")
    _check(dout.stride(-1) == 1, "dout tensor must have contiguous last dimension")

    batch_size, seqlen_q, num_heads, head_size = q.shape
    seqlen_k, num_heads_k = k.shape[1], k.shape[2]
    _check(batch_size > 0, "batch size must be positive")
    _check(head_size % 8 == 0, "head_size should be a multiple of 8")
    _check(head_size <= 256, "FlashAttention backward only supports head dimension at most 256")
    _check(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query")
    if softcap > 0.0:
        _check(p_dropout == 0.0, "Softcapping does not support dropout for now")
    _check(0.0 <= p_dropout < 1.0, "p_dropout must be in [0, 1)")
    if p_dropout > 0.0:  # the forward's (seed, offset); without it a fresh pair is drawn like the reference (:895-910)
        if rng_state is None:
            rng_state = _dropout_state(p_dropout, gen_, q.device)
        _check(rng_state.dtype == torch.int64 and rng_state.numel() == 2 and rng_state.is_cuda and rng_state.is_contiguous(),
               "rng_state must be a contiguous int64 CUDA tensor with 2 elements")
    alibi = _check_alibi(alibi_slopes_, batch_size, num_heads)

    _check_shape(q, "q", batch_size, seqlen_q, num_heads, head_size)
    _check_shape(k, "k", batch_size, seqlen_k, num_heads_k, head_size)
    _check_shape(v, "v", batch_size, seqlen_k, num_heads_k, head_size)
    _check_shape(out, "out", batch_size, seqlen_q, num_heads, head_size)
    _check_shape(dout, "dout", batch_size, seqlen_q, num_heads, head_size)
    dq = _grad_out(dq_, q, "dq", (batch_size, seqlen_q, num_heads, head_size))
    dk = _grad_out(dk_, k, "dk", (batch_size, seqlen_k, num_heads_k, head_size))
    dv = _grad_out(dv_, v, "dv", (batch_size, seqlen_k, num_heads_k, head_size))

    with torch.cuda.device(q.device):
        softmax_d = torch.empty((batch_size, num_heads, _round_multiple(seqlen_q, 128)), dtype=torch.float32,
                                device=q.device)
        if seqlen_q > 0 and seqlen_k > 0:
            ins = [x if _aligned(x) else x.contiguous() for x in (dout, q, k, v, out)]
            outs = [x if _aligned(x) else torch.empty_like(x, memory_format=torch.contiguous_format) for x in (dq, dk, dv)]
            lse = softmax_lse if softmax_lse.is_contiguous() else softmax_lse.contiguous()
            _dispatch.launch_bwd(*ins, lse, *outs, softmax_d, varlen=False, batch=batch_size, max_seqlen_q=seqlen_q,
                                 max_seqlen_k=seqlen_k, softmax_scale=softmax_scale, causal=is_causal,
                                 window_left=window_size_left, window_right=window_size_right, softcap=softcap,
                                 alibi_slopes=alibi, deterministic=deterministic, p_dropout=p_dropout,
                                 rng_state=rng_state if p_dropout > 0 else None, fa3_window=getattr(_window_rule, "fa3", False))
            for dst, src in zip((dq, dk, dv), outs):
                if dst is not src:
                    dst.copy_(src)
        else:
            # If seqlen_q == 0 (or there are no keys), the gradients are zero (:953-958)
            dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_()
    return [dq, dk, dv, softmax_d]


def varlen_bwd(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor,
               softmax_lse: torch.Tensor, dq_: Optional[torch.Tensor], dk_: Optional[torch.Tensor],
               dv_: Optional[torch.Tensor], cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
               alibi_slopes_: Optional[torch.Tensor], max_seqlen_q: int, max_seqlen_k: int, p_dropout: float,
               softmax_scale: float, zero_tensors: bool, is_causal: bool, window_size_left: int,
               window_size_right: int, softcap: float, deterministic: bool, gen_: Optional[torch.Generator],
               rng_state: Optional[torch.Tensor]) -> List[torch.Tensor]:
    """mha_varlen_bwd, csrc/flash_attn/flash_api.cpp:973-1200.  Returns [dq, dk, dv, softmax_d]."""
    _lib.load()
    q_dtype = q.dtype
    _check(q_dtype in (torch.float16, torch.bfloat16), "FlashAttention only support fp16 and bf16 data type")
    _check(k.dtype == q_dtype, "query and key must have the same dtype")
    _check(v.dtype == q_dtype, "query and value must have the same dtype")
    _check(out.dtype == q_dtype, "query and out must have the same dtype")
    _check(dout.dtype == q_dtype, "query and dout must have the same dtype")
    _check(cu_seqlens_q.dtype == torch.int32, "cu_seqlens_q must have dtype int32")
    _check(cu_seqlens_k.dtype == torch.int32, "cu_seqlens_k must have dtype int32")
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout"), (softmax_lse, "softmax_lse"),
                 (cu_seqlens_q, "cu_seqlens_q"), (cu_seqlens_k, "cu_seqlens_k")):
        _check_device(t, n)
    for t in (q, k, v):
        _check(t.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    _check(out.stride(-1) == 1, "out tensor must have contiguous last dimension")
    _check(dout.stride(-1) == 1, "dout tensor must have contiguous last dimension")
    _check(cu_seqlens_q.is_contiguous(), "cu_seqlens_q must be contiguous")
    _check(cu_seqlens_k.is_contiguous(), "cu_seqlens_k must be contiguous")

    total_q, num_heads, head_size = q.shape
    batch_size = cu_seqlens_q.numel() - 1
    total_k, num_heads_k = k.shape[0], k.shape[1]
    _check(batch_size > 0, "batch size must be positive")
    _check(head_size % 8 == 0, "head_size should be a multiple of 8")
    _check(head_size <= 256, "FlashAttention backward only supports head dimension at most 256")
    _check(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query")
    if softcap > 0.0:
        _check(p_dropout == 0.0, "Softcapping does not support dropout for now")
    _check(0.0 <= p_dropout < 1.0, "p_dropout must be in [0, 1)")
    if p_dropout > 0.0:  # the forward's (seed, offset); without it a fresh pair is drawn like the reference (:895-910)
        if rng_state is None:
            rng_state = _dropout_state(p_dropout, gen_, q.device)
        _check(rng_state.dtype == torch.int64 and rng_state.numel() == 2 and rng_state.is_cuda and rng_state.is_contiguous(),
               "rng_state must be a contiguous int64 CUDA tensor with 2 elements")
    alibi = _check_alibi(alibi_slopes_, batch_size, num_heads)

    _check_shape(q, "q", total_q, num_heads, head_size)
    _check_shape(k, "k", total_k, num_heads_k, head_size)
    _check_shape(v, "v", total_k, num_heads_k, head_size)
    _check_shape(out, "out", total_q, num_heads, head_size)
    _check_shape(dout, "dout", total_q, num_heads, head_size)
    _check_shape(cu_seqlens_q, "cu_seqlens_q", batch_size + 1)
    _check_shape(cu_seqlens_k, "cu_seqlens_k", batch_size + 1)
    dq = _grad_out(dq_, q, "dq", (total_q, num_heads, head_size))
    dk = _grad_out(dk_, k, "dk", (total_k, num_heads_k, head_size))
    dv = _grad_out(dv_, v, "dv", (total_k, num_heads_k, head_size))

    with torch.cuda.device(q.device):
        softmax_d = torch.empty((num_heads, total_q + 128 * batch_size), dtype=torch.float32, device=q.device)
        if zero_tensors:
            dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_()
        if max_seqlen_q > 0 and total_q > 0 and total_k > 0:
            ins = [x if _aligned(x) else x.contiguous() for x in (dout, q, k, v, out)]
            outs = [x if _aligned(x) else torch.empty_like(x, memory_format=torch.contiguous_format) for x in (dq, dk, dv)]
            lse = softmax_lse if softmax_lse.is_contiguous() else softmax_lse.contiguous()
            _dispatch.launch_bwd(*ins, lse, *outs, softmax_d, varlen=True, batch=batch_size,
                                 max_seqlen_q=max_seqlen_q, max_seqlen_k=max_seqlen_k, softmax_scale=softmax_scale,
                                 causal=is_causal, window_left=window_size_left, window_right=window_size_right,
                                 softcap=softcap, cu_seqlens_q=cu_seqlens_q, cu_seqlens_k=cu_seqlens_k,
                                 alibi_slopes=alibi, deterministic=deterministic, p_dropout=p_dropout,
                                 rng_state=rng_state if p_dropout > 0 else None, fa3_window=getattr(_window_rule, "fa3", False))
            for dst, src in zip((dq, dk, dv), outs):
                if dst is not src:
                    dst.copy_(src)
        else:
            dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_()
    return [dq, dk, dv, softmax_d]


def fwd_kvcache(q: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, k_: Optional[torch.Tensor],
                v_: Optional[torch.Tensor], seqlens_k_: Optional[torch.Tensor], rotary_cos_: Optional[torch.Tensor],
                rotary_sin_: Optional[torch.Tensor], cache_batch_idx_: Optional[torch.Tensor],
                leftpad_k_: Optional[torch.Tensor], block_table_: Optional[torch.Tensor],
                alibi_slopes_: Optional[torch.Tensor], out_: Optional[torch.Tensor], softmax_scale: float,
                is_causal: bool, window_size_left: int, window_size_right: int, softcap: float,
                is_rotary_interleaved: bool, num_splits: int) -> List[torch.Tensor]:
    """mha_fwd_kvcache, csrc/flash_attn/flash_api.cpp:1202-1476 (20 positional arguments).  Returns [out, softmax_lse]; see
    _fwd_kvcache_impl.  Paged caches follow the reference's rule here: page size divisible by 256 (:1265)."""
    return _fwd_kvcache_impl(q, kcache, vcache, k_, v_, seqlens_k_, rotary_cos_, rotary_sin_, cache_batch_idx_, leftpad_k_,
                             block_table_, alibi_slopes_, out_, softmax_scale, is_causal, window_size_left,
                             window_size_right, softcap, is_rotary_interleaved, num_splits, 256)


def _fwd_kvcache_impl(q: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, k_: Optional[torch.Tensor],
                v_: Optional[torch.Tensor], seqlens_k_: Optional[torch.Tensor], rotary_cos_: Optional[torch.Tensor],
                rotary_sin_: Optional[torch.Tensor], cache_batch_idx_: Optional[torch.Tensor],
                leftpad_k_: Optional[torch.Tensor], block_table_: Optional[torch.Tensor],
                alibi_slopes_: Optional[torch.Tensor], out_: Optional[torch.Tensor], softmax_scale: float,
                is_causal: bool, window_size_left: int, window_size_right: int, softcap: float,
                is_rotary_interleaved: bool, num_splits: int, _page_multiple: int,
                seqlens_rotary_: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """mha_fwd_kvcache, csrc/flash_attn/flash_api.cpp:1202-1476.  Returns [out, softmax_lse].

    Built: in-place append of k_/v_ at seqlens_k_ (keys optionally rotated), attention over the first seqlens_k_
    (+ appended) rows of each cache entry, cache_batch_idx_, paged caches (block_table_, page size % 256 == 0),
    causal / window / softcap / ALiBi, rotary embedding of q, the (b, 1, h) -> (b, ngroups, h_k) GQA swap (:1277-1285),
    split-KV (num_splits: 0 = library heuristic, 1 = off, N = forced), left-padded caches (leftpad_k_)."""
    _lib.load()
    q_dtype = q.dtype
    _check(q_dtype in (torch.float16, torch.bfloat16), "FlashAttention only support fp16 and bf16 data type")
    _check(kcache.dtype == q_dtype, "query and key must have the same dtype")
    _check(vcache.dtype == q_dtype, "query and value must have the same dtype")
    _check_device(q, "q"); _check_device(kcache, "kcache"); _check_device(vcache, "vcache")
    for t in (q, kcache, vcache):
        _check(t.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    paged = block_table_ is not None
    if paged:
        _check(cache_batch_idx_ is None, "Paged KVcache does not support cache_batch_idx")

    batch_size, seqlen_q, num_heads, head_size_og = q.shape
    batch_size_c, seqlen_k, num_heads_k = kcache.shape[0], kcache.shape[1], kcache.shape[2]
    if paged:
        page_block_size, max_blocks = _check_block_table(block_table_, kcache, batch_size, _page_multiple)
        seqlen_k, batch_size_c = max_blocks * page_block_size, batch_size  # (:1266-1268)
    _check(batch_size > 0, "batch size must be positive")
    _check(head_size_og <= 256, "FlashAttention forward only supports head dimension at most 256")
    _check(head_size_og % 8 == 0, "This flash attention build needs head_size to be a multiple of 8 in fwd_kvcache")
    _check(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query")
    alibi = _check_alibi(alibi_slopes_, batch_size, num_heads)

    if seqlen_q == 1 and alibi is None:
        is_causal = False  # (:1270)
    if is_causal:
        window_size_right = 0
    # (b, 1, (h_k ngroups), d) -> (b, ngroups, h_k, d): one pass over the cache serves the whole GQA group (:1272-1285)
    swapped = (seqlen_q == 1 and num_heads > num_heads_k and window_size_left < 0 and window_size_right < 0
               and alibi is None)
    if swapped:
        ngroups = num_heads // num_heads_k
        q = q.reshape(batch_size, num_heads_k, ngroups, head_size_og).transpose(1, 2)
        seqlen_q, num_heads = ngroups, num_heads_k

    _check_shape(q, "q", batch_size, seqlen_q, num_heads, head_size_og)
    if paged:
        _check_shape(kcache, "kcache", kcache.shape[0], page_block_size, num_heads_k, head_size_og)
        _check_shape(vcache, "vcache", kcache.shape[0], page_block_size, num_heads_k, head_size_og)
    else:
        _check_shape(kcache, "kcache", batch_size_c, seqlen_k, num_heads_k, head_size_og)
        _check_shape(vcache, "vcache", batch_size_c, seqlen_k, num_heads_k, head_size_og)

    if out_ is not None and not swapped:
        out = out_
        _check(out.dtype == q_dtype, "Output must have the same dtype as inputs")
        _check_device(out, "out")
        _check(out.stride(-1) == 1, "Output tensor must have contiguous last dimension")
        _check_shape(out, "out", batch_size, seqlen_q, num_heads, head_size_og)
    else:
        out = torch.empty((batch_size, seqlen_q, num_heads, head_size_og), dtype=q_dtype, device=q.device)

    seqlen_knew = 0
    if k_ is not None:
        _check(v_ is not None, "If key is supplied, value must also be passed in")
        _check(seqlens_k_ is not None, "If key is supplied, seqlens_k must also be passed in")
        _check(seqlen_q <= seqlen_k, "If key is supplied, it must have seqlen <= the seqlen of the KV cache")
        _check(k_.dtype == q_dtype, "Key must have the same dtype as query")
        _check(v_.dtype == q_dtype, "Value must have the same dtype as query")
        _check_device(k_, "k"); _check_device(v_, "v")
        _check(k_.stride(-1) == 1, "Key tensor must have contiguous last dimension")
        _check(v_.stride(-1) == 1, "Value tensor must have contiguous last dimension")
        seqlen_knew = k_.shape[1]
        _check_shape(k_, "k", batch_size, seqlen_knew, num_heads_k, head_size_og)
        _check_shape(v_, "v", batch_size, seqlen_knew, num_heads_k, head_size_og)
    if seqlens_k_ is not None:
        _check(seqlens_k_.dtype == torch.int32, "seqlens_k must have dtype int32")
        _check_device(seqlens_k_, "seqlens_k")
        _check(seqlens_k_.is_contiguous(), "seqlens_k must be contiguous")
        _check_shape(seqlens_k_, "seqlens_k", batch_size)
    _check_leftpad(leftpad_k_, batch_size, paged)
    if cache_batch_idx_ is not None:
        _check_device(cache_batch_idx_, "cache_batch_idx")
        _check(cache_batch_idx_.is_contiguous(), "cache_batch_idx must be contiguous")
        _check(cache_batch_idx_.dtype == torch.int32, "cache_batch_idx must have dtype int32")
    else:
        _check(batch_size_c >= batch_size, "the KV cache must have at least batch_size entries")
    for t in (kcache, vcache):
        _check(_aligned(t), "the KV cache must be 16-byte aligned with row/head/batch strides that are multiples of 8")
    rotary = rotary_cos_ is not None
    if rotary:  # (:1404-1428)
        _check(k_ is not None, "If rotary cos/sin are provided, new key / value to be appended to KV cache must also be provided")
        _check_device(rotary_cos_, "rotary_cos")
        rotary_dim = rotary_cos_.shape[1] * 2
        _check(rotary_dim <= head_size_og, "rotary_dim must be <= headdim")
        _check(rotary_dim % 16 == 0, "Only rotary dimensions divisible by 16 are currently supported")
        seqlen_ro = rotary_cos_.shape[0]
        _check(seqlen_ro >= seqlen_k, "cos/sin seqlen must be at least the seqlen of KV cache")
        _check_shape(rotary_cos_, "rotary_cos", seqlen_ro, rotary_dim // 2)
        _check(rotary_cos_.is_contiguous(), "rotary_cos must be contiguous")
        _check(rotary_cos_.dtype == q_dtype, "rotary_cos must have the same dtype as query")
        _check(rotary_sin_ is not None, "If rotary cos is provided, rotary sin must also be provided")
        _check_device(rotary_sin_, "rotary_sin")
        _check_shape(rotary_sin_, "rotary_sin", seqlen_ro, rotary_dim // 2)
        _check(rotary_sin_.is_contiguous(), "rotary_sin must be contiguous")
        _check(rotary_sin_.dtype == q_dtype, "rotary_cos must have the same dtype as query")

    with torch.cuda.device(q.device):
        softmax_lse = torch.empty((batch_size, num_heads, seqlen_q), dtype=torch.float32, device=q.device)
        seqused = seqlens_k_
        if seqlen_knew > 0:  # "Append_KV": new rows land at [seqlens_k, seqlens_k + seqlen_knew) of each cache entry
            kn, vn = (x if _aligned(x) else x.contiguous() for x in (k_, v_))
            _dispatch.kvcache_append(kn, vn, kcache, vcache, seqlens_k_, cache_batch_idx_, block_table_,
                                     rotary_cos_, rotary_sin_, is_rotary_interleaved, seqlens_rotary_)
            seqused = seqlens_k_ + seqlen_knew
        qc = q if _aligned(q) else q.contiguous()
        if rotary:
            # causal / local: query row i sits at position seqlens_k + i; otherwise every row at seqlens_k
            # (flash_attn/flash_attn_interface.py:1516-1524, src/flash_fwd_kernel.h:753-775)
            per_row = is_causal or window_size_left >= 0 or window_size_right >= 0
            q_ro = torch.empty_like(qc, memory_format=torch.contiguous_format)
            # (FA3 seqlens_rotary: the rotary positions when they are not the cache fill levels, hopper/seqlen.h:89)
            _dispatch.rotary_apply(qc, q_ro, rotary_cos_, rotary_sin_, seqlens_rotary_ if seqlens_rotary_ is not None else seqlens_k_,
                                   is_rotary_interleaved, per_row)
            qc = q_ro
        oc = out if _aligned(out) else torch.empty_like(out, memory_format=torch.contiguous_format)
        if seqlen_k > 0:
            _dispatch.launch(qc, kcache, vcache, oc, softmax_lse, varlen=False, batch=batch_size,
                             max_seqlen_q=seqlen_q, max_seqlen_k=seqlen_k, softmax_scale=softmax_scale,
                             causal=is_causal, window_left=window_size_left, window_right=window_size_right,
                             softcap=softcap, seqused_k=seqused, alibi_slopes=alibi, kv_batch_idx=cache_batch_idx_,
                             block_table=block_table_, num_splits=num_splits, leftpad_k=leftpad_k_)
            if oc is not out:
                out.copy_(oc)
        else:
            out.zero_()
            softmax_lse.fill_(math.inf)
    if swapped:
        out = out.transpose(1, 2).reshape(batch_size, 1, num_heads_k * seqlen_q, head_size_og)
        softmax_lse = softmax_lse.reshape(batch_size, num_heads_k * seqlen_q, 1)
        if out_ is not None:
            out_.copy_(out)
            out = out_
    return [out, softmax_lse]


# ---- the compiled binding takes over the five entry points when it is built (see the module docstring) -----------------
_py_entry_points = {"fwd": fwd, "varlen_fwd": varlen_fwd, "bwd": bwd, "varlen_bwd": varlen_bwd, "fwd_kvcache": fwd_kvcache,
                    "_fwd_kvcache_impl": _fwd_kvcache_impl, "fa3_window_rule": fa3_window_rule}
compiled = None
if os.environ.get("FA_BINDING", "compiled") != "python":
    try:
        from . import flash_attn_2_cuda_C as compiled  # noqa: F811
    except ImportError:
        compiled = None
if compiled is not None:
    fwd, varlen_fwd, bwd, varlen_bwd, fwd_kvcache = (compiled.fwd, compiled.varlen_fwd, compiled.bwd, compiled.varlen_bwd,
                                                     compiled.fwd_kvcache)
    _fwd_kvcache_impl = compiled._fwd_kvcache_impl

    @contextlib.contextmanager
    def fa3_window_rule():  # noqa: F811
        prev = compiled._set_fa3_window_rule(True)
        try:
            yield
        finally:
            compiled._set_fa3_window_rule(prev)
