"""The two attention modules of the reference that sit directly on the hot path:
`FlashSelfAttention` / `FlashCrossAttention` (reference flash_attn/modules/mha.py:53-141, 144-228).  Same constructor
arguments, same forward signatures; they only choose between the padded and the packed (cu_seqlens) entry point.
The rest of that file (the MHA block with projections, rotary and the inference cache) is outside SURVEY.md §8.
"""
import torch
import torch.nn as nn

from ..flash_attn_interface import (
    flash_attn_kvpacked_func,
    flash_attn_qkvpacked_func,
    flash_attn_varlen_kvpacked_func,
    flash_attn_varlen_qkvpacked_func,
)


class _AttnBase(nn.Module):
    def __init__(self, causal, softmax_scale, attention_dropout, window_size, alibi_slopes, deterministic):
        super().__init__()
        self.causal = causal
        self.softmax_scale = softmax_scale
        self.drop = nn.Dropout(attention_dropout)
        self.register_buffer("alibi_slopes", alibi_slopes, persistent=False)
        self.window_size = window_size
        self.deterministic = deterministic

    def _common(self, causal):
        if self.alibi_slopes is not None:
            self.alibi_slopes = self.alibi_slopes.to(torch.float32)
        return dict(softmax_scale=self.softmax_scale, causal=self.causal if causal is None else causal,
                    alibi_slopes=self.alibi_slopes, window_size=self.window_size, deterministic=self.deterministic)


class FlashSelfAttention(_AttnBase):
    """softmax(Q K^T * scale) V on a packed qkv tensor (reference :53-141)."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0, window_size=(-1, -1),
                 alibi_slopes=None, deterministic=False):
        super().__init__(causal, softmax_scale, attention_dropout, window_size, alibi_slopes, deterministic)

    def forward(self, qkv, causal=None, cu_seqlens=None, max_seqlen=None):
        """qkv: (B, S, 3, H, D), or (total, 3, H, D) with cu_seqlens (int32, (B+1,)) and max_seqlen (int).
        Returns (B, S, H, D) or (total, H, D)."""
        assert qkv.dtype in [torch.float16, torch.bfloat16]
        assert qkv.is_cuda
        p = self.drop.p if self.training else 0.0
        if cu_seqlens is not None:
            assert cu_seqlens.dtype == torch.int32
            assert max_seqlen is not None and isinstance(max_seqlen, int)
            return flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, p, **self._common(causal))
        return flash_attn_qkvpacked_func(qkv, p, **self._common(causal))


class FlashCrossAttention(_AttnBase):
    """Attention of q over a packed kv tensor (reference :144-228)."""

    def __init__(self, causal=False, softmax_scale=None, attention_dropout=0.0, alibi_slopes=None,
                 window_size=(-1, -1), deterministic=False):
        super().__init__(causal, softmax_scale, attention_dropout, window_size, alibi_slopes, deterministic)

    def forward(self, q, kv, causal=None, cu_seqlens=None, max_seqlen=None, cu_seqlens_k=None, max_seqlen_k=None):
        """q: (B, Sq, H, D), kv: (B, Sk, 2, H_k, D); or packed (total_q, H, D) / (total_k, 2, H_k, D) with
        cu_seqlens / cu_seqlens_k and the two max lengths."""
        assert q.dtype in [torch.float16, torch.bfloat16]
        assert q.is_cuda and kv.is_cuda
        p = self.drop.p if self.training else 0.0
        if cu_seqlens is not None:
            assert cu_seqlens.dtype == torch.int32
            assert max_seqlen is not None and isinstance(max_seqlen, int)
            assert cu_seqlens_k is not None and cu_seqlens_k.dtype == torch.int32
            assert max_seqlen_k is not None and isinstance(max_seqlen_k, int)
            return flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens, cu_seqlens_k, max_seqlen, max_seqlen_k, p,
                                                   **self._common(causal))
        assert kv.shape[0] == q.shape[0] and kv.shape[4] == q.shape[3]
        return flash_attn_kvpacked_func(q, kv, p, **self._common(causal))
