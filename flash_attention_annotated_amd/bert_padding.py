"""Ragged-batch helpers either side of the varlen forward (role of flash_attn/bert_padding.py:98-218).

`unpad_input` packs the valid rows of a padded (batch, seqlen, ...) tensor and produces the
`cu_seqlens` the varlen kernel indexes with; `pad_input` scatters packed rows back.
"""
import torch
import torch.nn.functional as F


def unpad_input(hidden_states, attention_mask, unused_mask=None):
    """hidden_states: (batch, seqlen, ...); attention_mask: (batch, seqlen) bool/int, 1 = valid.

    Returns (packed (total, ...), indices (total,), cu_seqlens (batch+1,) int32, max_seqlen_in_batch,
    seqused (batch,) int32) — the 5-tuple of the reference (flash_attn/bert_padding.py:98-128).
    `unused_mask` marks tokens that are allocated (kept in the packing) but not attended to.
    """
    all_masks = (attention_mask + unused_mask) if unused_mask is not None else attention_mask
    seqlens_in_batch = all_masks.sum(dim=-1, dtype=torch.int32)
    used_seqlens_in_batch = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(all_masks.flatten(), as_tuple=False).flatten()
    max_seqlen_in_batch = int(seqlens_in_batch.max().item()) if seqlens_in_batch.numel() else 0
    cu_seqlens = F.pad(torch.cumsum(seqlens_in_batch, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape(hidden_states.shape[0] * hidden_states.shape[1], *hidden_states.shape[2:])
    return flat[indices], indices, cu_seqlens, max_seqlen_in_batch, used_seqlens_in_batch


def pad_input(hidden_states, indices, batch, seqlen):
    """hidden_states: (total, ...) -> (batch, seqlen, ...) with zeros at padded positions."""
    out = torch.zeros((batch * seqlen, *hidden_states.shape[1:]), device=hidden_states.device,
                      dtype=hidden_states.dtype)
    out[indices] = hidden_states
    return out.reshape(batch, seqlen, *hidden_states.shape[1:])
