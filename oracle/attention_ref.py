"""ORACLE — test infrastructure only.  Never imported by the product path.

CPU/torch restatement of the reference's exact-attention oracle `attention_ref`
(tests/test_util.py:185-274 == tests/test_flash_attn.py:217-304; FA3 flavour
hopper/test_util.py:226-348) and its mask builder `construct_local_mask`
(tests/test_util.py:150-182).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module.

Pinned: `oracle/make_golden.py` (run in the build container, where /root/reference is
mounted) asserts this restatement equals the reference's own `attention_ref` on a sweep
(dense / causal / sq != sk / GQA / padding masks / local windows / softcap / low-precision
reorder) and freezes inputs+outputs into tests/golden/*.pt; `tests/test_oracle.py` replays
those fixtures without the reference.
"""
import math

import torch


def local_mask(seqlen_q, seqlen_k, window_size=(-1, -1), query_padding_mask=None, key_padding_mask=None,
               device=None, key_leftpad=None):
    """True where key j is NOT visible from query i.  Bottom-right aligned:
    visible iff  i + sk - sq - left <= j <= min(i + sk - sq + right, sk)   (tests/test_util.py:150-182).
    With padding masks sk / sq are the per-batch valid lengths and the result is (b,1,sq,sk)."""
    i = torch.arange(seqlen_q, device=device, dtype=torch.long).view(-1, 1)
    j = torch.arange(seqlen_k, device=device, dtype=torch.long).view(1, -1)
    if key_leftpad is not None:  # columns count from the first real key of each batch entry (tests/test_util.py:166-169)
        lp = key_leftpad.long().view(-1, 1, 1, 1)
        j = j.view(1, 1, 1, -1).expand(lp.shape[0], 1, 1, seqlen_k)
        j = torch.where(j >= lp, j - lp, 2 ** 32)
    sk = seqlen_k if key_padding_mask is None else key_padding_mask.sum(-1).view(-1, 1, 1, 1)
    sq = seqlen_q if query_padding_mask is None else query_padding_mask.sum(-1).view(-1, 1, 1, 1)
    left, right = window_size
    diag = i + sk - sq
    if left < 0:
        return j > diag + right
    sk_t = torch.full_like(j, seqlen_k) if key_padding_mask is None else sk
    return torch.logical_or(j > torch.minimum(diag + right, sk_t), j < diag - left)


def chunk_mask(seqlen_q, seqlen_k, attention_chunk, query_padding_mask=None, key_padding_mask=None, device=None,
               key_leftpad=None):
    """True where key j lies outside the attention chunk of query i (FA3 `attention_chunk`, construct_chunk_mask
    hopper/test_util.py:193-223): visible iff  c <= j < c + chunk  with  c = (i + sk - sq) - (i + sk - sq) % chunk  (floor
    remainder, so rows whose diagonal position is negative see a chunk that ends at or before key 0: nothing)."""
    i = torch.arange(seqlen_q, device=device, dtype=torch.long).view(-1, 1)
    j = torch.arange(seqlen_k, device=device, dtype=torch.long).view(1, -1)
    if key_leftpad is not None:
        lp = key_leftpad.long().view(-1, 1, 1, 1)
        j = j.view(1, 1, 1, -1).expand(lp.shape[0], 1, 1, seqlen_k)
        j = torch.where(j >= lp, j - lp, 2 ** 32)
    sk = seqlen_k if key_padding_mask is None else key_padding_mask.sum(-1).view(-1, 1, 1, 1)
    sq = seqlen_q if query_padding_mask is None else query_padding_mask.sum(-1).view(-1, 1, 1, 1)
    diag = i + sk - sq
    lo = diag - torch.remainder(diag, attention_chunk)
    return torch.logical_or(j < lo, j >= lo + attention_chunk)


def attn_bias_from_alibi_slopes(slopes, seqlen_q, seqlen_k, query_padding_mask=None, key_padding_mask=None,
                                causal=False, key_leftpad=None):
    """ALiBi bias (b, h, sq|1, sk) from fp32 slopes (b, h): tests/test_flash_attn.py:29-56.
    non-causal: -slope * |i + sk - sq - j| (sk, sq = per-batch valid lengths under padding masks);
    causal: slope * (j - seqlen_k + 1), the same for every row (equal to the general form up to a per-row constant,
    which softmax cancels)."""
    b, h = slopes.shape
    sl = slopes.view(b, h, 1, 1)
    if causal:
        return torch.arange(-seqlen_k + 1, 1, dtype=torch.float32, device=slopes.device) * sl
    i = torch.arange(seqlen_q, dtype=torch.long, device=slopes.device).view(-1, 1)
    j = torch.arange(seqlen_k, dtype=torch.long, device=slopes.device)
    if key_leftpad is not None:  # columns count from the first real key; padding columns get a huge distance (:40-43)
        lp = key_leftpad.long().view(-1, 1, 1, 1)
        j = j.view(1, 1, 1, -1).expand(lp.shape[0], 1, 1, seqlen_k)
        j = torch.where(j >= lp, j - lp, 2 ** 32)
    sk = seqlen_k if key_padding_mask is None else key_padding_mask.sum(-1).view(-1, 1, 1, 1)
    sq = seqlen_q if query_padding_mask is None else query_padding_mask.sum(-1).view(-1, 1, 1, 1)
    return -sl * torch.abs(i + sk - sq - j).to(slopes.dtype)


def apply_rotary_emb_ref(x, cos, sin, seqlen_offsets, interleaved=False, per_row_positions=True):
    """Rotary embedding of x (b, s, h, d) with per-batch position offsets: restates apply_rotary_emb_torch
    (flash_attn/layers/rotary.py:14-36) called the way tests/test_flash_attn.py:2017-2036 calls apply_rotary_emb
    (seqlen_offsets = cache_seqlens; the non-causal query case folds s into the head dim, i.e. one position).
    cos, sin: (seqlen_ro, rotary_dim / 2).  Math in fp32, result in x.dtype."""
    b, s, h, d = x.shape
    rd = cos.shape[-1] * 2
    pos = seqlen_offsets.long().view(b, 1) + (torch.arange(s).view(1, s) if per_row_positions else 0)
    pos = pos.expand(b, s)
    c, sn = cos.float()[pos], sin.float()[pos]  # (b, s, rd/2)
    if interleaved:
        c, sn = c.repeat_interleave(2, dim=-1), sn.repeat_interleave(2, dim=-1)
    else:
        c, sn = torch.cat([c, c], dim=-1), torch.cat([sn, sn], dim=-1)
    xr = x[..., :rd].float()
    if interleaved:
        x1, x2 = xr[..., ::2], xr[..., 1::2]
        rot = torch.stack((-x2, x1), dim=-1).flatten(-2)
    else:
        x1, x2 = xr.chunk(2, dim=-1)
        rot = torch.cat((-x2, x1), dim=-1)
    out = xr * c[:, :, None, :] + rot * sn[:, :, None, :]
    return torch.cat([out.to(x.dtype), x[..., rd:]], dim=-1)


def attention_ref(q, k, v, query_padding_mask=None, key_padding_mask=None, attn_bias=None, causal=False,
                  window_size=(-1, -1), softcap=0.0, upcast=True, reorder_ops=False, return_lse=False,
                  q_descale=None, k_descale=None, v_descale=None, intermediate_dtype=None, key_leftpad=None, dropout_p=0.0, dropout_mask=None,
                  attention_chunk=0):
    """Exact softmax attention.

    q: (b, sq, h, d); k: (b, sk, h_k, d), v: (b, sk, h_k, dv) with h % h_k == 0 (kv head = q head // (h/h_k)); dv may differ
    from d (FA3 headdim_v, hopper/test_util.py:245-246): out is (b, sq, h, dv).  attention_chunk > 0: chunk_mask() on top of the
    window mask (hopper/test_util.py:310-320).
    upcast=True  -> everything in fp32 ("out_ref" of the reference's tests);
    upcast=False, reorder_ops=True -> same math in the input precision with k scaled instead of q
    ("out_pt", the yardstick of the tolerance contract, tests/test_flash_attn.py:1121).
    Returns (out (b,sq,h,d) in q.dtype, attention (b,h,sq,sk)) and, if return_lse, the fp32
    logsumexp (b,h,sq) of the masked, scaled scores (+inf where no key is visible, the FA2
    convention csrc/flash_attn/src/softmax.h:178-180).
    """
    if causal:
        window_size = (window_size[0], 0)
    dtype_og = q.dtype
    if upcast:
        q, k, v = q.float(), k.float(), v.float()
    # fp8 descales, per (batch, kv head): hopper/test_util.py:272-279
    if q_descale is not None:
        q = (q.float() * q_descale.repeat_interleave(q.shape[2] // k.shape[2], dim=1)[:, None, :, None]).to(q.dtype)
    if k_descale is not None:
        k = (k.float() * k_descale[:, None, :, None]).to(k.dtype)
    if v_descale is not None:
        v = (v.float() * v_descale[:, None, :, None]).to(v.dtype)
    b, sq, h, d = q.shape
    sk = k.shape[1]
    g = h // k.shape[2]
    k = k.repeat_interleave(g, dim=2)
    v = v.repeat_interleave(g, dim=2)
    if not reorder_ops:
        scores = torch.einsum("bthd,bshd->bhts", q / math.sqrt(d), k)
    else:
        scores = torch.einsum("bthd,bshd->bhts", q, k / math.sqrt(d))
    if softcap > 0:
        scores = torch.tanh(scores / softcap) * softcap
    if key_padding_mask is not None:
        scores = scores.masked_fill(~key_padding_mask.view(b, 1, 1, sk), float("-inf"))
    masked = None
    if window_size[0] >= 0 or window_size[1] >= 0:
        masked = local_mask(sq, sk, window_size, query_padding_mask, key_padding_mask, q.device, key_leftpad)
    if attention_chunk > 0:
        cm = chunk_mask(sq, sk, attention_chunk, query_padding_mask, key_padding_mask, q.device, key_leftpad)
        masked = cm if masked is None else torch.logical_or(masked, cm)
    if masked is not None:
        scores = scores.masked_fill(masked, float("-inf"))
    if attn_bias is not None:
        scores = scores + attn_bias
    lse = torch.logsumexp(scores.float(), dim=-1)
    attention = torch.softmax(scores, dim=-1).to(v.dtype)
    if masked is not None:  # fully masked rows: zeros instead of NaN
        attention = attention.masked_fill(torch.all(masked, dim=-1, keepdim=True), 0.0)
    if query_padding_mask is not None:
        attention = attention.masked_fill(~query_padding_mask.view(b, 1, sq, 1), 0.0)
    attention_pv = attention
    if dropout_mask is not None:  # (b, h, sq, sk) bool, True = kept; the 1/(1-p) goes onto v (tests/test_util.py:262-269)
        attention_pv = attention.masked_fill(~dropout_mask, 0.0)
    if intermediate_dtype is not None:  # P rounded through e.g. e4m3 (hopper/test_util.py:343-344)
        attention_pv = attention.to(intermediate_dtype).to(attention.dtype)
    out = torch.einsum("bhts,bshd->bthd", attention_pv, v * (1.0 / (1 - dropout_p)))
    if query_padding_mask is not None:
        out = out.masked_fill(~query_padding_mask.view(b, sq, 1, 1), 0.0)
    if key_padding_mask is not None:
        out = out.masked_fill(~torch.any(key_padding_mask, 1).view(b, 1, 1, 1), 0.0)
    out = out.to(dtype_og)
    if return_lse:
        lse = torch.where(torch.isneginf(lse), torch.full_like(lse, float("inf")), lse)
        return out, attention.to(dtype_og), lse
    return out, attention.to(dtype_og)


def attention_varlen_ref(q, k, v, cu_seqlens_q, cu_seqlens_k, causal=False, window_size=(-1, -1), softcap=0.0,
                         upcast=True, reorder_ops=False, seqused_k=None):
    """Packed ragged batch: q (total_q,h,d), k/v (total_k,h_k,d).  Loops over sequences with
    attention_ref; returns out (total_q,h,d) and lse (h,total_q) — the layout of
    mha_varlen_fwd (csrc/flash_attn/flash_api.cpp:652)."""
    cq = cu_seqlens_q.tolist()
    ck = cu_seqlens_k.tolist()
    out = torch.zeros_like(q)
    lse = torch.full((q.shape[1], q.shape[0]), float("inf"), dtype=torch.float32, device=q.device)
    for i in range(len(cq) - 1):
        q0, q1, k0, k1 = cq[i], cq[i + 1], ck[i], ck[i + 1]
        if seqused_k is not None:
            k1 = k0 + int(seqused_k[i])
        if q1 == q0:
            continue
        if k1 == k0:
            continue  # zero keys: out = 0, lse = +inf
        o, _, l = attention_ref(q[q0:q1][None], k[k0:k1][None], v[k0:k1][None], causal=causal,
                                window_size=window_size, softcap=softcap, upcast=upcast,
                                reorder_ops=reorder_ops, return_lse=True)
        out[q0:q1] = o[0]
        lse[:, q0:q1] = l[0]
    return out, lse


def sdpa_cpu(q, k, v, causal=False):
    """PyTorch-eager fused CPU attention on (b,s,h,d) tensors — the CPU baseline BASELINE.json names
    ("PyTorch-eager SDPA timed on the same box's host CPUs").  Top-left causal == bottom-right when sq == sk."""
    import torch.nn.functional as F
    g = q.shape[2] // k.shape[2]
    if g > 1:
        k = k.repeat_interleave(g, dim=2)
        v = v.repeat_interleave(g, dim=2)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=causal)
    return o.transpose(1, 2)


def attention_combine_ref(out_partial, lse_partial):
    """Merge of split-KV partials (hopper/test_flash_attn.py:1105-1114).
    out_partial: (num_splits, b, seqlen, h, d); lse_partial: (num_splits, b, seqlen, h) -> out (b, seqlen, h, d), lse."""
    lse = torch.logsumexp(lse_partial, dim=0)
    scale = torch.exp(lse_partial - lse)
    scale = torch.where(torch.isinf(scale) | torch.isnan(scale), torch.zeros_like(scale), scale)
    out = (scale.unsqueeze(-1) * out_partial).sum(0)
    return out, lse


def sdmask_block_size_n(head_dim, is_dropout, is_causal):
    """The reference forward's key-block width for a head dim on a device that is neither sm8x nor sm90
    (flash_attn/flash_attn_interface.py:23-46 `_get_block_size_n`): the blocks behind the running maxima of S_dmask."""
    if head_dim <= 32:
        return 128
    if head_dim <= 64:
        return 128 if not is_dropout else 64
    if head_dim <= 96:
        return 64
    if head_dim <= 128:
        return 64 if not is_dropout else 32
    return 64


def convert_flash_attn_S_to_softmax(S, seqlen_q, seqlen_k, query_padding_mask, key_padding_mask, causal=False,
                                    window_size=(-1, -1)):
    """tests/test_flash_attn.py:411-463: the (b, h, seqlen_q rounded, seqlen_k rounded) tensor `return_softmax` hands back, cut to
    (b, h, seqlen_q, seqlen_k) with everything the attention does not look at zeroed.  Its sign is the dropout decision
    (>= 0: kept), its magnitude exp(score - running max of the key block)."""
    if causal:
        window_size = (window_size[0], 0)
    seqlen_q_rounded, seqlen_k_rounded = S.shape[-2:]
    S_converted = S
    if window_size[0] >= 0 or window_size[1] >= 0:
        lm = local_mask(seqlen_q, seqlen_k, window_size, query_padding_mask, key_padding_mask)
        lm = torch.nn.functional.pad(lm, (0, seqlen_k_rounded - seqlen_k, 0, seqlen_q_rounded - seqlen_q), value=True)
        S_converted = S_converted.masked_fill(lm, 0.0)
    seqlen_q_og = query_padding_mask.shape[-1] if query_padding_mask is not None else seqlen_q_rounded
    if query_padding_mask is not None:
        qm = torch.nn.functional.pad(query_padding_mask, (0, seqlen_q_rounded - seqlen_q_og))
        S_converted = S_converted.masked_fill(~qm[:, None, :, None], 0.0)
    seqlen_k_og = key_padding_mask.shape[-1] if key_padding_mask is not None else seqlen_k
    if key_padding_mask is not None:
        km = torch.nn.functional.pad(key_padding_mask, (0, seqlen_k_rounded - seqlen_k_og))
        S_converted = S_converted.masked_fill(~km[:, None, None, :], 0.0)
    S_converted = torch.nn.functional.pad(S_converted, (0, 0, 0, seqlen_q_og - seqlen_q_rounded))
    S_converted = torch.nn.functional.pad(S_converted, (0, seqlen_k_og - seqlen_k_rounded))
    return S_converted[:, :, :seqlen_q, :seqlen_k]


def normalize_flash_attn_S(attn_unnorm, q, k, v, query_padding_mask=None, key_padding_mask=None, attn_bias=None,
                           is_dropout=False, causal=False, window_size=(-1, -1), block_size_n=None):
    """tests/test_flash_attn.py:466-526: turns |S_dmask| into the softmax probabilities -- each key block of `block_size_n`
    keys carries exp(score - m) with m the maximum over its own and all LATER blocks (the reference's sweep runs from the last
    key block to the first), so a * exp(m - lse) is the probability."""
    if causal:
        window_size = (window_size[0], 0)
    q, k, v = q.float(), k.float(), v.float()
    _, seqlen_q, _, head_dim = q.shape
    seqlen_k = k.shape[1]
    if block_size_n is None:
        block_size_n = sdmask_block_size_n(head_dim, is_dropout, causal)
    k = k.repeat_interleave(q.shape[2] // k.shape[2], dim=2)
    scores = torch.einsum("bthd,bshd->bhts", q / math.sqrt(head_dim), k)
    if key_padding_mask is not None:
        scores.masked_fill_(~key_padding_mask[:, None, None, :], float("-inf"))
    if window_size[0] >= 0 or window_size[1] >= 0:
        lm = local_mask(seqlen_q, seqlen_k, window_size, query_padding_mask, key_padding_mask)
        scores.masked_fill_(lm, float("-inf"))
    if attn_bias is not None:
        scores = scores + attn_bias.to(dtype=scores.dtype)
    scores_block = scores.split(block_size_n, dim=-1)
    lse_block = torch.stack([torch.logsumexp(s, dim=-1) for s in scores_block], dim=-1)
    lse = torch.logsumexp(lse_block, dim=-1)
    lse[lse == float("-inf")] = float("inf")
    scores_max_block = torch.stack([torch.amax(s, dim=-1) for s in scores_block], dim=-1)
    cummax_block = torch.cummax(scores_max_block.flip(-1), dim=-1).values.flip(-1).unbind(dim=-1)
    attn_unnorm_block = attn_unnorm.split(block_size_n, dim=-1)
    attn_norm = torch.cat([a * torch.exp(m - lse)[..., None] for a, m in zip(attn_unnorm_block, cummax_block)], dim=-1)
    if query_padding_mask is not None:
        attn_norm.masked_fill_(~query_padding_mask[:, None, :, None], 0.0)
    return attn_norm.to(dtype=attn_unnorm.dtype)
