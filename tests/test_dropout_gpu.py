"""GPU parity tests of attention dropout through the public API.

Modelled on the reference's CUDA tests (tests/test_flash_attn.py:1007-1064, 1330-1380): `return_attn_probs=True` hands back
S_dmask -- (b, h, seqlen_q rounded to 128, seqlen_k rounded to 128) in the input dtype, the probabilities relative to the
running maximum of their key block with the dropout decision in the sign bit (csrc/flash_attn/src/dropout.h:26-33,
flash_fwd_kernel.h:350-360).  It is decoded with the restatement of the reference's own decoder
(oracle.convert_flash_attn_S_to_softmax / normalize_flash_attn_S, pinned to tests/test_flash_attn.py:411-526 by
oracle/make_golden.py): the sign gives the keep-mask the oracle is run with, the magnitude must normalise to the oracle's
attention probabilities, and then
    |out - out_ref|max <= 2 |out_pt - out_ref|max,   |dX - dX_ref|max <= 3 |dX_pt - dX_ref|max (+ atol)
and the measured drop fraction is within 0.01 of the effective p of the 8-bit decision (keep iff hash byte <=
floor(255 (1 - p))) (tests/test_flash_attn.py:1046-1064 `dropout_fraction`).  Floating point; bounds are in _bound / the asserts.
"""
import math

import pytest
import torch

from oracle import attention_ref as oracle

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _api():
    import flash_attention_annotated_amd as fa
    return fa


def _bound(ref, pt, mult):
    ref = ref.float()
    atol = 2 * (ref + 0.3 - 0.3 - ref).abs().max().item()
    return mult * (pt.float() - ref).abs().max().item() + atol + 1e-5


def _visible(sq, sk, causal, window, qm=None, km=None):
    """(b or 1, 1, sq, sk) bool: positions the attention looks at (to measure the drop fraction over)."""
    w = window
    if causal:
        w = (w[0], 0)
    vis = torch.ones(1, 1, sq, sk, dtype=torch.bool)
    if w != (-1, -1):
        vis = ~oracle.local_mask(sq, sk, w, None, None)
        vis = vis.view(1, 1, sq, sk) if vis.dim() == 2 else vis
    if qm is not None:
        vis = vis & qm.view(-1, 1, sq, 1)
    if km is not None:
        vis = vis & km.view(-1, 1, 1, sk)
    return vis


def _run_case(dtype, b, sq, sk, h, hk, d, p_drop, causal, window=(-1, -1), alibi=False, seed=0):
    fa = _api()
    gen = torch.Generator().manual_seed(seed)
    q = torch.randn(b, sq, h, d, generator=gen).to(dtype)
    k = torch.randn(b, sk, hk, d, generator=gen).to(dtype)
    v = torch.randn(b, sk, hk, d, generator=gen).to(dtype)
    g = torch.randn(b, sq, h, d, generator=gen).to(dtype)
    slopes = torch.rand(b, h, generator=gen) * 0.3 if alibi else None
    bias = oracle.attn_bias_from_alibi_slopes(slopes, sq, sk, causal=causal) if alibi else None
    ql, kl, vl = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    torch.manual_seed(1234 + seed)
    out, lse, S = fa.flash_attn_func(ql, kl, vl, p_drop, causal=causal, window_size=window,
                                     alibi_slopes=slopes.to(DEV) if alibi else None, return_attn_probs=True)
    r128 = lambda x: (x + 127) // 128 * 128
    assert S.dtype == dtype and tuple(S.shape) == (b, h, r128(sq), r128(sk))  # csrc/flash_attn/flash_api.cpp:436-449
    dq, dk, dv = torch.autograd.grad(out, (ql, kl, vl), g.to(DEV))
    S_conv = oracle.convert_flash_attn_S_to_softmax(S.cpu(), sq, sk, None, None, causal=causal, window_size=window)
    keep = S_conv >= 0                                   # tests/test_flash_attn.py:1021
    # the magnitudes normalise to the attention probabilities (:1023-1036; bound :1125-1127 `attn`)
    k_rep = k.repeat_interleave(h // hk, dim=2)
    attn = oracle.normalize_flash_attn_S(S_conv.abs(), q, k_rep, k_rep, None, None, bias, True, causal=causal,
                                         window_size=window)
    _, attn_ref = oracle.attention_ref(q, k, v, None, None, attn_bias=bias, causal=causal, window_size=window)
    _, attn_pt = oracle.attention_ref(q, k, v, None, None, attn_bias=bias, causal=causal, window_size=window, upcast=False,
                                      reorder_ops=True)
    aerr = (attn.float() - attn_ref.float()).abs().max().item()
    abound = 2 * (attn_pt.float() - attn_ref.float()).abs().max().item() + 2e-3
    assert aerr <= abound, f"attention probabilities decoded from S_dmask: {aerr:.3e} > {abound:.3e}"

    def run(**extra):
        q2, k2, v2 = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = oracle.attention_ref(q2, k2, v2, None, None, attn_bias=bias, dropout_p=p_drop, dropout_mask=keep,
                                 causal=causal, window_size=window, **extra)[0]
        return (o.detach(),) + torch.autograd.grad(o, (q2, k2, v2), g)
    ref, pt = run(), run(upcast=False, reorder_ops=True)
    for name, got, r, p_, mult in zip(("out", "dq", "dk", "dv"), (out, dq, dk, dv), ref, pt, (2, 3, 3, 3)):
        got = got.detach().float().cpu()
        assert torch.isfinite(got).all(), name
        err = (got - r.float()).abs().max().item()
        assert err <= _bound(r, p_, mult), f"{name}: {err:.3e} > {_bound(r, p_, mult):.3e}"
    vis = _visible(sq, sk, causal, window).expand(b, h, sq, sk)
    frac = ((~keep) & vis).sum().item() / max(1, vis.sum().item())
    return frac, S


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [32, 59, 64, 96, 128, 256])
@pytest.mark.parametrize("sq,sk", [(97, 97), (128, 203), (384, 256), (1, 239)])
def test_dropout_output_and_grads(sq, sk, d, causal, dtype):
    frac, _ = _run_case(dtype, 3, sq, sk, 4, 2, d, 0.17, causal)
    p_eff = 1 - (math.floor(255 * (1 - 0.17)) + 1) / 256  # the 8-bit grid: 212 of 256 values are kept
    if sq * sk >= 97 * 97:
        assert abs(frac - p_eff) <= 0.01, frac


@pytest.mark.parametrize("local", [False, True])
@pytest.mark.parametrize("alibi", [False, True])
@pytest.mark.parametrize("p_drop", [0.05, 0.5, 0.9])
def test_dropout_rates_local_alibi(p_drop, alibi, local):
    window = (37, 11) if local else (-1, -1)
    frac, _ = _run_case(torch.bfloat16, 2, 256, 320, 6, 6, 64, p_drop, False, window=window, alibi=alibi, seed=3)
    p_eff = 1 - (math.floor(255 * (1 - p_drop)) + 1) / 256
    assert abs(frac - p_eff) <= 0.01, (frac, p_eff)


def test_dropout_seed_semantics():
    """Same torch seed -> same mask and bit-identical output; a different seed or the next call -> another mask;
    different heads / batches get different masks; backward regenerates exactly the forward's mask (checked above
    through the gradients) and rng_state is the pair the forward drew."""
    fa = _api()
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, 256, 4, 64, device=DEV, dtype=torch.float16) for _ in range(3))
    torch.manual_seed(7)
    o1, _, m1 = fa.flash_attn_func(q, k, v, 0.3, return_attn_probs=True)
    o2, _, m2 = fa.flash_attn_func(q, k, v, 0.3, return_attn_probs=True)  # generator advanced
    torch.manual_seed(7)
    o3, _, m3 = fa.flash_attn_func(q, k, v, 0.3, return_attn_probs=True)
    assert torch.equal(m1, m3) and torch.equal(o1, o3)
    assert not torch.equal(m1, m2)
    k1 = m1 >= 0   # the decisions
    assert not torch.equal(k1[0, 0], k1[0, 1]) and not torch.equal(k1[0, 0], k1[1, 0])
    assert abs((~k1[:, :, :256, :256]).float().mean().item() - (1 - (math.floor(255 * 0.7) + 1) / 256)) < 0.01
    # without return_attn_probs the same seed gives the same output (the mask does not depend on recording it)
    torch.manual_seed(7)
    o4 = fa.flash_attn_func(q, k, v, 0.3)
    assert torch.equal(o1, o4)
    # dropout_p = 0 is the plain kernel
    assert torch.equal(fa.flash_attn_func(q, k, v, 0.0), fa.flash_attn_func(q, k, v))


@pytest.mark.parametrize("causal", [False, True])
def test_dropout_varlen(causal):
    """flash_attn_varlen_func under dropout: S_dmask comes back as (b, h, max_seqlen_q rounded to 128, max_seqlen_k rounded to
    128), sequence i's block at [i, :, :seqlen_q_i, :seqlen_k_i] (csrc/flash_attn/flash_api.cpp:648-660); its signs give the
    keep-mask the oracle is run with."""
    fa = _api()
    gen = torch.Generator().manual_seed(11)
    b, h, hk, d, p_drop = 4, 4, 2, 64, 0.17
    lens_q, lens_k = [113, 1, 256, 77], [200, 130, 256, 31]
    msq, msk = max(lens_q), max(lens_k)
    q = torch.randn(sum(lens_q), h, d, generator=gen).to(torch.bfloat16)
    k = torch.randn(sum(lens_k), hk, d, generator=gen).to(torch.bfloat16)
    v = torch.randn(sum(lens_k), hk, d, generator=gen).to(torch.bfloat16)
    g = torch.randn(sum(lens_q), h, d, generator=gen).to(torch.bfloat16)
    cq = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32)
    ck = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32)
    ql, kl, vl = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    torch.manual_seed(5)
    out, lse, S = fa.flash_attn_varlen_func(ql, kl, vl, cq.to(DEV), ck.to(DEV), msq, msk, p_drop, causal=causal,
                                            return_attn_probs=True)
    assert S.dtype == torch.bfloat16 and tuple(S.shape) == (b, h, (msq + 127) // 128 * 128, (msk + 127) // 128 * 128)
    dq, dk, dv = torch.autograd.grad(out, (ql, kl, vl), g.to(DEV))
    S = S.cpu()
    for i in range(b):
        sq_, sk_ = lens_q[i], lens_k[i]
        rq, rk = slice(cq[i], cq[i + 1]), slice(ck[i], ck[i + 1])
        S_i = oracle.convert_flash_attn_S_to_softmax(S[i:i + 1], sq_, sk_, None, None, causal=causal)
        keep = S_i >= 0
        k_rep = k[rk][None].repeat_interleave(h // hk, dim=2)
        attn = oracle.normalize_flash_attn_S(S_i.abs(), q[rq][None], k_rep, k_rep, None, None, None, True, causal=causal)
        _, attn_ref = oracle.attention_ref(q[rq][None], k[rk][None], v[rk][None], None, None, causal=causal)
        assert (attn.float() - attn_ref.float()).abs().max().item() <= 2e-2, f"seq {i}: decoded attention"

        def run(**extra):
            q2, k2, v2 = (t[None].clone().requires_grad_(True) for t in (q[rq], k[rk], v[rk]))
            o = oracle.attention_ref(q2, k2, v2, None, None, dropout_p=p_drop, dropout_mask=keep, causal=causal, **extra)[0]
            return (o.detach()[0],) + tuple(t[0] for t in torch.autograd.grad(o, (q2, k2, v2), g[rq][None]))
        ref, pt = run(), run(upcast=False, reorder_ops=True)
        got = (out[rq], dq[rq], dk[rk], dv[rk])
        for name, x, r, p_, mult in zip(("out", "dq", "dk", "dv"), got, ref, pt, (2, 3, 3, 3)):
            err = (x.detach().float().cpu() - r.float()).abs().max().item()
            assert err <= _bound(r, p_, mult), f"seq {i} {name}: {err:.3e} > {_bound(r, p_, mult):.3e}"


def test_dropout_argument_checks():
    fa = _api()
    q, k, v = (torch.randn(1, 64, 2, 64, device=DEV, dtype=torch.float16) for _ in range(3))
    with pytest.raises(RuntimeError, match=r"p_dropout must be in \[0, 1\)"):
        fa.flash_attn_func(q, k, v, 1.0)
    with pytest.raises(RuntimeError, match="Softcapping does not support dropout"):
        fa.flash_attn_func(q, k, v, 0.1, softcap=30.0)
    # return_attn_probs without dropout: the reference returns an empty S_dmask (flash_attn_interface.py:839)
    out, lse, s = fa.flash_attn_func(q, k, v, 0.0, return_attn_probs=True)
    assert s.numel() == 0
