"""GPU parity tests of the FA3-shaped surface: fp8 e4m3 inputs with per-(batch, kv head) descales (BASELINE config 5)
and the FA3 extras (seqused_q/k), with the reference's FA3 tolerance contract:
    |out - out_ref|max <= rtol * |out_pt - out_ref|max + fwd_atol,  rtol = 2,
    fwd_atol = 2 * |(out_ref + 0.3 - 0.3) - out_ref|max             (hopper/test_flash_attn.py:193-194, 223)
where out_pt is the same math in bf16 with P rounded through e4m3 (:180)."""
import math

import pytest
import torch

from oracle import attention_ref as oracle
from oracle.cases import CASES, checksum, make_descales, make_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP8 = torch.float8_e4m3fn


def _fa3():
    from flash_attention_annotated_amd import hopper_interface
    return hopper_interface


def _check(out, out_ref, out_pt, rtol=2):
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    fwd_atol = 2 * (out_ref.float() + 0.3 - 0.3 - out_ref.float()).abs().max().item()
    bound = rtol * (out_pt.float() - out_ref.float()).abs().max().item() + fwd_atol
    assert math.isfinite(err) and err <= bound, f"max err {err:.3e} > bound {bound:.3e}"


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c.get("fp8")])
def test_fp8_golden_cases(name, golden):
    """HIP output for fp8 inputs vs the reference FA3 oracle's frozen outputs."""
    fa3 = _fa3()
    c, g = CASES[name], golden[name]
    q, k, v = make_inputs(c)
    assert abs(checksum(q) - g["input_checksum"][0].item()) < 1e-6
    qd, kd, vd = make_descales(c)
    out, lse = fa3.flash_attn_func(q.to(FP8).to(DEV), k.to(FP8).to(DEV), v.to(FP8).to(DEV), causal=c["causal"],
                                   q_descale=qd.to(DEV), k_descale=kd.to(DEV), v_descale=vd.to(DEV),
                                   return_attn_probs=True)
    assert out.dtype == torch.bfloat16  # hopper/flash_api.cpp:859
    st = c["store_row_stride"]
    _check(out[:, ::st], g["out_ref_fp32"], g["out_pt"])
    fin = torch.isfinite(g["lse"])
    assert (lse.cpu()[:, :, ::st][fin] - g["lse"][fin]).abs().max().item() < 5e-3


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("mha_type", ["mha", "gqa", "mqa"])
@pytest.mark.parametrize("sq,sk,d", [(113, 203, 128), (512, 512, 128), (384, 1024, 64), (1024, 1024, 128)])
def test_fp8_output(sq, sk, d, mha_type, causal):
    """hopper/test_flash_attn.py::test_flash_attn_output, fp8 branch (:135-194)."""
    fa3 = _fa3()
    torch.manual_seed(0)
    b, h = 3, 6
    hk = {"mha": 6, "gqa": 2, "mqa": 1}[mha_type]
    q_ref = torch.randn(b, sq, h, d, dtype=torch.bfloat16).to(FP8).to(torch.bfloat16)
    k_ref = torch.randn(b, sk, hk, d, dtype=torch.bfloat16).to(FP8).to(torch.bfloat16)
    v_ref = torch.randn(b, sk, hk, d, dtype=torch.bfloat16).to(FP8).to(torch.bfloat16)
    qd, kd, vd = [torch.rand(b, hk, dtype=torch.float32) * 2 for _ in range(3)]
    out = fa3.flash_attn_func(q_ref.to(FP8).to(DEV), k_ref.to(FP8).to(DEV), v_ref.to(FP8).to(DEV), causal=causal,
                              q_descale=qd.to(DEV), k_descale=kd.to(DEV), v_descale=vd.to(DEV))
    kw = dict(causal=causal, q_descale=qd, k_descale=kd, v_descale=vd)
    out_ref, _ = oracle.attention_ref(q_ref, k_ref, v_ref, **kw)
    out_pt, _ = oracle.attention_ref(q_ref, k_ref, v_ref, **kw, upcast=False, reorder_ops=True, intermediate_dtype=FP8)
    _check(out, out_ref, out_pt)


def test_fp8_expansion_path_equals_bf16_path():
    """The fp8 shapes the native kernel does not take (here: forced by an explicit kernel variant) run on the exact
    e4m3 -> bf16 expansion in front of the 16-bit kernel: without descales that equals the bf16 kernel run on the expanded
    values, bit for bit."""
    from flash_attention_annotated_amd import _lib
    fa3 = _fa3()
    lib = _lib.load()
    torch.manual_seed(1)
    q = torch.randn(2, 300, 4, 128, dtype=torch.bfloat16).to(FP8)
    k = torch.randn(2, 333, 2, 128, dtype=torch.bfloat16).to(FP8)
    v = torch.randn(2, 333, 2, 128, dtype=torch.bfloat16).to(FP8)
    try:
        lib.fa_set_default_variant(3)
        o8 = fa3.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=True)
        o16 = fa3.flash_attn_func(q.to(torch.bfloat16).to(DEV), k.to(torch.bfloat16).to(DEV), v.to(torch.bfloat16).to(DEV), causal=True)
    finally:
        lib.fa_set_default_variant(0)
    assert torch.equal(o8, o16)


def test_fp8_native_close_to_expansion_path():
    """Native e4m3 MFMA path (P rounded to e4m3 with the 2^OFF offset) against the exact-expansion path of the same inputs:
    the difference is the e4m3 rounding of P only -- well inside the FA3 tolerance (the reference's own yardstick rounds P
    through e4m3 WITHOUT an offset, hopper/test_flash_attn.py:180) -- and the LSE is the same up to fp32 rounding."""
    from flash_attention_annotated_amd import _lib
    fa3 = _fa3()
    lib = _lib.load()
    torch.manual_seed(5)
    for sq, sk, causal in ((512, 512, False), (700, 1500, True), (2048, 2048, True)):
        q = torch.randn(2, sq, 4, 128, dtype=torch.bfloat16).to(FP8).to(DEV)
        k = torch.randn(2, sk, 2, 128, dtype=torch.bfloat16).to(FP8).to(DEV)
        v = torch.randn(2, sk, 2, 128, dtype=torch.bfloat16).to(FP8).to(DEV)
        on, ln = fa3.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
        try:
            lib.fa_set_default_variant(3)
            oe, le = fa3.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
        finally:
            lib.fa_set_default_variant(0)
        assert (ln - le).abs().max().item() < 1e-3
        d = (on.float() - oe.float()).abs()
        assert d.max().item() < 0.15 and d.mean().item() < 5e-3  # (e4m3 keeps 3 mantissa bits of each probability)


def test_fp8_varlen_with_seqused():
    """Ragged fp8 batch with "unused" tail tokens via seqused_q/k (hopper/test_flash_attn.py:388-433)."""
    fa3 = _fa3()
    torch.manual_seed(2)
    lens_q, lens_k = [70, 128, 33], [90, 200, 64]
    used_q, used_k = [64, 128, 20], [80, 150, 64]
    h, hk, d = 4, 2, 128
    cuq = torch.tensor([0, 70, 198, 231], dtype=torch.int32)
    cuk = torch.tensor([0, 90, 290, 354], dtype=torch.int32)
    q = torch.randn(231, h, d, dtype=torch.bfloat16).to(FP8).to(torch.bfloat16)
    k = torch.randn(354, hk, d, dtype=torch.bfloat16).to(FP8).to(torch.bfloat16)
    v = torch.randn(354, hk, d, dtype=torch.bfloat16).to(FP8).to(torch.bfloat16)
    qd, kd, vd = [torch.rand(3, hk, dtype=torch.float32) * 2 for _ in range(3)]
    out = fa3.flash_attn_varlen_func(q.to(FP8).to(DEV), k.to(FP8).to(DEV), v.to(FP8).to(DEV), cuq.to(DEV), cuk.to(DEV),
                                     max(lens_q), max(lens_k), seqused_q=torch.tensor(used_q, dtype=torch.int32, device=DEV),
                                     seqused_k=torch.tensor(used_k, dtype=torch.int32, device=DEV), causal=True,
                                     q_descale=qd.to(DEV), k_descale=kd.to(DEV), v_descale=vd.to(DEV))
    for i in range(3):
        qs = q[cuq[i]:cuq[i] + used_q[i]][None]
        ks = k[cuk[i]:cuk[i] + used_k[i]][None]
        vs = v[cuk[i]:cuk[i] + used_k[i]][None]
        kw = dict(causal=True, q_descale=qd[i:i + 1], k_descale=kd[i:i + 1], v_descale=vd[i:i + 1])
        ref, _ = oracle.attention_ref(qs, ks, vs, **kw)
        pt, _ = oracle.attention_ref(qs, ks, vs, **kw, upcast=False, reorder_ops=True, intermediate_dtype=FP8)
        _check(out[cuq[i]:cuq[i] + used_q[i]][None], ref, pt)


def test_fa3_bf16_matches_fa2_surface():
    """Same kernel behind both surfaces."""
    import flash_attention_annotated_amd as fa2
    fa3 = _fa3()
    torch.manual_seed(3)
    q = torch.randn(2, 257, 4, 64, dtype=torch.bfloat16, device=DEV)
    k = torch.randn(2, 300, 2, 64, dtype=torch.bfloat16, device=DEV)
    v = torch.randn(2, 300, 2, 64, dtype=torch.bfloat16, device=DEV)
    o3, lse3 = fa3.flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
    o2, lse2, _ = fa2.flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
    assert torch.equal(o3, o2) and torch.equal(lse3, lse2)


def test_fa3_rejections():
    fa3 = _fa3()
    q = torch.randn(1, 16, 2, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="does not support qv"):
        fa3.flash_attn_func(q, q, q, qv=q)
    q8 = torch.randn(1, 16, 2, 72, dtype=torch.bfloat16, device=DEV).to(FP8)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        fa3.flash_attn_func(q8, q8, q8)


def test_flash_attn_3_ops_fwd_and_bwd():
    """torch.ops.flash_attn_3.fwd / .bwd with keyword defaults (the schema of hopper/flash_api.cpp:1672-1731)."""
    import flash_attention_annotated_amd.hopper_interface  # noqa: F401
    from oracle import attention_ref as oracle
    torch.manual_seed(21)
    q = torch.randn(2, 150, 4, 64, dtype=torch.bfloat16)
    k = torch.randn(2, 180, 2, 64, dtype=torch.bfloat16)
    v = torch.randn(2, 180, 2, 64, dtype=torch.bfloat16)
    g = torch.randn(2, 150, 4, 64, dtype=torch.bfloat16)
    out, lse, acc, lse_acc = torch.ops.flash_attn_3.fwd(q.cuda(), k.cuda(), v.cuda(), is_causal=True)
    assert acc.numel() == 0 and lse_acc.numel() == 0
    out_ref, _ = oracle.attention_ref(q, k, v, causal=True)
    out_pt, _ = oracle.attention_ref(q, k, v, causal=True, upcast=False, reorder_ops=True)
    assert (out.float().cpu() - out_ref.float()).abs().max().item() <= 2 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    dq, dk, dv, sd, *_ = torch.ops.flash_attn_3.bwd(g.cuda(), q.cuda(), k.cuda(), v.cuda(), out, lse, is_causal=True)
    ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = torch.autograd.grad(oracle.attention_ref(ql, kl, vl, causal=True)[0], (ql, kl, vl), g)
    ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
    pt = torch.autograd.grad(oracle.attention_ref(ql, kl, vl, causal=True, upcast=False, reorder_ops=True)[0], (ql, kl, vl), g)
    for got, r, p_ in zip((dq, dk, dv), ref, pt):
        assert (got.float().cpu() - r.float()).abs().max().item() <= 3 * (p_.float() - r.float()).abs().max().item() + 1e-4
    meta = torch.ops.flash_attn_3.get_scheduler_metadata(2, 150, 180, 4, 2, 64, 64, torch.bfloat16,
                                                          torch.full((2,), 180, dtype=torch.int32, device="cuda"))
    assert meta.dtype == torch.int32


def test_fa3_interface_autograd_matches_fa2_surface():
    """hopper_interface.flash_attn_func / flash_attn_varlen_func / flash_attn_qkvpacked_func are differentiable
    (hopper/flash_attn_interface.py:157-442); same kernels as the FA2 surface, so gradients are bit-identical to it
    (whose parity against the oracle is tests/test_flash_attn_bwd_gpu.py)."""
    import flash_attention_annotated_amd as fa2
    fa3 = _fa3()
    torch.manual_seed(5)
    b, s, h, hk, d = 2, 200, 4, 2, 64
    q = torch.randn(b, s, h, d, dtype=torch.bfloat16, device=DEV, requires_grad=True)
    k = torch.randn(b, s, hk, d, dtype=torch.bfloat16, device=DEV, requires_grad=True)
    v = torch.randn(b, s, hk, d, dtype=torch.bfloat16, device=DEV, requires_grad=True)
    g = torch.randn(b, s, h, d, dtype=torch.bfloat16, device=DEV)
    for kw in (dict(causal=True), dict(window_size=(30, 10)), dict(causal=True, softcap=20.0)):
        o3, lse3 = fa3.flash_attn_func(q, k, v, return_attn_probs=True, **kw)
        o2 = fa2.flash_attn_func(q, k, v, **kw)
        assert torch.equal(o3, o2) and not lse3.requires_grad
        for a, b_ in zip(torch.autograd.grad(o3, (q, k, v), g), torch.autograd.grad(o2, (q, k, v), g)):
            assert torch.equal(a, b_)
    # varlen
    lens = [200, 57]
    cu = torch.tensor([0, 200, 257], dtype=torch.int32, device=DEV)
    qu = torch.randn(257, h, d, dtype=torch.float16, device=DEV, requires_grad=True)
    ku = torch.randn(257, hk, d, dtype=torch.float16, device=DEV, requires_grad=True)
    vu = torch.randn(257, hk, d, dtype=torch.float16, device=DEV, requires_grad=True)
    gu = torch.randn(257, h, d, dtype=torch.float16, device=DEV)
    o3 = fa3.flash_attn_varlen_func(qu, ku, vu, cu, cu, max(lens), max(lens), causal=True)
    o2 = fa2.flash_attn_varlen_func(qu, ku, vu, cu, cu, max(lens), max(lens), causal=True)
    assert torch.equal(o3, o2)
    for a, b_ in zip(torch.autograd.grad(o3, (qu, ku, vu), gu), torch.autograd.grad(o2, (qu, ku, vu), gu)):
        assert torch.equal(a, b_)
    # qkv packed: (b, s, 3, h, d) and the GQA form (b, s, h + 2 h_k, d)
    qkv = torch.randn(b, s, 3, h, d, dtype=torch.bfloat16, device=DEV, requires_grad=True)
    o3 = fa3.flash_attn_qkvpacked_func(qkv, causal=True)
    (d3,) = torch.autograd.grad(o3, qkv, g)
    o2 = fa2.flash_attn_qkvpacked_func(qkv, causal=True)
    (d2,) = torch.autograd.grad(o2, qkv, g)
    assert torch.equal(o3, o2) and torch.equal(d3, d2)
    packed = torch.cat([q, k, v], dim=2).detach().requires_grad_(True)  # (b, s, h + 2 hk, d)
    o3 = fa3.flash_attn_qkvpacked_func(packed, causal=True, num_heads_q=h)
    (dp,) = torch.autograd.grad(o3, packed, g)
    o2 = fa2.flash_attn_func(q, k, v, causal=True)
    want = torch.cat(torch.autograd.grad(o2, (q, k, v), g), dim=2)
    assert torch.equal(o3, o2) and torch.equal(dp, want)
    meta = fa3.get_scheduler_metadata(b, 1, 4096, h, hk, d, torch.full((b,), 100, dtype=torch.int32, device=DEV))
    assert meta.dtype == torch.int32


@pytest.mark.parametrize("sq,sk,window", [(300, 120, (50, -1)), (300, 120, (-1, 40)), (257, 515, (64, -1))])
def test_fa3_one_sided_window(sq, sk, window):
    """FA3 rule for a one-sided sliding window (hopper/flash_api.cpp:152-153, 589-590): the missing side is unbounded -- also
    when seqlen_q > seqlen_k, where the FA2 rule (the other side becomes seqlen_k) would mask whole rows.  Forward and
    backward; the oracle gets the equivalent two-sided window."""
    fa3 = _fa3()
    torch.manual_seed(4)
    b, h, hk, d = 2, 4, 2, 64
    q = torch.randn(b, sq, h, d, dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    g = torch.randn(b, sq, h, d, dtype=torch.bfloat16)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out = fa3.flash_attn_func(qd, kd, vd, window_size=window)
    grads = torch.autograd.grad(out, (qd, kd, vd), g.to(DEV))
    big = sq + sk
    w2 = (window[0] if window[0] >= 0 else big, window[1] if window[1] >= 0 else big)

    def ref(**kw):
        ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
        o, _ = oracle.attention_ref(ql, kl, vl, window_size=w2, **kw)
        return o, torch.autograd.grad(o, (ql, kl, vl), g)
    out_ref, g_ref = ref()
    out_pt, g_pt = ref(upcast=False, reorder_ops=True)
    assert torch.isfinite(out.float()).all()
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    assert err <= 2 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5, err
    for name, got, r, pt in zip(("dq", "dk", "dv"), grads, g_ref, g_pt):
        e = (got.float().cpu() - r.float()).abs().max().item()
        assert e <= 3 * (pt.float() - r.float()).abs().max().item() + 1e-4, (name, e)
