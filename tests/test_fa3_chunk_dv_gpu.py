"""GPU parity tests of the two FA3-only forward arguments of the reference's own test matrix (hopper/test_flash_attn.py:120-131):
`attention_chunk` (hopper/flash_api.cpp:148-160, hopper/mask.h:116-119) and a head dim of V that differs from that of Q / K
(`dv`, hopper/flash_api.cpp:764,782-792).  Contract as in hopper/test_flash_attn.py:193-194,223:
    |out - out_ref|max <= 2 |out_pt - out_ref|max + fwd_atol,  fwd_atol = 2 |(out_ref + 0.3 - 0.3) - out_ref|max."""
import math

import pytest
import torch

from oracle import attention_ref as oracle
from oracle.cases import FA3_CASES, checksum, make_inputs, padding_masks

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fa3():
    from flash_attention_annotated_amd import hopper_interface
    return hopper_interface


def _check(out, out_ref, out_pt, rtol=2):
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    fwd_atol = 2 * (out_ref.float() + 0.3 - 0.3 - out_ref.float()).abs().max().item()
    bound = rtol * (out_pt.float() - out_ref.float()).abs().max().item() + fwd_atol
    assert math.isfinite(err) and err <= bound, f"max err {err:.3e} > bound {bound:.3e}"


def _unpad(x, mask):
    """(b, s, h, d) + bool (b, s) -> packed rows, cu_seqlens, max length (what generate_qkv does, hopper/test_util.py:30-154)"""
    lens = mask.sum(-1).to(torch.int32)
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    return torch.cat([x[i, :int(lens[i])] for i in range(len(lens))]), cu, int(lens.max())


@pytest.mark.parametrize("name", list(FA3_CASES))
def test_fa3_golden_cases(name, golden_fa3):
    """HIP output vs the frozen outputs of the reference's FA3 oracle (dense cases through flash_attn_func, padded ones
    through flash_attn_varlen_func like hopper/test_flash_attn.py:323-420)."""
    fa3 = _fa3()
    c, g = FA3_CASES[name], golden_fa3[name]
    q, k, v = make_inputs(c)
    assert abs(checksum(q) - g["input_checksum"][0].item()) < 1e-6
    qm, km = padding_masks(c)
    kw = dict(causal=c["causal"], window_size=tuple(c["window"]), attention_chunk=c["chunk"])
    st = c["store_row_stride"]
    if qm is None:
        out, lse = fa3.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), return_attn_probs=True, **kw)
        assert tuple(out.shape) == (c["b"], c["sq"], c["h"], c["dv"]) and out.dtype == q.dtype
        lse = lse.cpu()
    else:
        qu, cu_q, mq = _unpad(q, qm)
        ku, cu_k, mk = _unpad(k, km)
        vu, _, _ = _unpad(v, km)
        ou, lse_u = fa3.flash_attn_varlen_func(qu.to(DEV), ku.to(DEV), vu.to(DEV), cu_q.to(DEV), cu_k.to(DEV), mq, mk,
                                               return_attn_probs=True, **kw)
        assert tuple(ou.shape) == (qu.shape[0], c["h"], c["dv"])
        out = torch.zeros(c["b"], c["sq"], c["h"], c["dv"], dtype=q.dtype)
        lse = torch.full((c["b"], c["h"], c["sq"]), float("inf"))
        for i in range(c["b"]):
            n = int(cu_q[i + 1] - cu_q[i])
            out[i, :n] = ou[int(cu_q[i]):int(cu_q[i + 1])].cpu()
            lse[i, :, :n] = lse_u[:, int(cu_q[i]):int(cu_q[i + 1])].cpu()
    _check(out[:, ::st], g["out_ref_fp32"], g["out_pt"])
    fin = torch.isfinite(g["lse"])
    if qm is not None:  # rows behind a sequence's end: the oracle's LSE there belongs to padding rows
        fin = fin & qm[:, None, ::st]
    got = lse[:, :, ::st]
    assert torch.equal(torch.isfinite(got)[fin], torch.ones_like(got[fin], dtype=torch.bool))
    assert (got[fin] - g["lse"][fin]).abs().max().item() < 2e-3
    if qm is None:  # fully masked rows (no key inside chunk and window): out = 0, lse = +inf
        empty = ~torch.isfinite(g["lse"])
        assert torch.equal(~torch.isfinite(got), empty)
        rows = empty.permute(0, 2, 1)  # (b, s, h)
        assert out[:, ::st][rows].abs().max().item() == 0.0 if rows.any() else True


@pytest.mark.parametrize("causal,local", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("mha_type", ["gqa"])
@pytest.mark.parametrize("sq,sk", [(1, 1), (64, 128), (239, 1), (113, 203), (640, 128), (777, 1023)])
@pytest.mark.parametrize("d,dtype", [(64, torch.bfloat16), (128, torch.float16), (192, torch.bfloat16)])
def test_attention_chunk_and_dv_sweep(d, dtype, sq, sk, mha_type, causal, local):
    """The (dv, attention_chunk) loop of hopper/test_flash_attn.py::test_flash_attn_output (:120-200): dv in {128, d} for d in
    (128, 192], {256, 512, d} for d <= 64; attention_chunk random in [1, 2 seqlen_k) and 0."""
    fa3 = _fa3()
    torch.manual_seed(sq * 7 + sk + d)
    b, h = 2, 6
    hk = 6 if mha_type == "mha" else 2
    dv_vals = [128, d] if 128 < d <= 192 else ([256, 512, d] if d <= 64 else [d])
    chunks = [int(torch.randint(1, sk * 2, (1,)).item()), 0]
    window = tuple(int(x) for x in torch.randint(0, sk, (2,)).tolist()) if local else (-1, -1)
    for dv in dv_vals:
        for chunk in chunks:
            if dv == d and chunk == 0:
                continue  # (the plain problem: tests/test_flash_attn_gpu.py)
            q = torch.randn(b, sq, h, d, dtype=dtype)
            k = torch.randn(b, sk, hk, d, dtype=dtype)
            v = torch.randn(b, sk, hk, dv, dtype=dtype)
            kw = dict(causal=causal, window_size=window, attention_chunk=chunk)
            out, lse = fa3.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), return_attn_probs=True, **kw)
            out_ref, _, lse_ref = oracle.attention_ref(q, k, v, **kw, return_lse=True)
            out_pt, _ = oracle.attention_ref(q, k, v, **kw, upcast=False, reorder_ops=True)
            assert tuple(out.shape) == (b, sq, h, dv)
            _check(out, out_ref, out_pt)
            fin = torch.isfinite(lse_ref)
            assert torch.equal(torch.isfinite(lse.cpu()), fin), (dv, chunk)
            if fin.any():
                assert (lse.cpu()[fin] - lse_ref[fin]).abs().max().item() < 2e-3, (dv, chunk)


def test_attention_chunk_long_and_split_free():
    """A chunked problem at a length where the key-block range matters (only the tiles of a block's chunks are swept):
    s = 4224 (hopper/test_flash_attn.py's largest), chunk 1000, causal, GQA; every 7th row against the oracle."""
    fa3 = _fa3()
    torch.manual_seed(5)
    b, s, h, hk, d = 1, 4224, 4, 2, 128
    q, k, v = (torch.randn(b, s, hh, d, dtype=torch.bfloat16) for hh in (h, hk, hk))
    out = fa3.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=True, attention_chunk=1000)
    rows = torch.arange(0, s, 7)
    jj = torch.arange(s).view(1, -1)
    lo = rows.view(-1, 1) - rows.view(-1, 1) % 1000
    bias = torch.where((jj <= rows.view(-1, 1)) & (jj >= lo), 0.0, float("-inf")).view(1, 1, len(rows), s)
    ref, _ = oracle.attention_ref(q[:, rows], k, v, attn_bias=bias)
    pt, _ = oracle.attention_ref(q[:, rows], k, v, attn_bias=bias, upcast=False, reorder_ops=True)
    _check(out[:, rows.to(DEV)], ref, pt)


def test_fp8_with_attention_chunk_takes_the_expansion_path():
    """fp8 inputs + attention_chunk: the native e4m3 kernel has no chunk mask, so the exact e4m3 -> bf16 expansion runs in
    front of the 16-bit kernel; result = the bf16 surface on the same (e4m3-representable) values."""
    fa3 = _fa3()
    FP8 = torch.float8_e4m3fn
    torch.manual_seed(11)
    q, k, v = (torch.randn(2, 200, hh, 128, dtype=torch.bfloat16).to(FP8) for hh in (4, 2, 2))
    o8 = fa3.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), attention_chunk=60)
    o16 = fa3.flash_attn_func(q.to(torch.bfloat16).to(DEV), k.to(torch.bfloat16).to(DEV), v.to(torch.bfloat16).to(DEV),
                              attention_chunk=60)
    assert torch.equal(o8, o16)


def test_rejections_and_backward():
    fa3 = _fa3()
    q = torch.randn(1, 16, 2, 128, dtype=torch.bfloat16, device=DEV)
    v = torch.randn(1, 16, 2, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="If V headdim is different from Q/K dim"):
        fa3.flash_attn_func(q, q, v)                                    # hopper/flash_api.cpp:783-786
    q64 = torch.randn(1, 16, 2, 64, dtype=torch.bfloat16, device=DEV, requires_grad=True)
    v128 = torch.randn(1, 16, 2, 128, dtype=torch.bfloat16, device=DEV)
    out = fa3.flash_attn_func(q64, q64.detach(), v128)
    with pytest.raises(RuntimeError, match="only when the larger of the two is above 128"):
        out.sum().backward()
    qc = torch.randn(1, 16, 2, 64, dtype=torch.bfloat16, device=DEV, requires_grad=True)
    oc = fa3.flash_attn_func(qc, qc.detach(), qc.detach(), attention_chunk=4)
    with pytest.raises(AssertionError, match="attention_chunk"):       # hopper/flash_attn_interface.py: backward has no chunk
        oc.sum().backward()


@pytest.mark.parametrize("varlen", [False, True])
@pytest.mark.parametrize("causal,window", [(False, (-1, -1)), (True, (-1, -1)), (False, (60, 20))])
@pytest.mark.parametrize("d,dv,dtype", [(192, 128, torch.bfloat16), (160, 128, torch.float16), (136, 104, torch.bfloat16), (64, 256, torch.bfloat16)])
def test_backward_with_own_v_head_dim(d, dv, dtype, causal, window, varlen):
    """dq / dk / dv for a V head dim that differs from the Q / K one (hopper/flash_api.cpp:1345-1369; the `dv` axis of the
    backward half of hopper/test_flash_attn.py:225-286), contract |g - g_ref| <= 3 |g_pt - g_ref| + atol (:281-286)."""
    fa3 = _fa3()
    torch.manual_seed(d + dv + int(causal))
    b, sq, sk, h, hk = 2, 200, 328, 4, 2
    q = torch.randn(b, sq, h, d, dtype=dtype)
    k = torch.randn(b, sk, hk, d, dtype=dtype)
    v = torch.randn(b, sk, hk, dv, dtype=dtype)
    g = torch.randn(b, sq, h, dv, dtype=dtype)
    kw = dict(causal=causal, window_size=window)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    if not varlen:
        out = fa3.flash_attn_func(qd, kd, vd, **kw)
        got = torch.autograd.grad(out, (qd, kd, vd), g.to(DEV))
        qm = km = None
    else:
        lens_q, lens_k = torch.tensor([sq, sq - 37]), torch.tensor([sk - 50, sk])
        qm = torch.arange(sq).view(1, -1) < lens_q.view(-1, 1)
        km = torch.arange(sk).view(1, -1) < lens_k.view(-1, 1)
        qu, cu_q, mq = _unpad(qd, qm)
        ku, cu_k, mk = _unpad(kd, km)
        vu, _, _ = _unpad(vd, km)
        gu, _, _ = _unpad(g.to(DEV), qm)
        ou = fa3.flash_attn_varlen_func(qu, ku, vu, cu_q.to(DEV), cu_k.to(DEV), mq, mk, **kw)
        got = torch.autograd.grad(ou, (qd, kd, vd), gu)

    def oracle_grads(**extra):
        ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
        o, _ = oracle.attention_ref(ql, kl, vl, qm, km, **kw, **extra)
        gg = g if qm is None else g.masked_fill(~qm.view(b, sq, 1, 1), 0)
        return torch.autograd.grad(o, (ql, kl, vl), gg)
    ref, pt = oracle_grads(), oracle_grads(upcast=False, reorder_ops=True)
    for name, a, r_, p_ in zip(("dq", "dk", "dv"), got, ref, pt):
        err = (a.float().cpu() - r_.float()).abs().max().item()
        atol = 2 * (r_.float() + 0.3 - 0.3 - r_.float()).abs().max().item()
        bound = 3 * (p_.float() - r_.float()).abs().max().item() + atol
        assert math.isfinite(err) and err <= bound, f"{name}: err {err:.3e} > bound {bound:.3e}"
