"""CPU tests of the oracle (test infrastructure): replay of the reference's frozen outputs, cross-check of the
torch restatement against the independent C fp64 restatement, and the mask diagrams of the reference docstring."""
import ctypes
import os
import subprocess

import pytest
import torch

from oracle import attention_ref as oracle
from oracle.cases import CASES, FA3_CASES, checksum, make_alibi_slopes, make_descales, make_inputs, padding_masks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_reference_golden(name, golden):
    """tests/golden/attention_ref_golden.pt holds the outputs of the reference's own attention_ref
    (tests/test_util.py:185-274) for these seeded inputs; the restatement must reproduce them."""
    c, g = CASES[name], golden[name]
    q, k, v = make_inputs(c)
    for t, want in zip((q, k, v), g["input_checksum"].tolist()):
        assert abs(checksum(t) - want) <= 1e-6 * max(1.0, abs(want)), "seeded inputs differ from the frozen ones"
    qm, km = padding_masks(c)
    kw = dict(causal=c["causal"], window_size=tuple(c["window"]), softcap=c["softcap"])
    st = c["store_row_stride"]
    ptkw = {}
    if c.get("fp8"):
        qd, kd, vd = make_descales(c)
        kw.update(q_descale=qd, k_descale=kd, v_descale=vd)
        ptkw = dict(intermediate_dtype=torch.float8_e4m3fn)
    slopes = make_alibi_slopes(c)
    if slopes is not None:  # bias built exactly as the reference's tests build it (causal shortcut included)
        kw["attn_bias"] = oracle.attn_bias_from_alibi_slopes(slopes, c["sq"], c["sk"], qm, km, causal=c["causal"])
    out32, _ = oracle.attention_ref(q.float(), k.float(), v.float(), qm, km, **kw)
    out_pt, _ = oracle.attention_ref(q, k, v, qm, km, **kw, **ptkw, upcast=False, reorder_ops=True)
    # fp32 math: same op order as the reference -> agreement to rounding (bit-identical in the build container)
    assert (out32[:, ::st] - g["out_ref_fp32"]).abs().max().item() <= (5e-6 if c.get("fp8") else 2e-6)
    tol_pt = 0.0 if q.dtype == torch.float32 else 2e-2  # low-precision path: allow one ulp-level flip across hosts
    assert (out_pt[:, ::st].float() - g["out_pt"].float()).abs().max().item() <= tol_pt + 1e-6
    _, _, lse = oracle.attention_ref(q, k, v, qm, km, **kw, return_lse=True)
    fin = torch.isfinite(g["lse"])
    assert torch.equal(torch.isfinite(lse[:, :, ::st]), fin)
    if fin.any():
        assert (lse[:, :, ::st][fin] - g["lse"][fin]).abs().max().item() <= 1e-5


@pytest.mark.parametrize("name", list(FA3_CASES))
def test_oracle_reproduces_fa3_golden(name, golden_fa3):
    """tests/golden/attention_fa3_golden.pt: the reference's FA3 oracle (hopper/test_util.py:226-348) on the attention_chunk /
    head-dim-of-V cases (the `dv` and `attention_chunk` axes of hopper/test_flash_attn.py:120-131)."""
    c, g = FA3_CASES[name], golden_fa3[name]
    q, k, v = make_inputs(c)
    assert v.shape[-1] == c["dv"]
    for t, want in zip((q, k, v), g["input_checksum"].tolist()):
        assert abs(checksum(t) - want) <= 1e-6 * max(1.0, abs(want)), "seeded inputs differ from the frozen ones"
    qm, km = padding_masks(c)
    kw = dict(causal=c["causal"], window_size=tuple(c["window"]), softcap=c["softcap"], attention_chunk=c["chunk"])
    st = c["store_row_stride"]
    out32, _ = oracle.attention_ref(q.float(), k.float(), v.float(), qm, km, **kw)
    out_pt, _ = oracle.attention_ref(q, k, v, qm, km, **kw, upcast=False, reorder_ops=True)
    assert out32.shape[-1] == c["dv"]
    assert (out32[:, ::st] - g["out_ref_fp32"]).abs().max().item() <= 5e-6
    assert (out_pt[:, ::st].float() - g["out_pt"].float()).abs().max().item() <= 2e-2 + 1e-6
    _, _, lse = oracle.attention_ref(q, k, v, qm, km, **kw, return_lse=True)
    fin = torch.isfinite(g["lse"])
    assert torch.equal(torch.isfinite(lse[:, :, ::st]), fin)
    if fin.any():
        assert (lse[:, :, ::st][fin] - g["lse"][fin]).abs().max().item() <= 1e-5


def test_chunk_mask_by_definition():
    """construct_chunk_mask (hopper/test_util.py:193-223) against the definition spelled out per element, incl. rows whose
    diagonal position is negative (seqlen_q > seqlen_k: Python floor remainder -> the chunk ends at or before key 0)."""
    for sq, sk, chunk in ((7, 10, 3), (10, 4, 4), (5, 5, 1), (6, 9, 20)):
        m = oracle.chunk_mask(sq, sk, chunk)
        for i in range(sq):
            diag = i + sk - sq
            lo = (diag // chunk) * chunk
            for j in range(sk):
                assert bool(m[i, j]) == (not (lo <= j < lo + chunk)), (sq, sk, chunk, i, j)


def _c_oracle():
    so = os.path.join(ROOT, "oracle", "_ref", "liboracle_attn.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(so)
    f = lib.fa_oracle_attention_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_float] + [ctypes.c_int] * 3 + [ctypes.c_float]
    return f


@pytest.mark.parametrize("causal,window,softcap,sq,sk,h,hk,d", [
    (False, (-1, -1), 0.0, 64, 64, 2, 2, 32),
    (True, (-1, -1), 0.0, 37, 91, 4, 2, 64),
    (True, (-1, -1), 0.0, 91, 37, 4, 1, 64),
    (False, (7, 3), 0.0, 50, 70, 2, 2, 16),
    (False, (0, 0), 0.0, 33, 33, 1, 1, 8),
    (False, (-1, -1), 15.0, 40, 48, 2, 1, 32),
])
def test_c_oracle_agrees_with_torch_oracle(causal, window, softcap, sq, sk, h, hk, d):
    """Two independent restatements (torch fp32, plain C fp64) of the same definition agree."""
    f = _c_oracle()
    torch.manual_seed(sq * 1000 + sk)
    q = torch.randn(2, sq, h, d) * (4.0 if softcap else 1.0)
    k = torch.randn(2, sk, hk, d)
    v = torch.randn(2, sk, hk, d)
    out = torch.empty_like(q)
    lse = torch.empty(2, h, sq)
    st = f(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), 2, sq, sk, h, hk, d,
           d ** -0.5, int(causal), window[0], window[1], softcap)
    assert st == 0
    ref, _, lse_ref = oracle.attention_ref(q, k, v, causal=causal, window_size=window, softcap=softcap, return_lse=True)
    assert (out - ref).abs().max().item() <= 1e-5
    fin = torch.isfinite(lse_ref)
    assert torch.equal(torch.isfinite(lse), fin)
    assert (lse[fin] - lse_ref[fin]).abs().max().item() <= 2e-5


def test_docstring_causal_masks():
    """flash_attn/flash_attn_interface.py:1164-1174: causal masks are aligned to the bottom-right corner;
    (sq, sk) = (2, 5) keeps [[1,1,1,1,0],[1,1,1,1,1]]; (5, 2) keeps [[0,0],[0,0],[0,0],[1,0],[1,1]]."""
    m = ~oracle.local_mask(2, 5, (-1, 0))
    assert m.int().tolist() == [[1, 1, 1, 1, 0], [1, 1, 1, 1, 1]]
    m = ~oracle.local_mask(5, 2, (-1, 0))
    assert m.int().tolist() == [[0, 0], [0, 0], [0, 0], [1, 0], [1, 1]]
    q = torch.randn(1, 5, 1, 8)
    k = torch.randn(1, 2, 1, 8)
    v = torch.randn(1, 2, 1, 8)
    out, attn, lse = oracle.attention_ref(q, k, v, causal=True, return_lse=True)
    assert torch.all(out[0, :3] == 0), "rows with no visible key produce zeros"
    assert torch.all(torch.isposinf(lse[0, 0, :3]))
    assert torch.allclose(out[0, 3, 0], v[0, 0, 0], atol=1e-6)


def test_varlen_oracle_matches_padded_oracle():
    torch.manual_seed(0)
    b, sq, sk, h, hk, d = 3, 17, 29, 4, 2, 16
    q = torch.randn(b, sq, h, d)
    k = torch.randn(b, sk, hk, d)
    v = torch.randn(b, sk, hk, d)
    lens_q, lens_k = [17, 5, 11], [29, 13, 1]
    qm = torch.arange(sq).view(1, -1) < torch.tensor(lens_q).view(-1, 1)
    km = torch.arange(sk).view(1, -1) < torch.tensor(lens_k).view(-1, 1)
    ref, _ = oracle.attention_ref(q, k, v, qm, km, causal=True)
    cq = torch.tensor([0, 17, 22, 33], dtype=torch.int32)
    ck = torch.tensor([0, 29, 42, 43], dtype=torch.int32)
    qu = torch.cat([q[i, :lens_q[i]] for i in range(b)])
    ku = torch.cat([k[i, :lens_k[i]] for i in range(b)])
    vu = torch.cat([v[i, :lens_k[i]] for i in range(b)])
    out_u, lse_u = oracle.attention_varlen_ref(qu, ku, vu, cq, ck, causal=True)
    for i in range(b):
        assert (out_u[cq[i]:cq[i + 1]] - ref[i, :lens_q[i]]).abs().max().item() <= 1e-6
    assert lse_u.shape == (h, 33)


@pytest.mark.parametrize("name", list(__import__("oracle.cases", fromlist=["GRAD_CASES"]).GRAD_CASES))
def test_oracle_autograd_reproduces_reference_gradients(name, golden_grads):
    """tests/golden/attention_grad_golden.pt: dq/dk/dv obtained by differentiating the reference's own attention_ref
    (tests/test_flash_attn.py:1071-1105); autograd through the restatement must reproduce them."""
    from oracle.cases import GRAD_CASES, make_grad_output
    c, gold = GRAD_CASES[name], golden_grads[name]
    q, k, v = make_inputs(c)
    g = make_grad_output(c)
    for t, want in zip((q, k, v, g), gold["input_checksum"].tolist()):
        assert abs(checksum(t) - want) <= 1e-6 * max(1.0, abs(want))
    qm, km = padding_masks(c)
    kw = dict(causal=c["causal"], window_size=tuple(c["window"]), softcap=c["softcap"])
    slopes = make_alibi_slopes(c)
    if slopes is not None:
        kw["attn_bias"] = oracle.attn_bias_from_alibi_slopes(slopes, c["sq"], c["sk"], qm, km, causal=c["causal"])
    for tag, extra in (("ref", {}), ("pt", dict(upcast=False, reorder_ops=True))):
        ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = oracle.attention_ref(ql, kl, vl, qm, km, **kw, **extra)[0]
        grads = torch.autograd.grad(out, (ql, kl, vl), g)
        for nm, got in zip(("dq", "dk", "dv"), grads):
            want = gold[f"{nm}_{tag}"].float()
            tol = 2.0 ** -6 * max(1.0, want.abs().max().item())  # a few 16-bit ulps across hosts / the FA3-oracle case
            assert (got.float() - want).abs().max().item() <= tol, (tag, nm)


@pytest.mark.parametrize("causal", [False, True])
def test_oracle_dropout_reproduces_reference(causal, golden_grads):
    """Dropout branch of the oracle (tests/test_util.py:262-269) against a fixture made by the reference's own
    attention_ref with the same keep-mask (oracle/make_golden.py): output and gradients, fp32 and 16-bit orders."""
    gold = golden_grads[f"dropout_pin_causal{int(causal)}"]
    for tag, extra in (("ref", {}), ("pt", dict(upcast=False, reorder_ops=True))):
        ql, kl, vl = (gold[n].clone().requires_grad_(True) for n in ("q", "k", "v"))
        out = oracle.attention_ref(ql, kl, vl, None, None, dropout_p=gold["p_dropout"], dropout_mask=gold["keep"],
                                   causal=causal, **extra)[0]
        grads = torch.autograd.grad(out, (ql, kl, vl), gold["g"])
        for nm, got in zip(("out", "dq", "dk", "dv"), (out.detach(),) + grads):
            want = gold[f"{nm}_{tag}"].float()
            tol = 2.0 ** -6 * max(1.0, want.abs().max().item())
            assert (got.float() - want).abs().max().item() <= tol, (tag, nm)
    # dropping nothing is the plain softmax; the mask really is applied
    q, k, v = gold["q"], gold["k"], gold["v"]
    plain = oracle.attention_ref(q, k, v, causal=causal)[0]
    all_kept = oracle.attention_ref(q, k, v, dropout_p=0.0, dropout_mask=torch.ones_like(gold["keep"]), causal=causal)[0]
    assert torch.equal(plain, all_kept)
    assert not torch.equal(plain, gold["out_ref"])


def test_oracle_combine_reproduces_reference(golden_grads):
    """Split-KV merge restatement against the fixture made by the reference's attention_combine_ref
    (hopper/test_flash_attn.py:1105-1114), -inf splits and an all -inf row included."""
    gold = golden_grads["combine_pin"]
    out, lse = oracle.attention_combine_ref(gold["out_partial"], gold["lse_partial"])
    assert torch.allclose(out, gold["out"], atol=1e-6, rtol=1e-6)
    assert torch.equal(torch.isinf(lse), torch.isinf(gold["lse"]))
    fin = ~torch.isinf(lse)
    assert torch.allclose(lse[fin], gold["lse"][fin], atol=1e-6, rtol=1e-6)
    assert (out[2, 3] == 0).all() and torch.isinf(lse[2, 3]).all()


def test_row_subset_oracle_equals_full_oracle():
    """tests/test_full_size_gpu.py evaluates the oracle for a sample of query rows with the causal mask passed as an
    additive -inf bias.  That is bit-identical to the oracle's own causal path on those rows (out, and LSE), in the
    fp32 and in the low-precision reordered flavour, for sq == sk and sq != sk."""
    from oracle import attention_ref as oracle
    import torch
    torch.manual_seed(3)
    for sq, sk in ((96, 96), (64, 131)):
        q = torch.randn(2, sq, 4, 32, dtype=torch.bfloat16)
        k = torch.randn(2, sk, 2, 32, dtype=torch.bfloat16)
        v = torch.randn(2, sk, 2, 32, dtype=torch.bfloat16)
        rows = [0, 1, 17, 40, sq - 1]
        i = torch.tensor(rows).view(-1, 1)
        j = torch.arange(sk).view(1, -1)
        bias = torch.where(j <= i + sk - sq, 0.0, float("-inf")).view(1, 1, len(rows), sk)
        for kw in (dict(), dict(upcast=False, reorder_ops=True)):
            full, _, lse_full = oracle.attention_ref(q, k, v, causal=True, return_lse=True, **kw)
            sub, _, lse_sub = oracle.attention_ref(q[:, rows], k, v, attn_bias=bias, return_lse=True, **kw)
            assert torch.equal(sub, full[:, rows])
            assert torch.equal(lse_sub, lse_full[:, :, rows])
