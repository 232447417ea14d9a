"""Every BASELINE.json configuration at its FULL stated size (C2 b4 h16 s8192, C3 b4 h16 s16384 causal, C4 varlen GQA
hq32/hkv8 lens 8192..1024, C5 fp8 e4m3 s8192 with random descales; SURVEY.md 8(d) recipes, seed 0) through the C-ABI, against the
oracle evaluated for a SAMPLE of query rows: >= 128 rows per batch entry, at least one in every 256-row m-block (random
offsets inside the block, so every wave and both q-blocks of a wave are hit), ALL heads, all keys.  The code that only
switches on at scale runs here: the generated asm loop over > 100 tiles, unit_tiles scheduling, successor-Q prefetch, 64
causal m-blocks heaviest-first, the ragged scheduler.

Bounds = the reference's inequalities: FA2 `|out - out_ref|max <= 2 |out_pt - out_ref|max + 1e-5`
(tests/test_flash_attn.py:1121,1440,1556); FA3/fp8 `<= 2 |out_pt - out_ref|max + 2 |(out_ref + 0.3 - 0.3) - out_ref|max` with
out_pt rounded through e4m3 (hopper/test_flash_attn.py:180,193-194,223).  LSE (never compared by the reference): 2e-3.

The causal / varlen oracle for a row subset is the full oracle with the mask passed as an additive -inf bias
(`tests/test_oracle.py::test_row_subset_oracle_equals_full_oracle` pins that this is bit-identical to the causal path)."""
import math

import pytest
import torch

from oracle import attention_ref as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP8 = torch.float8_e4m3fn


def sample_rows(sq, n=128, block=256, seed=0):
    """Sorted row indices: ceil(n / #blocks) random rows in every `block`-row m-block (>= n rows in total)."""
    g = torch.Generator().manual_seed(seed)
    nblocks = (sq + block - 1) // block
    per = max(1, -(-n // nblocks))
    rows = []
    for mb in range(nblocks):
        lo, hi = mb * block, min(sq, (mb + 1) * block)
        rows += (lo + torch.randperm(hi - lo, generator=g)[:per]).tolist()
    return sorted(set(rows))


def causal_bias(rows, sq, sk):
    """(1, 1, len(rows), sk): 0 where key j <= i + sk - sq (bottom-right aligned causal), -inf elsewhere."""
    i = torch.tensor(rows, dtype=torch.long).view(-1, 1)
    j = torch.arange(sk, dtype=torch.long).view(1, -1)
    return torch.where(j <= i + sk - sq, 0.0, float("-inf")).view(1, 1, len(rows), sk)


def _check_rows(out_rows, lse_rows, q_rows, k, v, bias, what, fp8_kw=None, lse_tol=2e-3):
    kw = dict(attn_bias=bias)
    if fp8_kw:
        kw.update(fp8_kw)
    out_ref, _, lse_ref = oracle.attention_ref(q_rows, k, v, return_lse=True, **kw)
    if fp8_kw:
        out_pt, _ = oracle.attention_ref(q_rows, k, v, upcast=False, reorder_ops=True, intermediate_dtype=FP8, **kw)
        atol = 2 * (out_ref.float() + 0.3 - 0.3 - out_ref.float()).abs().max().item()
    else:
        out_pt, _ = oracle.attention_ref(q_rows, k, v, upcast=False, reorder_ops=True, **kw)
        atol = 1e-5
    err = (out_rows.float().cpu() - out_ref.float()).abs().max().item()
    bound = 2 * (out_pt.float() - out_ref.float()).abs().max().item() + atol
    assert math.isfinite(err) and err <= bound, f"{what}: max err {err:.3e} > bound {bound:.3e}"
    fin = torch.isfinite(lse_ref)
    lse_rows = lse_rows.float().cpu()
    assert torch.equal(torch.isfinite(lse_rows), fin), f"{what}: lse inf pattern"
    lerr = (lse_rows[fin] - lse_ref[fin]).abs().max().item()
    assert lerr <= lse_tol, f"{what}: lse err {lerr:.3e}"


def _api():
    import flash_attention_annotated_amd as fa
    return fa


@pytest.mark.parametrize("name,s,causal", [("C2", 8192, False), ("C3", 16384, True)])
def test_dense_full_size(name, s, causal):
    """C2 / C3 exactly as SURVEY.md 8(d): seed 0, randn(4, s, 16, 128) bf16 x3 in q, k, v order."""
    fa = _api()
    torch.manual_seed(0)
    b, h, d = 4, 16, 128
    q = torch.randn(b, s, h, d, dtype=torch.bfloat16)
    k = torch.randn(b, s, h, d, dtype=torch.bfloat16)
    v = torch.randn(b, s, h, d, dtype=torch.bfloat16)
    out, lse, _ = fa.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), causal=causal, return_attn_probs=True)
    assert torch.isfinite(out.float()).all()
    out, lse = out.cpu(), lse.cpu()
    for bi in range(b):
        rows = sample_rows(s, seed=bi)
        assert len(rows) >= 128 and len({r // 256 for r in rows}) == s // 256
        bias = causal_bias(rows, s, s) if causal else None
        _check_rows(out[bi:bi + 1, rows], lse[bi:bi + 1, :, rows], q[bi:bi + 1, rows], k[bi:bi + 1], v[bi:bi + 1], bias,
                    f"{name} batch {bi}")


@pytest.mark.parametrize("causal", [False, True])
def test_c4_varlen_gqa_full_size(causal):
    """C4: hq 32 / hkv 8, d 128, lens 8192 .. 1024 packed (cu_seqlens), seed 0."""
    fa = _api()
    torch.manual_seed(0)
    lens = [8192, 7168, 6144, 5120, 4096, 3072, 2048, 1024]
    hq, hk, d = 32, 8, 128
    total = sum(lens)
    q = torch.randn(total, hq, d, dtype=torch.bfloat16)
    k = torch.randn(total, hk, d, dtype=torch.bfloat16)
    v = torch.randn(total, hk, d, dtype=torch.bfloat16)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    out, lse, _ = fa.flash_attn_varlen_func(q.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV), cu.to(DEV), max(lens), max(lens),
                                            causal=causal, return_attn_probs=True)
    out, lse = out.cpu(), lse.cpu()  # lse: (hq, total)
    for i, L in enumerate(lens):
        o0 = int(cu[i])
        rows = sample_rows(L, seed=100 + i)
        assert len(rows) >= 128
        bias = causal_bias(rows, L, L) if causal else None
        idx = [o0 + r for r in rows]
        _check_rows(out[idx][None], lse[:, idx][None], q[idx][None], k[o0:o0 + L][None], v[o0:o0 + L][None], bias,
                    f"C4 causal={causal} seq {i} (len {L})")


def test_c5_fp8_full_size():
    """C5: e4m3 inputs, b 4 (one GPU's share of the global 32), h 16, s 8192, d 128, random descales rand * 2
    (hopper/test_flash_attn.py:146), bf16 output."""
    from flash_attention_annotated_amd import hopper_interface as fa3
    torch.manual_seed(0)
    b, s, h, d = 4, 8192, 16, 128
    q = torch.randn(b, s, h, d, dtype=torch.bfloat16).to(FP8)
    k = torch.randn(b, s, h, d, dtype=torch.bfloat16).to(FP8)
    v = torch.randn(b, s, h, d, dtype=torch.bfloat16).to(FP8)
    qd, kd, vd = [torch.rand(b, h, dtype=torch.float32) * 2 for _ in range(3)]
    out, lse = fa3.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), q_descale=qd.to(DEV), k_descale=kd.to(DEV),
                                   v_descale=vd.to(DEV), return_attn_probs=True)
    assert out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    out, lse = out.cpu(), lse.cpu()
    q16, k16, v16 = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    for bi in range(b):
        rows = sample_rows(s, seed=200 + bi)
        kw = dict(q_descale=qd[bi:bi + 1], k_descale=kd[bi:bi + 1], v_descale=vd[bi:bi + 1])
        _check_rows(out[bi:bi + 1, rows], lse[bi:bi + 1, :, rows], q16[bi:bi + 1, rows], k16[bi:bi + 1], v16[bi:bi + 1], None,
                    f"C5 batch {bi}", fp8_kw=kw, lse_tol=5e-3)


def _grad_bound(ref, pt):
    """tests/test_flash_attn.py:1129-1132: |dX - dX_ref|max <= 3 |dX_pt - dX_ref|max (+ the FA3 atol, hopper/test_flash_attn.py:262-286)."""
    ref = ref.float()
    return 3 * (pt.float() - ref).abs().max().item() + 2 * (ref + 0.3 - 0.3 - ref).abs().max().item() + 1e-5


@pytest.mark.parametrize("name,s,causal", [("C2", 8192, False), ("C3", 16384, True)])
def test_backward_full_size(name, s, causal):
    """Backward at the BASELINE sequence lengths (b1 of C2 / C3: h16 d128) against the oracle's autograd -- not properties:
      * dQ for a >= 128-row sample with at least one row in every 256-row m-block, ALL heads (dQ_i only needs row i of q and
        dout, so the oracle is differentiated on the row subset with the mask as an additive bias);
      * dK and dV in FULL for 2 heads (every query row contributes: the full attention matrix of those heads).
    This is where the generated dQ loop's cross-tile pipeline over three LDS slots and the dK/dV run-per-head logic execute
    > 100 tiles per workgroup.  Bound: the reference's inequality with factor 3."""
    fa = _api()
    torch.manual_seed(0)
    h, d = 16, 128
    q = torch.randn(1, s, h, d, dtype=torch.bfloat16)
    k = torch.randn(1, s, h, d, dtype=torch.bfloat16)
    v = torch.randn(1, s, h, d, dtype=torch.bfloat16)
    g = torch.randn(1, s, h, d, dtype=torch.bfloat16)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out = fa.flash_attn_func(qd, kd, vd, causal=causal)
    dq, dk, dv = (t.cpu() for t in torch.autograd.grad(out, (qd, kd, vd), g.to(DEV)))
    del out, qd, kd, vd
    assert all(torch.isfinite(t.float()).all() for t in (dq, dk, dv))

    # dQ on the row sample, all heads
    rows = sample_rows(s, n=128, block=256, seed=1)
    idx = torch.tensor(rows)
    bias = causal_bias(rows, s, s) if causal else None

    def dq_rows(**kw):
        ql = q[:, idx].clone().requires_grad_(True)
        o = oracle.attention_ref(ql, k, v, attn_bias=bias, **kw)[0]
        return torch.autograd.grad(o, ql, g[:, idx])[0]
    ref, pt = dq_rows(), dq_rows(upcast=False, reorder_ops=True)
    err = (dq[:, idx].float() - ref.float()).abs().max().item()
    assert err <= _grad_bound(ref, pt), f"{name} dq rows: {err:.3e} > {_grad_bound(ref, pt):.3e}"

    # dK / dV in full, one head at a time (16384^2 fp32 scores = 1 GiB per head)
    for head in (3, 12):
        sl = slice(head, head + 1)

        def dkv(**kw):
            kl, vl = (t[:, :, sl].clone().requires_grad_(True) for t in (k, v))
            o = oracle.attention_ref(q[:, :, sl], kl, vl, causal=causal, **kw)[0]
            return torch.autograd.grad(o, (kl, vl), g[:, :, sl])
        (dk_ref, dv_ref), (dk_pt, dv_pt) = dkv(), dkv(upcast=False, reorder_ops=True)
        for nm, got, r, p in (("dk", dk[:, :, sl], dk_ref, dk_pt), ("dv", dv[:, :, sl], dv_ref, dv_pt)):
            err = (got.float() - r.float()).abs().max().item()
            assert err <= _grad_bound(r, p), f"{name} {nm} head {head}: {err:.3e} > {_grad_bound(r, p):.3e}"
