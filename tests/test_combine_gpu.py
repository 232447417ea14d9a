"""GPU parity of flash_attn_combine / torch.ops.flash_attn_3.fwd_combine (fa_fwd_combine, include/fa_fwd.h).

Modelled on hopper/test_flash_attn.py:1117-1155: non-contiguous partials, -inf splits (short-circuit), every output
type.  Bounds (floating point): lse allclose(atol=1e-5, rtol=1e-5); |out - out_ref|max <= 2 |out_pt - out_ref|max
or allclose(out, out_pt, 1e-5), out_pt = the fp32 oracle result rounded to the output type.
"""
import pytest
import torch

from oracle import attention_ref as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fa3():
    from flash_attention_annotated_amd import hopper_interface
    return hopper_interface


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [64, 59, 256, 512])
@pytest.mark.parametrize("seqlen", [1, 113, 640])
@pytest.mark.parametrize("num_splits", [1, 3, 17, 133])
def test_flash_attn_combine(num_splits, seqlen, d, dtype):
    torch.random.manual_seed(1)
    batch_size, nheads = 5, 16
    out_partial = torch.randn(num_splits * 2, batch_size, nheads, seqlen, d, device=DEV,
                              dtype=torch.float32).transpose(2, 3)[:num_splits]  # non-contiguous
    lse_partial = torch.randn(num_splits, batch_size, nheads * 2, seqlen, device=DEV,
                              dtype=torch.float32).transpose(-1, -2)[:, :, :, :nheads]  # non-contiguous
    lse_partial[num_splits // 2:, :batch_size // 3] = -float("inf")
    out, lse = _fa3().flash_attn_combine(out_partial, lse_partial, out_dtype=dtype)
    assert out.dtype == dtype and tuple(out.shape) == (batch_size, seqlen, nheads, d)
    assert tuple(lse.shape) == (batch_size, seqlen, nheads) and lse.stride(1) == 1  # (b, h, s) storage, hopper/flash_api.cpp:1632
    out_ref, lse_ref = oracle.attention_combine_ref(out_partial.cpu(), lse_partial.cpu())
    out_pt = out_ref.to(dtype)
    assert torch.allclose(lse.cpu(), lse_ref, atol=1e-5, rtol=1e-5)
    err = (out.cpu().float() - out_ref).abs().max().item()
    assert err <= 2 * (out_pt.float() - out_ref).abs().max().item() or torch.allclose(out.cpu(), out_pt, atol=1e-5, rtol=1e-5)


def test_combine_golden_and_out_argument(golden_grads):
    gold = golden_grads["combine_pin"]
    op = gold["out_partial"].to(DEV)
    lp = gold["lse_partial"].to(DEV).transpose(2, 3).contiguous().transpose(2, 3)  # seqlen-contiguous, as the op demands
    dst = torch.full((3, 17, 4, 40), 7.0, device=DEV)
    out, lse = _fa3().flash_attn_combine(op, lp, out=dst)
    assert out.data_ptr() == dst.data_ptr()
    assert torch.allclose(out.cpu(), gold["out"], atol=1e-5, rtol=1e-5)
    assert (out[2, 3] == 0).all() and torch.isinf(lse[2, 3]).all() and (lse[2, 3] < 0).all()
    fin = ~torch.isinf(gold["lse"])
    assert torch.allclose(lse.cpu()[fin], gold["lse"][fin], atol=1e-5, rtol=1e-5)


def test_combine_merges_split_results_of_the_attention_kernel():
    """Linearity property at a size the oracle does not reach: attention over 4 disjoint key ranges, merged by
    fwd_combine, equals attention over all keys."""
    import flash_attention_annotated_amd as fa
    torch.manual_seed(3)
    b, sq, sk, h, d = 2, 512, 4096, 8, 128
    q = torch.randn(b, sq, h, d, device=DEV, dtype=torch.bfloat16)
    k = torch.randn(b, sk, h, d, device=DEV, dtype=torch.bfloat16)
    v = torch.randn(b, sk, h, d, device=DEV, dtype=torch.bfloat16)
    full, lse_full, _ = fa.flash_attn_func(q, k, v, 0.0, return_attn_probs=True)
    parts = [fa.flash_attn_func(q, k[:, i:i + 1024], v[:, i:i + 1024], 0.0, return_attn_probs=True) for i in range(0, sk, 1024)]
    op = torch.stack([p[0].float() for p in parts])
    lp = torch.stack([p[1] for p in parts]).transpose(2, 3)  # (S, b, h, s) storage viewed as (S, b, s, h)
    out, lse = _fa3().flash_attn_combine(op, lp, out_dtype=torch.bfloat16)
    assert torch.allclose(lse.transpose(1, 2), lse_full, atol=1e-4, rtol=1e-5)
    assert (out.float() - full.float()).abs().max().item() <= 2e-2


def test_combine_errors():
    f = _fa3().flash_attn_combine
    op = torch.randn(2, 1, 4, 2, 64, device=DEV)
    lp = torch.randn(2, 1, 4, 2, device=DEV).transpose(-1, -2).contiguous().transpose(-1, -2)
    with pytest.raises(RuntimeError, match="only support fp32"):
        f(op.half(), lp)
    with pytest.raises(RuntimeError, match="Output type must be FP32, FP16 or BF16"):
        f(op, lp, out_dtype=torch.float64)
    with pytest.raises(RuntimeError, match="num_splits at most 256"):
        f(torch.randn(257, 1, 1, 1, 8, device=DEV), torch.randn(257, 1, 1, 1, device=DEV))
