"""GPU tests of the persistent form of the 256-row forward kernel (fwd_kernel_w64<.., PERSIST>, fa_fwd_kernel_w64.h): one
workgroup per CU walks a chain of work items; the look-ahead K / V stream of an item's last tiles fetches the next item's
first tiles, its Q is prefetched into the Q image, O leaves straight from the accumulators.  Same arithmetic in the same
order as the hand-over kernel: results must be BIT-identical to it, and inside the reference's bound against the oracle
(tests/test_flash_attn.py:1121,1556).  Role in the reference: the persistent tile schedulers, hopper/tile_scheduler.hpp:140-363."""
import pytest
import torch

from oracle import attention_ref as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(mode, q, k, v, **kw):
    import flash_attention_annotated_amd as fa
    from flash_attention_annotated_amd import _lib
    lib = _lib.load()
    lib.fa_set_persist_mode(mode)
    try:
        out, lse, _ = fa.flash_attn_func(q, k, v, return_attn_probs=True, **kw)
        torch.cuda.synchronize()
    finally:
        lib.fa_set_persist_mode(0)
    return out, lse


# (b, sq, sk, h, h_k, d, causal): chains of 1 .. 20 items per CU at 256 CUs; GQA, sq < sk, head dims 104 / 128, a head count the
# XCDs do not divide (the `remaining heads` part of the slot list), a chain with padding slots, a partial last m-block
SHAPES = [(32, 512, 512, 16, 16, 128, False), (32, 512, 512, 16, 16, 128, True), (6, 1024, 1024, 20, 20, 128, True),
          (3, 2048, 2048, 21, 7, 128, False), (2, 4096, 4096, 32, 8, 104, True), (16, 300, 1024, 16, 4, 128, True),
          (40, 1000, 1024, 8, 8, 128, False), (5, 1280, 3072, 24, 24, 128, True), (1, 8192, 8192, 40, 40, 128, False),
          (64, 256, 192, 16, 2, 128, False)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_persistent_is_bit_identical_to_handover(shape, dtype):
    b, sq, sk, h, hk, d, causal = shape
    if dtype == torch.float16 and b * sq * h > 2 ** 20:
        pytest.skip("covered in bf16")
    g = torch.Generator(device=DEV).manual_seed(b * 131 + sq)
    q = torch.randn(b, sq, h, d, device=DEV, dtype=dtype, generator=g)
    k = torch.randn(b, sk, hk, d, device=DEV, dtype=dtype, generator=g)
    v = torch.randn(b, sk, hk, d, device=DEV, dtype=dtype, generator=g)
    o0, l0 = _run(-1, q, k, v, causal=causal)
    o1, l1 = _run(1, q, k, v, causal=causal)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    assert not torch.isnan(o1).any()


@pytest.mark.parametrize("window", [(1024, 0), (512, 512), (300, -1), (64, 0), (4096, 0), (200, 77)])
@pytest.mark.parametrize("shape", [(4, 4096, 4096, 16, 16, 128), (2, 2048, 4096, 24, 8, 128), (8, 1024, 1024, 16, 4, 104), (1, 8192, 8192, 20, 20, 128)])
def test_persistent_under_sliding_windows(shape, window):
    """Left windows in the persistent form (round 3): the next item's first key tile is its own n_min -- the look-ahead stream of
    an item's last tiles, the hybrid K tile and the switch offsets of the generated block all start there.  Bit-identical to the
    hand-over kernel; one shape also against the oracle."""
    b, sq, sk, h, hk, d = shape
    g = torch.Generator(device=DEV).manual_seed(sq + window[0])
    q = torch.randn(b, sq, h, d, device=DEV, dtype=torch.bfloat16, generator=g)
    k = torch.randn(b, sk, hk, d, device=DEV, dtype=torch.bfloat16, generator=g)
    v = torch.randn(b, sk, hk, d, device=DEV, dtype=torch.bfloat16, generator=g)
    o0, l0 = _run(-1, q, k, v, window_size=window)
    o1, l1 = _run(1, q, k, v, window_size=window)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    assert not torch.isnan(o1).any()
    if sq == 1024:
        ref_window = (window[0], sk) if (window[0] >= 0 and window[1] < 0) else window
        out_ref, _ = oracle.attention_ref(q.cpu(), k.cpu(), v.cpu(), window_size=ref_window)
        out_pt, _ = oracle.attention_ref(q.cpu(), k.cpu(), v.cpu(), window_size=ref_window, upcast=False, reorder_ops=True)
        err = (o1.float().cpu() - out_ref.float()).abs().max().item()
        assert err <= 2 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5


@pytest.mark.parametrize("causal", [False, True])
def test_persistent_against_oracle(causal):
    """A chain-carrying problem small enough for the fp32 oracle: 1536 tiles on 256 CUs = 6 items per CU, GQA 3:1, strided
    (non-contiguous) inputs, with spikes that trip the generated block's guards in the first and in the last tile of items."""
    torch.manual_seed(5)
    b, s, h, hk, d = 8, 768, 24, 8, 128
    qkv = torch.randn(b, s, h + 2 * hk, d, dtype=torch.bfloat16)
    q, k, v = qkv[:, :, :h], qkv[:, :, h:h + hk], qkv[:, :, h + hk:]
    for i in range(0, b):
        k[i, 5 + i, i % hk] = q[i, 300 + 7 * i, 3 * (i % hk)] * 3.0     # an early key dominating a late row
        k[i, s - 3 - i, i % hk] = q[i, s - 1 - i, 3 * (i % hk) + 1] * 3.0  # a key in the last tile
    out_ref, _, lse_ref = oracle.attention_ref(q, k, v, causal=causal, return_lse=True)
    out_pt, _ = oracle.attention_ref(q, k, v, causal=causal, upcast=False, reorder_ops=True)
    o1, l1 = _run(1, q.to(DEV), k.to(DEV), v.to(DEV), causal=causal)
    err = (o1.float().cpu() - out_ref.float()).abs().max().item()
    bound = 2 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    assert err <= bound, (err, bound)
    fin = torch.isfinite(lse_ref)
    assert (l1.float().cpu()[fin] - lse_ref[fin]).abs().max().item() <= 2e-3


def test_persistent_is_the_default_when_chains_exist():
    """The library's own choice (mode 0) runs the persistent form from two items per CU on: same bits either way, so the check
    is on time -- the default must not be slower than the forced hand-over kernel by more than noise at a short-sequence shape
    where the persistent form wins by ~10 %."""
    import flash_attention_annotated_amd as fa
    from flash_attention_annotated_amd import _lib
    lib = _lib.load()
    q, k, v = (torch.randn(32, 512, 16, 128, device=DEV, dtype=torch.bfloat16) for _ in range(3))

    def t():
        for _ in range(5):
            fa.flash_attn_func(q, k, v, causal=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fa.flash_attn_func(q, k, v, causal=True)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 50
    auto = t()
    lib.fa_set_persist_mode(-1)
    try:
        handover = t()
    finally:
        lib.fa_set_persist_mode(0)
    assert auto <= handover * 1.02, (auto, handover)
