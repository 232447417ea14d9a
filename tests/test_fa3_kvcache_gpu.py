"""GPU tests of the FA3 KV-cache surface (hopper/flash_attn_interface.py:640-800 `flash_attn_with_kvcache`, i.e.
flash_attn_3::fwd with k_new / page_table / kv_batch_idx / leftpad_k / rotary arguments).

The FA3 entry point is served by the same routines as the FA2 `fwd_kvcache` surface, whose parity against the oracle is
tests/test_kvcache_gpu.py; here: the FA3 call against the oracle for the basic cases (tolerance
|out - ref| <= 3 |pt - ref| + 1e-5, hopper/test_flash_attn.py:1001-1010) and bit-equality (output, LSE, mutated cache)
with the FA2 call for every argument combination.
"""
import pytest
import torch

from oracle import attention_ref as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fa3():
    from flash_attention_annotated_amd import hopper_interface
    return hopper_interface


def _fa2():
    import flash_attention_annotated_amd as fa
    return fa


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("new_kv", [False, True])
@pytest.mark.parametrize("sq,sk,d", [(1, 700, 128), (5, 512, 64)])
def test_fa3_kvcache_against_oracle(sq, sk, d, new_kv, causal):
    torch.manual_seed(sq + sk)
    b, h, hk = 3, 4, 2
    q = torch.randn(b, sq, h, d, dtype=torch.bfloat16)
    kc = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    vc = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    lens = torch.randint(1, sk - sq, (b,), dtype=torch.int32)
    k = torch.randn(b, sq, hk, d, dtype=torch.bfloat16) if new_kv else None
    v = torch.randn(b, sq, hk, d, dtype=torch.bfloat16) if new_kv else None
    kc_ref, vc_ref = kc.clone(), vc.clone()
    total = lens.clone()
    if new_kv:
        for i in range(b):
            kc_ref[i, lens[i]:lens[i] + sq] = k[i]
            vc_ref[i, lens[i]:lens[i] + sq] = v[i]
        total = lens + sq
    kmask = torch.arange(sk).view(1, -1) < total.view(-1, 1)
    ref = oracle.attention_ref(q, kc_ref, vc_ref, None, kmask, causal=causal)[0]
    pt = oracle.attention_ref(q, kc_ref, vc_ref, None, kmask, causal=causal, upcast=False, reorder_ops=True)[0]
    kc_d, vc_d = kc.to(DEV), vc.to(DEV)
    out, lse, *_ = _fa3().flash_attn_with_kvcache(q.to(DEV), kc_d, vc_d, None if k is None else k.to(DEV),
                                                 None if v is None else v.to(DEV), cache_seqlens=lens.to(DEV),
                                                 causal=causal, return_softmax_lse=True)
    err = (out.float().cpu() - ref.float()).abs().max().item()
    assert err <= 3 * (pt.float() - ref.float()).abs().max().item() + 1e-5
    assert tuple(lse.shape) == (b, h, sq)
    assert torch.equal(kc_d.cpu(), kc_ref) and torch.equal(vc_d.cpu(), vc_ref)


@pytest.mark.parametrize("feature", ["batch_idx", "paged", "rotary", "rotary_interleaved", "leftpad", "splits", "int_seqlens",
                                     "local_softcap"])
def test_fa3_kvcache_equals_fa2_surface(feature):
    torch.manual_seed(17)
    b, sq, sk, h, hk, d = 3, 4, 1024, 8, 2, 128
    q = torch.randn(b, sq, h, d, dtype=torch.float16, device=DEV)
    k = torch.randn(b, sq, hk, d, dtype=torch.float16, device=DEV)
    v = torch.randn(b, sq, hk, d, dtype=torch.float16, device=DEV)
    lens = torch.tensor([100, 517, 1000], dtype=torch.int32, device=DEV)
    kw3, kw2 = {}, {}
    bc = b
    if feature == "batch_idx":
        bc = 5
        idx = torch.tensor([4, 0, 2], dtype=torch.int32, device=DEV)
        kw3["cache_batch_idx"] = kw2["cache_batch_idx"] = idx
    kc = torch.randn(bc, sk, hk, d, dtype=torch.float16, device=DEV)
    vc = torch.randn(bc, sk, hk, d, dtype=torch.float16, device=DEV)
    if feature == "paged":
        table = torch.randperm(b * 4, dtype=torch.int32, device=DEV).view(b, 4)
        kc, vc = (x.reshape(b * 4, 256, hk, d).contiguous() for x in (kc, vc))
        kw3["page_table"] = kw2["block_table"] = table
    if feature.startswith("rotary"):
        ang = torch.rand(sk, d // 4, device=DEV) * 6.28
        il = feature == "rotary_interleaved"
        kw3.update(rotary_cos=torch.cos(ang).half(), rotary_sin=torch.sin(ang).half(), rotary_interleaved=il)
        kw2.update(rotary_cos=torch.cos(ang).half(), rotary_sin=torch.sin(ang).half(), rotary_interleaved=il)
    if feature == "leftpad":
        lp = torch.tensor([0, 64, 130], dtype=torch.int32, device=DEV)
        kw3["cache_leftpad"] = kw2["cache_leftpad"] = lp
    if feature == "splits":
        kw3["num_splits"] = kw2["num_splits"] = 4
    if feature == "local_softcap":
        kw3.update(window_size=(200, 0), softcap=20.0)
        kw2.update(window_size=(200, 0), softcap=20.0)
    seqlens = 300 if feature == "int_seqlens" else lens
    if "num_splits" not in kw3:
        kw3["num_splits"] = kw2["num_splits"] = 1
    kc3, vc3, kc2, vc2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    o3, lse3, *_ = _fa3().flash_attn_with_kvcache(q, kc3, vc3, k, v, cache_seqlens=seqlens, causal=True,
                                                 return_softmax_lse=True, **kw3)
    o2, lse2 = _fa2().flash_attn_with_kvcache(q, kc2, vc2, k, v, cache_seqlens=seqlens, causal=True,
                                              return_softmax_lse=True, **kw2)
    assert torch.equal(o3, o2) and torch.equal(lse3, lse2)
    assert torch.equal(kc3, kc2) and torch.equal(vc3, vc2)
    assert not torch.equal(kc3, kc)  # the append happened


def test_fa3_kvcache_rejections():
    fa3 = _fa3()
    q = torch.randn(2, 1, 4, 64, dtype=torch.bfloat16, device=DEV)
    kc = torch.randn(2, 256, 4, 64, dtype=torch.bfloat16, device=DEV)
    lens = torch.tensor([5, 9], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="seqlens_rotary must have dtype torch.int32"):   # hopper/flash_api.cpp:1077
        fa3.flash_attn_with_kvcache(q, kc, kc, cache_seqlens=lens, rotary_seqlens=lens.long())
    with pytest.raises(RuntimeError, match="does not support cu_seqlens_k_new"):
        fa3.flash_attn_with_kvcache(q, kc, kc, cache_seqlens=lens, cu_seqlens_k_new=lens)
    with pytest.raises(RuntimeError, match="k_new and v_new must be passed together"):
        fa3.flash_attn_with_kvcache(q, kc, kc, k=q, cache_seqlens=lens)


@pytest.mark.parametrize("new_kv", [False, True])
@pytest.mark.parametrize("page", [1, 4, 16, 48, 64, 128])
@pytest.mark.parametrize("sq,d", [(1, 128), (9, 64)])
def test_fa3_kvcache_any_page_size(sq, d, page, new_kv):
    """page sizes of hopper/test_flash_attn.py:587 ([1, 4, 128]) and the common serving ones: the paged call equals the
    call on the contiguous cache the pages were scattered from, appended rows land in the right pages."""
    torch.manual_seed(page + sq)
    b, h, hk, sk = 3, 4, 2, 768
    fa3 = _fa3()
    q = torch.randn(b, sq, h, d, dtype=torch.bfloat16, device=DEV)
    kc = torch.randn(b, sk, hk, d, dtype=torch.bfloat16, device=DEV)
    vc = torch.randn(b, sk, hk, d, dtype=torch.bfloat16, device=DEV)
    k = torch.randn(b, sq, hk, d, dtype=torch.bfloat16, device=DEV) if new_kv else None
    v = torch.randn(b, sq, hk, d, dtype=torch.bfloat16, device=DEV) if new_kv else None
    lens = torch.tensor([5, 700, 333], dtype=torch.int32, device=DEV)
    nblk = sk // page
    table = torch.randperm(b * nblk, dtype=torch.int32, device=DEV).view(b, nblk)
    kp = torch.empty(b * nblk, page, hk, d, dtype=torch.bfloat16, device=DEV)
    vp = torch.empty_like(kp)
    kp[table.flatten().long()] = kc.reshape(b * nblk, page, hk, d)
    vp[table.flatten().long()] = vc.reshape(b * nblk, page, hk, d)
    # variant pinned to the paged kernel shape for the contiguous call, so that both runs use the same tiling
    from flash_attention_annotated_amd import _lib
    _lib.load().fa_set_default_variant(1)
    try:
        want, lse_w, *_ = fa3.flash_attn_with_kvcache(q, kc, vc, k, v, cache_seqlens=lens, causal=True, num_splits=1,
                                                     return_softmax_lse=True)
    finally:
        _lib.load().fa_set_default_variant(0)
    got, lse_g, *_ = fa3.flash_attn_with_kvcache(q, kp, vp, k, v, cache_seqlens=lens, page_table=table, causal=True,
                                                num_splits=1, return_softmax_lse=True)
    # (same kernel, but hipcc specialises its main loop for the paged / contiguous address path and contracts the
    #  softmax arithmetic differently in the two copies: results agree to the last bit or two, not always bit-exactly)
    assert (got.float() - want.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, want.float().abs().max().item())
    assert torch.allclose(lse_g, lse_w, atol=1e-5, rtol=1e-6)
    assert torch.equal(kp[table.flatten().long()].reshape(b, sk, hk, d), kc)
    assert torch.equal(vp[table.flatten().long()].reshape(b, sk, hk, d), vc)


@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("causal", [False, True])
def test_fa3_rotary_seqlens(causal, interleaved):
    """`rotary_seqlens` (flash_attn_3::fwd's seqlens_rotary, hopper/flash_api.cpp:1074-1079, hopper/seqlen.h:89): the rotary
    position of the appended keys and of q is taken from it instead of the cache fill level.  Expected: the oracle on a cache
    whose new rows were rotated at positions rotary_seqlens + i by apply_rotary_emb_ref, q rotated at rotary_seqlens (+ row
    under a causal mask)."""
    fa3 = _fa3()
    torch.manual_seed(3)
    b, sq, sk, h, hk, d, rd = 3, 4, 384, 4, 2, 64, 32
    q = torch.randn(b, sq, h, d, dtype=torch.bfloat16)
    kc = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    vc = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    kn = torch.randn(b, sq, hk, d, dtype=torch.bfloat16)
    vn = torch.randn(b, sq, hk, d, dtype=torch.bfloat16)
    lens = torch.tensor([100, 7, 300], dtype=torch.int32)
    rot = torch.tensor([20, 333, 0], dtype=torch.int32)   # not the fill levels
    ang = torch.rand(sk, rd // 2) * 6.28
    cos, sin = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    kcd, vcd = kc.clone().to(DEV), vc.clone().to(DEV)
    out = fa3.flash_attn_with_kvcache(q.to(DEV), kcd, vcd, kn.to(DEV), vn.to(DEV), rotary_cos=cos.to(DEV), rotary_sin=sin.to(DEV),
                                      cache_seqlens=lens.to(DEV), rotary_seqlens=rot.to(DEV), causal=causal,
                                      rotary_interleaved=interleaved)
    k_rot = oracle.apply_rotary_emb_ref(kn, cos, sin, rot, interleaved=interleaved)
    q_rot = oracle.apply_rotary_emb_ref(q, cos, sin, rot, interleaved=interleaved, per_row_positions=causal)
    k_ref, v_ref = kc.clone(), vc.clone()
    for i in range(b):
        k_ref[i, int(lens[i]):int(lens[i]) + sq] = k_rot[i]
        v_ref[i, int(lens[i]):int(lens[i]) + sq] = vn[i]
    assert torch.equal(kcd.cpu(), k_ref) and torch.equal(vcd.cpu(), v_ref)   # the cache holds the rows rotated at rotary_seqlens
    mask = torch.arange(sk).view(1, -1) < (lens + sq).view(-1, 1)
    out_ref, _ = oracle.attention_ref(q_rot, k_ref, v_ref, None, mask, causal=causal)
    out_pt, _ = oracle.attention_ref(q_rot, k_ref, v_ref, None, mask, causal=causal, upcast=False, reorder_ops=True)
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    assert err <= 3 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    # and it differs from the default (positions = fill levels)
    kcd2, vcd2 = kc.clone().to(DEV), vc.clone().to(DEV)
    fa3.flash_attn_with_kvcache(q.to(DEV), kcd2, vcd2, kn.to(DEV), vn.to(DEV), rotary_cos=cos.to(DEV), rotary_sin=sin.to(DEV),
                                cache_seqlens=lens.to(DEV), causal=causal, rotary_interleaved=interleaved)
    assert not torch.equal(kcd2, kcd)
