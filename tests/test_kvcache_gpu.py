"""GPU parity tests of the decode path (`flash_attn_with_kvcache` -> fwd_kvcache -> fa_kvcache_append + fa_fwd),
modelled on tests/test_flash_attn.py::test_flash_attn_kvcache (:1885-2165): the expected cache is built by masked
assignment, the expected output by the oracle with a key-padding mask of cache_seqlens (+ appended rows).
Tolerance: the reference's  |out - out_ref| <= 3 |out_pt - out_ref| + 1e-5  (:2153); appended rows are exact copies."""
import pytest
import torch

from oracle import attention_ref as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _api():
    import flash_attention_annotated_amd as fa
    return fa


def _expected(q, k_cache, v_cache, k, v, cache_seqlens, cache_batch_idx, **kw):
    b, sk = q.shape[0], k_cache.shape[1]
    idx = cache_batch_idx.long() if cache_batch_idx is not None else torch.arange(b)
    kc, vc = k_cache[idx].clone(), v_cache[idx].clone()
    new = 0
    if k is not None:
        new = k.shape[1]
        ar = torch.arange(sk).view(1, -1)
        upd = (ar >= cache_seqlens.view(-1, 1)) & (ar < cache_seqlens.view(-1, 1) + new)
        kc[upd] = k.reshape(-1, *k.shape[2:])
        vc[upd] = v.reshape(-1, *v.shape[2:])
    mask = torch.arange(sk).view(1, -1) < (cache_seqlens.view(-1, 1) + new)
    out_ref, _, lse = oracle.attention_ref(q, kc, vc, None, mask, **kw, return_lse=True)
    out_pt, _ = oracle.attention_ref(q, kc, vc, None, mask, **kw, upcast=False, reorder_ops=True)
    return out_ref, out_pt, lse, kc, vc, idx


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mha_type", ["mha", "gqa", "mqa"])
@pytest.mark.parametrize("new_kv", [False, True])
@pytest.mark.parametrize("causal,window", [(False, (-1, -1)), (True, (-1, -1)), (False, (40, 0))])
@pytest.mark.parametrize("has_batch_idx", [False, True])
@pytest.mark.parametrize("sq,sk,d", [(1, 339, 64), (1, 1024, 128), (3, 800, 128), (64, 128, 64), (16, 600, 256)])
def test_kvcache(sq, sk, d, has_batch_idx, causal, window, new_kv, mha_type, dtype):
    fa = _api()
    torch.manual_seed(sq * 7 + sk)
    b, h = 3, 6
    hk = {"mha": 6, "gqa": 2, "mqa": 1}[mha_type]
    b_cache = b + 2 if has_batch_idx else b
    sk_new = (sq if sq > 1 else 1) if new_kv else 0
    q = torch.randn(b, sq, h, d, dtype=dtype)
    k_cache = torch.randn(b_cache, sk, hk, d, dtype=dtype)
    v_cache = torch.randn(b_cache, sk, hk, d, dtype=dtype)
    k = torch.randn(b, sk_new, hk, d, dtype=dtype) if new_kv else None
    v = torch.randn(b, sk_new, hk, d, dtype=dtype) if new_kv else None
    cache_seqlens = torch.randint(0 if new_kv else 1, sk - sk_new + 1, (b,), dtype=torch.int32)
    cache_batch_idx = torch.randperm(b_cache, dtype=torch.int32)[:b] if has_batch_idx else None
    kw = dict(causal=causal, window_size=window)
    out_ref, out_pt, lse_ref, kc_ref, vc_ref, idx = _expected(q, k_cache, v_cache, k, v, cache_seqlens, cache_batch_idx, **kw)

    kc_d, vc_d = k_cache.to(DEV), v_cache.to(DEV)
    out, lse = fa.flash_attn_with_kvcache(
        q.to(DEV), kc_d, vc_d, None if k is None else k.to(DEV), None if v is None else v.to(DEV),
        cache_seqlens=cache_seqlens.to(DEV), cache_batch_idx=None if cache_batch_idx is None else cache_batch_idx.to(DEV),
        causal=causal, window_size=window, return_softmax_lse=True)
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    bound = 3 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    assert err <= bound, f"out err {err:.3e} > {bound:.3e}"
    fin = torch.isfinite(lse_ref)
    assert (lse.cpu()[fin] - lse_ref[fin]).abs().max().item() <= 2e-3
    # the cache: appended rows exact, everything else untouched (also the entries cache_batch_idx does not name)
    got_k, got_v = kc_d.cpu(), vc_d.cpu()
    assert torch.equal(got_k[idx], kc_ref) and torch.equal(got_v[idx], vc_ref)
    others = [i for i in range(b_cache) if i not in idx.tolist()]
    assert torch.equal(got_k[others], k_cache[others]) and torch.equal(got_v[others], v_cache[others])


def test_kvcache_int_seqlens_alibi_softcap_and_out_of_capacity_rows():
    fa = _api()
    torch.manual_seed(3)
    b, h, hk, d, sk = 2, 4, 4, 64, 256
    q = torch.randn(b, 5, h, d, dtype=torch.bfloat16)
    k_cache = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    v_cache = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    slopes = torch.rand(b, h) * 0.3
    out = fa.flash_attn_with_kvcache(q.to(DEV), k_cache.to(DEV), v_cache.to(DEV), cache_seqlens=200, causal=True,
                                     softcap=25.0, alibi_slopes=slopes.to(DEV))
    mask = (torch.arange(sk) < 200).expand(b, sk)
    bias = oracle.attn_bias_from_alibi_slopes(slopes, 5, sk, None, mask)
    out_ref, _ = oracle.attention_ref(q, k_cache, v_cache, None, mask, attn_bias=bias, causal=True, softcap=25.0)
    out_pt, _ = oracle.attention_ref(q, k_cache, v_cache, None, mask, attn_bias=bias, causal=True, softcap=25.0,
                                     upcast=False, reorder_ops=True)
    assert (out.float().cpu() - out_ref.float()).abs().max().item() <= 3 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5


def test_kvcache_rejects_unbuilt_features_by_message():
    fa = _api()
    q = torch.randn(1, 1, 2, 64, dtype=torch.bfloat16, device=DEV)
    kc = torch.randn(1, 256, 2, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="divisible by 256"):
        fa.flash_attn_with_kvcache(q, kc[:, :100], kc[:, :100], block_table=torch.zeros(1, 1, dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError, match="new key / value to be appended to KV cache must also be provided"):
        fa.flash_attn_with_kvcache(q, kc, kc, rotary_cos=torch.zeros(256, 16, device=DEV, dtype=torch.bfloat16),
                                   rotary_sin=torch.zeros(256, 16, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="seqlens_k must also be passed in"):
        fa.flash_attn_with_kvcache(q, kc, kc, k=q[:, :, :2], v=q[:, :, :2])


def _paged(k_cache, v_cache, page, seed):
    """Scatter contiguous (b, sk, h_k, d) caches into a shuffled page pool, as tests/test_flash_attn.py:1975-1992 does."""
    b, sk = k_cache.shape[:2]
    nblk = sk // page
    g = torch.Generator().manual_seed(seed)
    num_blocks = b * nblk * 3
    table = torch.randperm(num_blocks, generator=g, dtype=torch.int32)[: b * nblk].view(b, nblk)
    kp = torch.randn(num_blocks, page, *k_cache.shape[2:], generator=g).to(k_cache.dtype)
    vp = torch.randn(num_blocks, page, *v_cache.shape[2:], generator=g).to(v_cache.dtype)
    kp[table.flatten().long()] = k_cache.reshape(b * nblk, page, *k_cache.shape[2:])
    vp[table.flatten().long()] = v_cache.reshape(b * nblk, page, *v_cache.shape[2:])
    return kp, vp, table


@pytest.mark.parametrize("new_kv", [False, True])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("sq,sk,d,page", [(1, 1024, 128, 256), (1, 2048, 64, 512), (5, 768, 128, 256), (70, 512, 256, 256)])
def test_kvcache_paged(sq, sk, d, page, causal, new_kv):
    """Paged cache (block_table): same expectation as the contiguous cache it was scattered from; appended rows land
    in the right pages and nothing else in the pool changes."""
    fa = _api()
    torch.manual_seed(sq + sk + d)
    b, h, hk = 3, 4, 2
    sk_new = sq if new_kv else 0
    q = torch.randn(b, sq, h, d, dtype=torch.bfloat16)
    k_cache = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    v_cache = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    k = torch.randn(b, sk_new, hk, d, dtype=torch.bfloat16) if new_kv else None
    v = torch.randn(b, sk_new, hk, d, dtype=torch.bfloat16) if new_kv else None
    cache_seqlens = torch.randint(0 if new_kv else 1, sk - sk_new + 1, (b,), dtype=torch.int32)
    out_ref, out_pt, lse_ref, kc_ref, vc_ref, _ = _expected(q, k_cache, v_cache, k, v, cache_seqlens, None, causal=causal)
    kp, vp, table = _paged(k_cache, v_cache, page, seed=d)
    kp_d, vp_d = kp.to(DEV), vp.to(DEV)
    out = fa.flash_attn_with_kvcache(q.to(DEV), kp_d, vp_d, None if k is None else k.to(DEV),
                                     None if v is None else v.to(DEV), cache_seqlens=cache_seqlens.to(DEV),
                                     block_table=table.to(DEV), causal=causal)
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    bound = 3 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    assert err <= bound, f"out err {err:.3e} > {bound:.3e}"
    kp_want, vp_want, _ = _paged(kc_ref, vc_ref, page, seed=d)  # same pool, same table, with the appended rows
    assert torch.equal(kp_d.cpu(), kp_want) and torch.equal(vp_d.cpu(), vp_want)


def test_varlen_paged_kv():
    """flash_attn_varlen_func(..., block_table=...) (csrc/flash_attn/flash_api.cpp:554-560, 608-612): k, v are a page
    pool, cu_seqlens_k gives the lengths."""
    fa = _api()
    torch.manual_seed(9)
    b, h, hk, d, page, sk = 3, 4, 2, 128, 256, 768
    lens_q, lens_k = [40, 1, 130], [700, 256, 300]
    k_cache = torch.randn(b, sk, hk, d, dtype=torch.float16)
    v_cache = torch.randn(b, sk, hk, d, dtype=torch.float16)
    kp, vp, table = _paged(k_cache, v_cache, page, seed=1)
    q = torch.randn(sum(lens_q), h, d, dtype=torch.float16)
    cuq = torch.tensor([0, 40, 41, 171], dtype=torch.int32)
    cuk = torch.tensor([0, 700, 956, 1256], dtype=torch.int32)
    out = fa.flash_attn_varlen_func(q.to(DEV), kp.to(DEV), vp.to(DEV), cuq.to(DEV), cuk.to(DEV), max(lens_q), sk,
                                    causal=True, block_table=table.to(DEV))
    for i in range(b):
        qs = slice(cuq[i], cuq[i + 1])
        ref, _ = oracle.attention_ref(q[qs][None], k_cache[i:i + 1, :lens_k[i]], v_cache[i:i + 1, :lens_k[i]], causal=True)
        pt, _ = oracle.attention_ref(q[qs][None], k_cache[i:i + 1, :lens_k[i]], v_cache[i:i + 1, :lens_k[i]], causal=True,
                                     upcast=False, reorder_ops=True)
        err = (out[qs].float().cpu() - ref[0].float()).abs().max().item()
        assert err <= 2 * (pt.float() - ref.float()).abs().max().item() + 1e-5, (i, err)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("rotary_fraction", [0.5, 1.0])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("paged", [False, True])
@pytest.mark.parametrize("sq,sk,d", [(1, 512, 128), (4, 768, 64), (33, 512, 128)])
def test_kvcache_rotary(sq, sk, d, paged, causal, rotary_fraction, interleaved, dtype):
    """Rotary embedding of q and of the appended keys (tests/test_flash_attn.py:2005-2036): keys at position
    cache_seqlens + i; queries at the same positions when causal, else all at cache_seqlens.  The cache comparison uses
    the reference's rtol = atol = 1e-3 (:2160-2161)."""
    fa = _api()
    torch.manual_seed(sq + sk + int(rotary_fraction * 10))
    b, h, hk = 2, 4, 2
    rotary_dim = int(rotary_fraction * d) // 16 * 16
    q = torch.randn(b, sq, h, d, dtype=dtype)
    k_cache = torch.randn(b, sk, hk, d, dtype=dtype)
    v_cache = torch.randn(b, sk, hk, d, dtype=dtype)
    k = torch.randn(b, sq, hk, d, dtype=dtype)
    v = torch.randn(b, sq, hk, d, dtype=dtype)
    cache_seqlens = torch.randint(0, sk - sq + 1, (b,), dtype=torch.int32)
    angle = torch.rand(sk, rotary_dim // 2) * 2 * 3.141592653589793
    cos, sin = torch.cos(angle).to(dtype), torch.sin(angle).to(dtype)
    q_ro = oracle.apply_rotary_emb_ref(q, cos, sin, cache_seqlens, interleaved, per_row_positions=causal)
    k_ro = oracle.apply_rotary_emb_ref(k, cos, sin, cache_seqlens, interleaved, per_row_positions=True)
    out_ref, out_pt, _, kc_ref, vc_ref, _ = _expected(q_ro, k_cache, v_cache, k_ro, v, cache_seqlens, None, causal=causal)
    if paged:
        kc, vc, table = _paged(k_cache, v_cache, 256, seed=5)
        kc_d, vc_d, table_d = kc.to(DEV), vc.to(DEV), table.to(DEV)
    else:
        kc_d, vc_d, table_d = k_cache.to(DEV), v_cache.to(DEV), None
    out = fa.flash_attn_with_kvcache(q.to(DEV), kc_d, vc_d, k.to(DEV), v.to(DEV), rotary_cos=cos.to(DEV),
                                     rotary_sin=sin.to(DEV), cache_seqlens=cache_seqlens.to(DEV), block_table=table_d,
                                     causal=causal, rotary_interleaved=interleaved)
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    bound = 3 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    assert err <= bound, f"out err {err:.3e} > {bound:.3e}"
    if paged:
        kp_want, vp_want, _ = _paged(kc_ref, vc_ref, 256, seed=5)
        assert torch.allclose(kc_d.cpu().float(), kp_want.float(), rtol=1e-3, atol=1e-3) and torch.equal(vc_d.cpu(), vp_want)
    else:
        assert torch.allclose(kc_d.cpu().float(), kc_ref.float(), rtol=1e-3, atol=1e-3) and torch.equal(vc_d.cpu(), vc_ref)


@pytest.mark.parametrize("num_splits", [0, 2, 5, 16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("b,sq,sk,h,hk,d", [(1, 1, 4096, 8, 2, 128), (2, 1, 1500, 4, 4, 64), (1, 7, 3000, 4, 1, 128),
                                             (2, 16, 2048, 2, 2, 256), (1, 130, 1024, 4, 2, 64)])
def test_kvcache_split_kv(b, sq, sk, h, hk, d, causal, num_splits):
    """Split-KV (num_splits forced, or 0 = heuristic): the merged result obeys the same bound as the unsplit one, LSE
    included; ragged cache_seqlens leave some splits without keys (weight 0 in the merge)."""
    fa = _api()
    torch.manual_seed(b * 100 + sq + num_splits)
    q = torch.randn(b, sq, h, d, dtype=torch.bfloat16)
    k_cache = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    v_cache = torch.randn(b, sk, hk, d, dtype=torch.bfloat16)
    cache_seqlens = torch.randint(sq, sk + 1, (b,), dtype=torch.int32)
    cache_seqlens[0] = max(sq, sk // 9)  # far shorter than the cache: most splits of this row are empty
    out_ref, out_pt, lse_ref, _, _, _ = _expected(q, k_cache, v_cache, None, None, cache_seqlens, None, causal=causal)
    out, lse = fa.flash_attn_with_kvcache(q.to(DEV), k_cache.to(DEV), v_cache.to(DEV), cache_seqlens=cache_seqlens.to(DEV),
                                          causal=causal, num_splits=num_splits, return_softmax_lse=True)
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    bound = 3 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    assert err <= bound, f"out err {err:.3e} > {bound:.3e}"
    fin = torch.isfinite(lse_ref)
    assert torch.equal(torch.isfinite(lse.cpu()), fin)
    assert (lse.cpu()[fin] - lse_ref[fin]).abs().max().item() <= 2e-3
    out1 = fa.flash_attn_with_kvcache(q.to(DEV), k_cache.to(DEV), v_cache.to(DEV), cache_seqlens=cache_seqlens.to(DEV),
                                      causal=causal, num_splits=1)
    assert (out.float() - out1.float()).abs().max().item() <= 4 * bound  # split and unsplit agree to rounding


@pytest.mark.parametrize("alibi", [False, True])
@pytest.mark.parametrize("new_kv,rotary", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("causal,window", [(False, (-1, -1)), (True, (-1, -1)), (False, (30, 5))])
@pytest.mark.parametrize("sq,sk,d", [(1, 600, 128), (6, 512, 64), (40, 384, 128)])
def test_kvcache_leftpad(sq, sk, d, causal, window, new_kv, rotary, alibi):
    """cache_leftpad (tests/test_flash_attn.py:1978-1992, 2094-2109): the first leftpad rows of each cache entry are
    padding; masks, ALiBi distances and the bottom-right alignment count from the first real key; appended rows and
    rotary positions stay absolute."""
    fa = _api()
    torch.manual_seed(sq * 3 + sk + d)
    b, h, hk = 3, 4, 2
    dtype = torch.bfloat16
    sk_new = sq if new_kv else 0
    q = torch.randn(b, sq, h, d, dtype=dtype)
    k_cache = torch.randn(b, sk, hk, d, dtype=dtype)
    v_cache = torch.randn(b, sk, hk, d, dtype=dtype)
    k = torch.randn(b, sk_new, hk, d, dtype=dtype) if new_kv else None
    v = torch.randn(b, sk_new, hk, d, dtype=dtype) if new_kv else None
    cache_seqlens = torch.randint(max(sq, 8), sk - sk_new + 1, (b,), dtype=torch.int32)
    leftpad = torch.stack([torch.randint(0, int(c), (1,), dtype=torch.int32)[0] for c in cache_seqlens]).to(torch.int32)
    leftpad[0] = 0
    cos = sin = None
    q_in, k_in = q, k
    if rotary:
        angle = torch.rand(sk, d // 4) * 6.283185
        cos, sin = torch.cos(angle).to(dtype), torch.sin(angle).to(dtype)
        per_row = causal or window != (-1, -1)
        q_in = oracle.apply_rotary_emb_ref(q, cos, sin, cache_seqlens, False, per_row_positions=per_row)
        k_in = oracle.apply_rotary_emb_ref(k, cos, sin, cache_seqlens, False, per_row_positions=True)
    # expectation: cache with the appended rows, keys valid in [leftpad, cache_seqlens + new)
    kc, vc = k_cache.clone(), v_cache.clone()
    ar = torch.arange(sk).view(1, -1)
    if new_kv:
        upd = (ar >= cache_seqlens.view(-1, 1)) & (ar < cache_seqlens.view(-1, 1) + sk_new)
        kc[upd] = k_in.reshape(-1, hk, d)
        vc[upd] = v.reshape(-1, hk, d)
    mask = (ar < cache_seqlens.view(-1, 1) + sk_new) & (ar >= leftpad.view(-1, 1))
    slopes = torch.rand(b, h) * 0.3 if alibi else None
    bias = None if slopes is None else oracle.attn_bias_from_alibi_slopes(slopes, sq, sk, None, mask, key_leftpad=leftpad)
    kw = dict(attn_bias=bias, causal=causal, window_size=window, key_leftpad=leftpad)
    out_ref, _ = oracle.attention_ref(q_in, kc, vc, None, mask, **kw)
    out_pt, _ = oracle.attention_ref(q_in, kc, vc, None, mask, **kw, upcast=False, reorder_ops=True)
    kc_d, vc_d = k_cache.to(DEV), v_cache.to(DEV)
    out = fa.flash_attn_with_kvcache(
        q.to(DEV), kc_d, vc_d, None if k is None else k.to(DEV), None if v is None else v.to(DEV),
        rotary_cos=None if cos is None else cos.to(DEV), rotary_sin=None if sin is None else sin.to(DEV),
        cache_seqlens=cache_seqlens.to(DEV), cache_leftpad=leftpad.to(DEV), causal=causal, window_size=window,
        rotary_interleaved=False, alibi_slopes=None if slopes is None else slopes.to(DEV))
    err = (out.float().cpu() - out_ref.float()).abs().max().item()
    bound = 3 * (out_pt.float() - out_ref.float()).abs().max().item() + 1e-5
    assert err <= bound, f"out err {err:.3e} > {bound:.3e}"
    assert torch.allclose(kc_d.cpu().float(), kc.float(), rtol=1e-3, atol=1e-3) and torch.equal(vc_d.cpu(), vc)
