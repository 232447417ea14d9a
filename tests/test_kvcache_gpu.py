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
    with pytest.raises(RuntimeError, match="paged KV"):
        fa.flash_attn_with_kvcache(q, kc, kc, block_table=torch.zeros(1, 1, dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError, match="rotary"):
        fa.flash_attn_with_kvcache(q, kc, kc, rotary_cos=torch.zeros(256, 16, device=DEV), rotary_sin=torch.zeros(256, 16, device=DEV))
    with pytest.raises(RuntimeError, match="seqlens_k must also be passed in"):
        fa.flash_attn_with_kvcache(q, kc, kc, k=q[:, :, :2], v=q[:, :, :2])
