"""CPU tests of the Python host layer: it mirrors the reference's operator interface (names, argument order,
defaults) and it never computes attention on a fallback path."""
import inspect

import pytest
import torch

import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import bert_padding, flash_attn_2_cuda, sharding


def _params(f):
    return [(n, p.default) for n, p in inspect.signature(f).parameters.items()]


def test_flash_attn_func_signature():
    """flash_attn/flash_attn_interface.py:1145-1157"""
    assert _params(fa.flash_attn_func) == [
        ("q", inspect._empty), ("k", inspect._empty), ("v", inspect._empty), ("dropout_p", 0.0),
        ("softmax_scale", None), ("causal", False), ("window_size", (-1, -1)), ("softcap", 0.0),
        ("alibi_slopes", None), ("deterministic", False), ("return_attn_probs", False)]


def test_flash_attn_varlen_func_signature():
    """flash_attn/flash_attn_interface.py:1380-1397"""
    assert _params(fa.flash_attn_varlen_func) == [
        ("q", inspect._empty), ("k", inspect._empty), ("v", inspect._empty), ("cu_seqlens_q", inspect._empty),
        ("cu_seqlens_k", inspect._empty), ("max_seqlen_q", inspect._empty), ("max_seqlen_k", inspect._empty),
        ("dropout_p", 0.0), ("softmax_scale", None), ("causal", False), ("window_size", (-1, -1)), ("softcap", 0.0),
        ("alibi_slopes", None), ("deterministic", False), ("return_attn_probs", False), ("block_table", None)]


def test_extension_module_surface():
    """The five entry points flash_attn_interface.py looks up on flash_attn_2_cuda (:91,:168,:269,:369,:1594),
    with the positional arity of the pybind signatures (csrc/flash_attn/flash_api.cpp:350-363, 514-535)."""
    import re

    def arity(fn):  # pybind functions carry their signature in the first docstring line
        try:
            return len(inspect.signature(fn).parameters)
        except ValueError:
            return len(re.findall(r"\barg\d+:", fn.__doc__.splitlines()[0]))
    want = {"fwd": 13, "varlen_fwd": 21, "bwd": 19, "varlen_bwd": 24, "fwd_kvcache": 20}  # :350-363, 514-535, 767-786, 973-997, 1202-1222
    for name, n in want.items():
        assert callable(getattr(flash_attn_2_cuda, name))
        assert arity(getattr(flash_attn_2_cuda, name)) == n, name
        assert arity(flash_attn_2_cuda._py_entry_points[name]) == n, name   # the ctypes fallback states the same lists
    import flash_attn_2_cuda as top_level_alias  # the name the reference imports
    assert top_level_alias.fwd is flash_attn_2_cuda.fwd
    if flash_attn_2_cuda.compiled is not None:  # the reference's flash_attn_gpu is then the compiled module's functions
        assert top_level_alias.fwd is flash_attn_2_cuda.compiled.fwd
        assert type(top_level_alias.fwd).__name__ == "builtin_function_or_method"
    assert [n for n, _ in _params(fa.flash_attn_with_kvcache)] == [
        "q", "k_cache", "v_cache", "k", "v", "rotary_cos", "rotary_sin", "cache_seqlens", "cache_batch_idx",
        "cache_leftpad", "block_table", "softmax_scale", "causal", "window_size", "softcap", "rotary_interleaved",
        "alibi_slopes", "num_splits", "return_softmax_lse"]  # flash_attn/flash_attn_interface.py:1474-1494


def test_no_cpu_fallback():
    """CPU tensors are rejected like the reference's CHECK_DEVICE does; nothing is computed off the GPU."""
    q = torch.randn(1, 16, 2, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fa.flash_attn_func(q, q, q)
    with pytest.raises(RuntimeError, match="only support fp16 and bf16"):
        fa.flash_attn_func(q.float(), q.float(), q.float())


def test_unpad_pad_roundtrip():
    torch.manual_seed(0)
    x = torch.randn(3, 7, 2, 4)
    mask = torch.tensor([[1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1], [1, 0, 0, 0, 0, 0, 0]], dtype=torch.bool)
    xu, idx, cu, mx, used = bert_padding.unpad_input(x, mask)
    assert cu.tolist() == [0, 3, 10, 11] and cu.dtype == torch.int32 and mx == 7 and used.tolist() == [3, 7, 1]
    back = bert_padding.pad_input(xu, idx, 3, 7)
    assert torch.equal(back[mask], x[mask]) and torch.all(back[~mask] == 0)


def test_shard_range_partitions():
    for n in (1, 4, 7, 32):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_torch_library_ops_have_fake_implementations():
    """torch.compile surface (reference flash_attn/flash_attn_interface.py:76,109-136,241,292-325): the host entry
    points are custom ops whose fake kernels give the shapes of the real ones; meta tensors exercise them."""
    q = torch.empty(2, 100, 4, 64, dtype=torch.bfloat16, device="meta")
    k = torch.empty(2, 130, 2, 64, dtype=torch.bfloat16, device="meta")
    out, lse, p, rng = torch.ops.flash_attn_amd._flash_attn_forward(q, k, k, 0.0, 0.125, True, -1, -1, 0.0, None, False)
    assert out.shape == q.shape and lse.shape == (2, 4, 100) and lse.dtype == torch.float32 and p.shape == (0,)
    sd = torch.ops.flash_attn_amd._flash_attn_backward(out, q, k, k, out, lse, torch.empty_like(q), torch.empty_like(k),
                                                       torch.empty_like(k), 0.0, 0.125, True, -1, -1, 0.0, None, False)
    assert sd.shape == (2, 4, 128)
    qv = torch.empty(393, 4, 64, dtype=torch.float16, device="meta")
    cu = torch.empty(4, dtype=torch.int32, device="meta")
    out, lse, _, _ = torch.ops.flash_attn_amd._flash_attn_varlen_forward(qv, qv, qv, cu, cu, 256, 256, 0.0, 0.125, False)
    assert out.shape == qv.shape and lse.shape == (4, 393)
    sd = torch.ops.flash_attn_amd._flash_attn_varlen_backward(out, qv, qv, qv, out, lse, None, None, None, cu, cu, 256,
                                                              256, 0.0, 0.125, False, -1, -1, 0.0, None, False)
    assert sd.shape == (4, 393 + 128 * 3)


def test_attention_modules_mirror_the_reference_constructors():
    from flash_attention_annotated_amd.modules.mha import FlashCrossAttention, FlashSelfAttention
    sa = FlashSelfAttention(causal=True, softmax_scale=0.1, attention_dropout=0.0, window_size=(4, 0))
    ca = FlashCrossAttention(causal=False, alibi_slopes=torch.rand(4))
    assert [p for p in inspect.signature(sa.forward).parameters] == ["qkv", "causal", "cu_seqlens", "max_seqlen"]
    assert [p for p in inspect.signature(ca.forward).parameters] == [
        "q", "kv", "causal", "cu_seqlens", "max_seqlen", "cu_seqlens_k", "max_seqlen_k"]
    with pytest.raises(AssertionError):
        sa(torch.randn(1, 8, 3, 2, 64, dtype=torch.bfloat16))  # CPU tensor: rejected like the reference does


def test_flash3_bw_compatibility():
    """hopper/test_flash_attn.py:1163-1201: the four `flash_attn_3` op schemas stay backward compatible with the
    reference's (arguments only appended with defaults).  The strings below are the reference test's."""
    from torch._C import parse_schema
    import flash_attention_annotated_amd.hopper_interface  # noqa: F401  (registers torch.ops.flash_attn_3)
    assert torch.ops.flash_attn_3.fwd.default._schema.is_backward_compatible_with(parse_schema(
        "flash_attn_3::fwd(Tensor q, Tensor k, Tensor v, Tensor(k_new!)? k_new=None, "
        "Tensor(v_new!)? v_new=None, Tensor? q_v=None, Tensor(out!)? out=None, "
        "Tensor? cu_seqlens_q=None, Tensor? cu_seqlens_k=None, "
        "Tensor? cu_seqlens_k_new=None, Tensor? seqused_q=None, Tensor? seqused_k=None, "
        "int? max_seqlen_q=None, int? max_seqlen_k=None, Tensor? page_table=None, "
        "Tensor? kv_batch_idx=None, Tensor? leftpad_k=None, Tensor? rotary_cos=None, Tensor? rotary_sin=None, "
        "Tensor? seqlens_rotary=None, Tensor? q_descale=None, Tensor? k_descale=None, Tensor? v_descale=None, "
        "float? softmax_scale=None, bool is_causal=False, int window_size_left=-1, int window_size_right=-1, "
        "int attention_chunk=0, float softcap=0., bool is_rotary_interleaved=False, "
        "Tensor? scheduler_metadata=None, int num_splits=0, bool? pack_gqa=None, int sm_margin=0) "
        "-> (Tensor(out!), Tensor, Tensor, Tensor)"))
    assert torch.ops.flash_attn_3.bwd.default._schema.is_backward_compatible_with(parse_schema(
        "flash_attn_3::bwd(Tensor dout, Tensor q, Tensor k, Tensor v, Tensor out, Tensor softmax_lse, "
        "Tensor(dq!)? dq=None, Tensor(dk!)? dk=None, Tensor(dv!)? dv=None, Tensor? cu_seqlens_q=None, "
        "Tensor? cu_seqlens_k=None, Tensor? seqused_q=None, Tensor? seqused_k=None, int? max_seqlen_q=None, "
        "int? max_seqlen_k=None, float? softmax_scale=None, bool is_causal=False, int window_size_left=-1, "
        "int window_size_right=-1, float softcap=0., bool deterministic=False, int sm_margin=0) "
        "-> (Tensor(dq!), Tensor(dk!), Tensor(dv!), Tensor, Tensor, Tensor, Tensor, Tensor)"))
    assert torch.ops.flash_attn_3.fwd_combine.default._schema.is_backward_compatible_with(parse_schema(
        "flash_attn_3::fwd_combine(Tensor out_partial, Tensor lse_partial, Tensor(out!)? out=None, "
        "ScalarType? out_dtype=None) -> (Tensor(out!), Tensor)"))
    assert torch.ops.flash_attn_3.get_scheduler_metadata.default._schema.is_backward_compatible_with(parse_schema(
        "flash_attn_3::get_scheduler_metadata(int batch_size, int max_seqlen_q, int max_seqlen_k, "
        "int num_heads, int num_heads_k, int headdim, int headdim_v, ScalarType qkv_dtype, Tensor seqused_k, "
        "Tensor? cu_seqlens_q=None, Tensor? cu_seqlens_k=None, Tensor? cu_seqlens_k_new=None, "
        "Tensor? seqused_q=None, Tensor? leftpad_k=None, int? page_size=None, int max_seqlen_k_new=0, "
        "bool is_causal=False, int window_size_left=-1, int window_size_right=-1, "
        "int attention_chunk=0, bool has_softcap=False, int num_splits=0, bool? pack_gqa=None, "
        "int sm_margin=0) -> Tensor"))
