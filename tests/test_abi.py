"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950 (cross-compiled, no GPU), loads, exports
every symbol include/fa_fwd.h declares, and rejects bad parameters with the documented status codes.  No kernel is
launched here."""
import ctypes
import os
import re

import pytest

from flash_attention_annotated_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    names = set()
    for header in ("fa_fwd.h", "fa_bwd.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(fa_\w+)\s*\(", text))
    return sorted(names)


def test_header_symbols_all_exported(built_lib):
    names = _header_functions()
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), "binding list and header disagree"
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in include/*.h but not exported"


def test_struct_layout_and_version(built_lib):
    assert built_lib.fa_fwd_params_size() == ctypes.sizeof(_lib.FaFwdParams)
    assert built_lib.fa_bwd_params_size() == ctypes.sizeof(_lib.FaBwdParams)
    assert built_lib.fa_abi_version() == _lib.FA_ABI_VERSION


def _good():
    p = _lib.new_params()
    for f in ("q", "k", "v", "o", "softmax_lse"):
        setattr(p, f, 0x10000)
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d = 2, 128, 128, 4, 2, 64
    p.dtype = _lib.FA_DTYPE_BF16
    for t in ("q", "k", "v", "o"):
        setattr(p, f"{t}_row_stride", 4 * 64)
        setattr(p, f"{t}_head_stride", 64)
        setattr(p, f"{t}_batch_stride", 128 * 4 * 64)
    p.k_row_stride = p.v_row_stride = 2 * 64
    p.softmax_scale = 0.125
    p.window_size_left = p.window_size_right = -1
    return p


def test_validate_accepts_good_params(built_lib):
    assert built_lib.fa_fwd_validate(ctypes.byref(_good())) == 0
    assert built_lib.fa_strerror(0) == b"ok"


@pytest.mark.parametrize("mutate,code", [
    (lambda p: setattr(p, "dtype", 7), -2),                 # FA_ERR_BAD_DTYPE
    (lambda p: setattr(p, "d", 264), -3),                   # head dim > 256
    (lambda p: setattr(p, "d", 60), -3),                    # head dim % 8
    (lambda p: setattr(p, "h_k", 3), -4),                   # h % h_k
    (lambda p: setattr(p, "b", 0), -5),                     # batch size must be positive
    (lambda p: setattr(p, "q", 0), -1),                     # NULL tensor
    (lambda p: setattr(p, "q_row_stride", 250), -6),        # rows not 16-byte aligned
    (lambda p: setattr(p, "k", 0x10008), -6),               # base not 16-byte aligned
    (lambda p: setattr(p, "abi_version", 99), -9),          # FA_ERR_BAD_ABI
    (lambda p: setattr(p, "struct_size", 8), -9),
    (lambda p: setattr(p, "attention_chunk", -1), -5),      # ABI v12
    (lambda p: setattr(p, "d_v", 520), -3),                 # V head dim > 512
    (lambda p: setattr(p, "d_v", 100), -3),                 # V head dim % 8
    (lambda p: (setattr(p, "d_v", 128), setattr(p, "dtype", 2), setattr(p, "d", 64)), -7),   # own V head dim: 16-bit types only
    (lambda p: (setattr(p, "d_v", 128), setattr(p, "num_splits", 4)), -7),                   # ... and no split-KV
])
def test_validate_rejects(built_lib, mutate, code):
    p = _good()
    mutate(p)
    st = built_lib.fa_fwd_validate(ctypes.byref(p))
    assert st == code
    assert len(built_lib.fa_strerror(st)) > 0
    # fa_fwd runs the same validation before touching the device: same code, nothing launched
    assert built_lib.fa_fwd(ctypes.byref(p), None) == code


def test_v12_fields_accepted(built_lib):
    """attention_chunk and d_v (include/fa_fwd.h, ABI v12): valid combinations pass validation and need no workspace."""
    for chunk, dv in ((64, 0), (0, 128), (0, 512), (7, 256), (0, 64)):
        p = _good()
        p.attention_chunk, p.d_v, p.num_splits = chunk, dv, 1
        assert built_lib.fa_fwd_validate(ctypes.byref(p)) == 0, (chunk, dv)
        assert built_lib.fa_fwd_workspace_size(ctypes.byref(p)) == 0


def test_error_texts_are_the_reference_messages(built_lib):
    assert b"only support fp16 and bf16" in built_lib.fa_strerror(-2)      # csrc/flash_attn/flash_api.cpp:373
    assert b"at most 256" in built_lib.fa_strerror(-3)                     # :391
    assert b"must divide number of heads in query" in built_lib.fa_strerror(-4)  # :393


def test_tile_shape(built_lib):
    bm, bn = ctypes.c_int32(), ctypes.c_int32()
    assert built_lib.fa_fwd_tile_shape(128, _lib.FA_DTYPE_BF16, 0, ctypes.byref(bm), ctypes.byref(bn)) == 0
    assert (bm.value, bn.value) == (256, 64)
    assert built_lib.fa_fwd_tile_shape(256, _lib.FA_DTYPE_BF16, 1, ctypes.byref(bm), ctypes.byref(bn)) == 0
    assert (bm.value, bn.value) == (128, 64)
    assert built_lib.fa_fwd_tile_shape(300, 0, 0, None, None) == -3


def _good_bwd():
    p = _lib.new_bwd_params()
    for f in ("q", "k", "v", "o", "dout", "softmax_lse", "dq", "dk", "dv", "softmax_d"):
        setattr(p, f, 0x10000)
    p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d = 2, 128, 128, 4, 2, 64
    p.dtype = _lib.FA_DTYPE_BF16
    for t in ("q", "o", "do", "dq"):
        setattr(p, f"{t}_row_stride", 4 * 64)
        setattr(p, f"{t}_head_stride", 64)
        setattr(p, f"{t}_batch_stride", 128 * 4 * 64)
    for t in ("k", "v", "dk", "dv"):
        setattr(p, f"{t}_row_stride", 2 * 64)
        setattr(p, f"{t}_head_stride", 64)
        setattr(p, f"{t}_batch_stride", 128 * 2 * 64)
    p.softmax_d_row_len = 128
    p.softmax_scale = 0.125
    p.window_size_left = p.window_size_right = -1
    return p


@pytest.mark.parametrize("mutate,code", [
    (lambda p: None, 0),
    (lambda p: setattr(p, "dtype", 2), -2),                 # no fp8 backward
    (lambda p: setattr(p, "d", 264), -3),
    (lambda p: setattr(p, "h_k", 3), -4),
    (lambda p: setattr(p, "dout", 0), -1),
    (lambda p: setattr(p, "dk_row_stride", 100), -6),
    (lambda p: setattr(p, "softmax_d_row_len", 64), -5),    # shorter than seqlen_q
    (lambda p: setattr(p, "abi_version", 1), -9),
])
def test_bwd_validate(built_lib, mutate, code):
    """include/fa_bwd.h: the checks of mha_bwd (csrc/flash_attn/flash_api.cpp:803-870) as status codes."""
    p = _good_bwd()
    mutate(p)
    assert built_lib.fa_bwd_validate(ctypes.byref(p)) == code
    if code != 0:
        assert built_lib.fa_bwd(ctypes.byref(p), None) == code


def test_kvcache_append_validate(built_lib):
    """fa_kvcache_append (include/fa_fwd.h): rejected before launch with the documented codes."""
    p = _lib.FaKvcacheAppendParams()
    p.abi_version = _lib.FA_ABI_VERSION
    p.struct_size = ctypes.sizeof(_lib.FaKvcacheAppendParams)
    assert built_lib.fa_kvcache_append_params_size() == ctypes.sizeof(_lib.FaKvcacheAppendParams)
    p.b, p.seqlen_new, p.seqlen_cache, p.h_k, p.d = 2, 1, 128, 2, 64
    for f in ("k_new", "v_new", "k_cache", "v_cache", "cache_seqlens"):
        setattr(p, f, 0x10000)
    for t in ("knew", "vnew", "kcache", "vcache"):
        setattr(p, f"{t}_row_stride", 128)
        setattr(p, f"{t}_head_stride", 64)
        setattr(p, f"{t}_batch_stride", 128 * 128)
    p.d = 60
    assert built_lib.fa_kvcache_append(ctypes.byref(p), None) == -3
    p.d = 64
    p.kcache_row_stride = 100
    assert built_lib.fa_kvcache_append(ctypes.byref(p), None) == -6
    p.kcache_row_stride = 128
    p.cache_seqlens = 0
    assert built_lib.fa_kvcache_append(ctypes.byref(p), None) == -1
    p.abi_version = 1
    assert built_lib.fa_kvcache_append(ctypes.byref(p), None) == -9


def test_fp8_native_switch_is_pinned_by_workspace_size(built_lib):
    """fp8 at head dim 128 runs natively (no workspace) only while one (batch, kv head)'s K and V stay below 2 GiB: the native
    kernel addresses them through 32-bit raw buffer descriptors (fa_fwd_kernel_fp8.h), longer ones take the expansion path
    (ADVICE round 2).  fa_fwd_workspace_size() returns 0 exactly for the native shapes."""
    def fp8(seqlen_k, h_k=32, d=128):
        p = _lib.new_params()
        for f in ("q", "k", "v", "o", "softmax_lse"):
            setattr(p, f, 0x10000)
        p.b, p.seqlen_q, p.seqlen_k, p.h, p.h_k, p.d = 1, 256, seqlen_k, h_k, h_k, d
        p.dtype = _lib.FA_DTYPE_FP8_E4M3
        for t in ("q", "k", "v", "o"):
            setattr(p, f"{t}_row_stride", h_k * d)
            setattr(p, f"{t}_head_stride", d)
        p.q_batch_stride = p.o_batch_stride = 256 * h_k * d
        p.k_batch_stride = p.v_batch_stride = seqlen_k * h_k * d
        p.softmax_scale = 0.088
        p.window_size_left = p.window_size_right = -1
        return p
    size = lambda p: built_lib.fa_fwd_workspace_size(ctypes.byref(p))
    assert size(fp8(8192)) == 0                          # BASELINE config 5's shape: native
    assert size(fp8(8192, d=64)) > 0                     # other head dims: expansion workspace
    assert size(fp8((1 << 31) // (32 * 128) - 1)) == 0   # last row still below 2 GiB
    big = fp8((1 << 31) // (32 * 128))                   # seqlen_k * k_row_stride == 2^31 bytes
    assert size(big) > 0
    big.v_row_stride = 16                                # only K too long: still the expansion path
    assert size(big) > 0
    p = fp8(8192)
    p.k_row_stride = 1 << 31                             # a row stride that does not fit an int
    assert size(p) > 0
