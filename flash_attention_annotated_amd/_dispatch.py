"""Shared host glue: torch tensors -> fa_fwd_params -> fa_fwd on torch's current stream.  Used by the FA2-shaped
module (flash_attn_2_cuda.py) and the FA3-shaped one (flash_attn_3_cuda.py).  No compute happens here."""
import ctypes

import torch

from . import _lib

_DT = {torch.float16: _lib.FA_DTYPE_FP16, torch.bfloat16: _lib.FA_DTYPE_BF16}
if hasattr(torch, "float8_e4m3fn"):
    _DT[torch.float8_e4m3fn] = _lib.FA_DTYPE_FP8_E4M3


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def aligned(t):
    """The kernels move 16-byte vectors (the fp8 expansion pass 8-byte ones): bases and the non-unit strides must keep
    rows aligned.  Views that are not get copied by the callers."""
    esz = t.element_size()
    if t.data_ptr() % (16 if esz == 2 else 8) != 0:
        return False
    return all(s % 8 == 0 for s in t.stride()[:-1])


def launch(q, k, v, out, lse, *, varlen, batch, max_seqlen_q, max_seqlen_k, softmax_scale, causal, window_left,
           window_right, softcap, cu_seqlens_q=None, cu_seqlens_k=None, seqused_q=None, seqused_k=None,
           q_descale=None, k_descale=None, v_descale=None, alibi_slopes=None, kv_batch_idx=None, block_table=None, num_splits=1, leftpad_k=None,
           p_dropout=0.0, rng_state=None, s_dmask=None, fa3_window=False, s_dmask_block_n=0, attention_chunk=0):
    """q/k/v/out: dense (b, s, h, d) or packed (total, h, d) tensors on one GPU, last stride 1, aligned()."""
    lib = _lib.load()
    prm = _lib.new_params()
    prm.q, prm.k, prm.v, prm.o = ptr(q), ptr(k), ptr(v), ptr(out)
    prm.softmax_lse = ptr(lse)
    if varlen:
        for name, t in (("q", q), ("o", out)) + ((() if block_table is not None else (("k", k), ("v", v)))):
            setattr(prm, f"{name}_batch_stride", 0)
            setattr(prm, f"{name}_row_stride", t.stride(0))
            setattr(prm, f"{name}_head_stride", t.stride(1))
        if block_table is not None:  # k, v: (num_blocks, page_block_size, h_k, d)
            for name, t in (("k", k), ("v", v)):
                setattr(prm, f"{name}_batch_stride", t.stride(0))
                setattr(prm, f"{name}_row_stride", t.stride(1))
                setattr(prm, f"{name}_head_stride", t.stride(2))
        prm.total_q, prm.total_k = q.shape[0], (0 if block_table is not None else k.shape[0])
        prm.h, prm.h_k, prm.d = q.shape[1], k.shape[-2], q.shape[2]
    else:
        for name, t in (("q", q), ("k", k), ("v", v), ("o", out)):
            setattr(prm, f"{name}_batch_stride", t.stride(0))
            setattr(prm, f"{name}_row_stride", t.stride(1))
            setattr(prm, f"{name}_head_stride", t.stride(2))
        prm.total_q = prm.total_k = 0
        prm.h, prm.h_k, prm.d = q.shape[2], k.shape[2], q.shape[3]
    prm.b, prm.seqlen_q, prm.seqlen_k = int(batch), int(max_seqlen_q), int(max_seqlen_k)
    prm.dtype = _DT[q.dtype]
    prm.cu_seqlens_q, prm.cu_seqlens_k = ptr(cu_seqlens_q), ptr(cu_seqlens_k)
    prm.seqused_q, prm.seqused_k = ptr(seqused_q), ptr(seqused_k)
    prm.softmax_scale = float(softmax_scale)
    prm.softcap = float(softcap)
    prm.is_causal = int(bool(causal))
    prm.window_size_left, prm.window_size_right = int(window_left), int(window_right)
    for name, t in (("q", q_descale), ("k", k_descale), ("v", v_descale)):
        setattr(prm, f"{name}_descale", ptr(t))
        if t is not None:
            setattr(prm, f"{name}_descale_batch_stride", t.stride(0))
            setattr(prm, f"{name}_descale_head_stride", t.stride(1))
    if alibi_slopes is not None:  # (h) or (b, h) fp32, last stride 1 (checked by the callers)
        prm.alibi_slopes = ptr(alibi_slopes)
        prm.alibi_slopes_batch_stride = alibi_slopes.stride(0) if alibi_slopes.dim() == 2 else 0
    prm.kv_batch_idx = ptr(kv_batch_idx)
    prm.leftpad_k = ptr(leftpad_k)
    prm.p_dropout = float(p_dropout)
    prm.rng_state, prm.s_dmask = ptr(rng_state), ptr(s_dmask)
    prm.flags = _lib.FA_FLAG_FA3_WINDOW if fa3_window else 0
    if s_dmask is not None and s_dmask_block_n > 0:  # the reference's sign-encoded layout (b, h, rows, cols), input dtype
        prm.flags |= _lib.FA_FLAG_SDMASK_SIGNED
        prm.s_dmask_rows, prm.s_dmask_cols, prm.s_dmask_block_n = s_dmask.shape[-2], s_dmask.shape[-1], int(s_dmask_block_n)
    prm.attention_chunk = int(attention_chunk)
    prm.d_v = int(v.shape[-1]) if v.shape[-1] != q.shape[-1] else 0  # FA3 headdim_v (include/fa_fwd.h, ABI v12)
    prm.num_splits = int(num_splits)  # 1 = off (prefill entry points), 0 = library heuristic (decode), N = forced
    if block_table is not None:
        prm.block_table = ptr(block_table)
        prm.block_table_batch_stride = block_table.stride(0)
        prm.page_block_size = k.shape[1]
    workspace = None
    need = lib.fa_fwd_workspace_size(ctypes.byref(prm))
    if need < 0:
        raise RuntimeError(f"fa_fwd_workspace_size failed ({need}): {_lib.strerror(int(need))}")
    if need > 0:  # fp8 expansion / split-KV partials: scratch from torch's caching allocator (callee never allocates)
        workspace = torch.empty(int(need) + 256, dtype=torch.uint8, device=q.device)
        base = (workspace.data_ptr() + 255) // 256 * 256
        prm.workspace = ctypes.c_void_p(base)
        prm.workspace_bytes = int(need)
    stream = torch.cuda.current_stream(q.device).cuda_stream
    st = lib.fa_fwd(ctypes.byref(prm), ctypes.c_void_p(stream))
    if st != 0:
        raise RuntimeError(f"fa_fwd failed ({st}): {_lib.strerror(st)}")
    if workspace is not None:
        workspace.record_stream(torch.cuda.current_stream(q.device))
    return out, lse


def launch_bwd(dout, q, k, v, out, lse, dq, dk, dv, softmax_d, *, varlen, batch, max_seqlen_q, max_seqlen_k,
               softmax_scale, causal, window_left, window_right, softcap, cu_seqlens_q=None, cu_seqlens_k=None,
               alibi_slopes=None, deterministic=False, p_dropout=0.0, rng_state=None, fa3_window=False):
    """All tensors dense (b, s, h, d) or packed (total, h, d), last stride 1, aligned(); softmax_d fp32
    (b, h, row_len) / (h, row_len).  Enqueues fa_bwd (include/fa_bwd.h) on torch's current stream."""
    lib = _lib.load()
    prm = _lib.new_bwd_params()
    prm.q, prm.k, prm.v, prm.o, prm.dout = ptr(q), ptr(k), ptr(v), ptr(out), ptr(dout)
    prm.softmax_lse, prm.softmax_d = ptr(lse), ptr(softmax_d)
    prm.dq, prm.dk, prm.dv = ptr(dq), ptr(dk), ptr(dv)
    names = (("q", q), ("k", k), ("v", v), ("o", out), ("do", dout), ("dq", dq), ("dk", dk), ("dv", dv))
    if varlen:
        for name, t in names:
            setattr(prm, f"{name}_batch_stride", 0)
            setattr(prm, f"{name}_row_stride", t.stride(0))
            setattr(prm, f"{name}_head_stride", t.stride(1))
        prm.total_q, prm.total_k = q.shape[0], k.shape[0]
        prm.h, prm.h_k, prm.d = q.shape[1], k.shape[1], q.shape[2]
    else:
        for name, t in names:
            setattr(prm, f"{name}_batch_stride", t.stride(0))
            setattr(prm, f"{name}_row_stride", t.stride(1))
            setattr(prm, f"{name}_head_stride", t.stride(2))
        prm.total_q = prm.total_k = 0
        prm.h, prm.h_k, prm.d = q.shape[2], k.shape[2], q.shape[3]
    prm.d_v = int(v.shape[-1]) if v.shape[-1] != q.shape[-1] else 0  # FA3 headdim_v (include/fa_bwd.h, ABI v12)
    prm.softmax_d_row_len = softmax_d.shape[-1]
    prm.b, prm.seqlen_q, prm.seqlen_k = int(batch), int(max_seqlen_q), int(max_seqlen_k)
    prm.dtype = _DT[q.dtype]
    prm.cu_seqlens_q, prm.cu_seqlens_k = ptr(cu_seqlens_q), ptr(cu_seqlens_k)
    prm.softmax_scale = float(softmax_scale)
    prm.softcap = float(softcap)
    prm.is_causal = int(bool(causal))
    prm.window_size_left, prm.window_size_right = int(window_left), int(window_right)
    if alibi_slopes is not None:
        prm.alibi_slopes = ptr(alibi_slopes)
        prm.alibi_slopes_batch_stride = alibi_slopes.stride(0) if alibi_slopes.dim() == 2 else 0
    prm.flags = _lib.FA_FLAG_FA3_WINDOW if fa3_window else 0
    prm.deterministic = int(bool(deterministic))
    prm.p_dropout = float(p_dropout)
    prm.rng_state = ptr(rng_state)
    stream = torch.cuda.current_stream(q.device).cuda_stream
    st = lib.fa_bwd(ctypes.byref(prm), ctypes.c_void_p(stream))
    if st != 0:
        raise RuntimeError(f"fa_bwd failed ({st}): {_lib.strerror(st)}")


def rotary_apply(src, dst, cos, sin, seqlen_offsets, interleaved, per_row_positions):
    """dst[b, i] = rotary(src[b, i]) at position seqlen_offsets[b] + (i if per_row_positions else 0); (b, s, h, d)."""
    lib = _lib.load()
    prm = _lib.FaRotaryParams()
    prm.abi_version = _lib.FA_ABI_VERSION
    prm.struct_size = ctypes.sizeof(_lib.FaRotaryParams)
    prm.src, prm.dst = ptr(src), ptr(dst)
    for name, t in (("src", src), ("dst", dst)):
        setattr(prm, f"{name}_batch_stride", t.stride(0))
        setattr(prm, f"{name}_row_stride", t.stride(1))
        setattr(prm, f"{name}_head_stride", t.stride(2))
    prm.b, prm.s, prm.h, prm.d = src.shape
    prm.dtype = _DT[src.dtype]
    prm.rotary_dim = cos.shape[1] * 2
    prm.rotary_interleaved = int(bool(interleaved))
    prm.per_row_positions = int(bool(per_row_positions))
    prm.rotary_cos, prm.rotary_sin, prm.seqlen_offsets = ptr(cos), ptr(sin), ptr(seqlen_offsets)
    stream = torch.cuda.current_stream(src.device).cuda_stream
    st = lib.fa_rotary_apply(ctypes.byref(prm), ctypes.c_void_p(stream))
    if st != 0:
        raise RuntimeError(f"fa_rotary_apply failed ({st}): {_lib.strerror(st)}")


def kvcache_append(k_new, v_new, k_cache, v_cache, cache_seqlens, cache_batch_idx=None, block_table=None,
                   rotary_cos=None, rotary_sin=None, rotary_interleaved=False, rotary_seqlens=None):
    """(b, s_new, h_k, d) rows appended in place to (b_cache, s_cache, h_k, d) caches at cache_seqlens (int32, (b,))."""
    lib = _lib.load()
    prm = _lib.FaKvcacheAppendParams()
    prm.abi_version = _lib.FA_ABI_VERSION
    prm.struct_size = ctypes.sizeof(_lib.FaKvcacheAppendParams)
    prm.k_new, prm.v_new, prm.k_cache, prm.v_cache = ptr(k_new), ptr(v_new), ptr(k_cache), ptr(v_cache)
    for name, t in (("knew", k_new), ("vnew", v_new), ("kcache", k_cache), ("vcache", v_cache)):
        setattr(prm, f"{name}_batch_stride", t.stride(0))
        setattr(prm, f"{name}_row_stride", t.stride(1))
        setattr(prm, f"{name}_head_stride", t.stride(2))
    prm.b, prm.seqlen_new, prm.h_k, prm.d = k_new.shape
    prm.seqlen_cache = k_cache.shape[1]
    if block_table is not None:
        prm.block_table = ptr(block_table)
        prm.block_table_batch_stride = block_table.stride(0)
        prm.page_block_size = k_cache.shape[1]
        prm.seqlen_cache = block_table.shape[1] * k_cache.shape[1]
    prm.cache_seqlens = ptr(cache_seqlens)
    prm.cache_batch_idx = ptr(cache_batch_idx)
    prm.dtype = _DT[k_new.dtype]
    if rotary_cos is not None:
        prm.rotary_cos, prm.rotary_sin = ptr(rotary_cos), ptr(rotary_sin)
        prm.rotary_dim = rotary_cos.shape[1] * 2
        prm.rotary_interleaved = int(bool(rotary_interleaved))
        prm.rotary_seqlens = ptr(rotary_seqlens)   # FA3 seqlens_rotary (None: the cache fill levels)
    stream = torch.cuda.current_stream(k_new.device).cuda_stream
    st = lib.fa_kvcache_append(ctypes.byref(prm), ctypes.c_void_p(stream))
    if st != 0:
        raise RuntimeError(f"fa_kvcache_append failed ({st}): {_lib.strerror(st)}")


_COMBINE_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 3}


def combine(out_partial, lse_partial, out, softmax_lse):
    """fa_fwd_combine (include/fa_fwd.h): out_partial (S, b, s, h, d) fp32, lse_partial (S, b, s, h) fp32 -- any strides
    with head-dim stride 1; out (b, s, h, d) fp32/fp16/bf16; softmax_lse fp32 indexed (b, s, h) through its strides."""
    lib = _lib.load()
    prm = _lib.FaCombineParams()
    prm.abi_version = _lib.FA_ABI_VERSION
    prm.struct_size = ctypes.sizeof(prm)
    prm.out_partial, prm.lse_partial, prm.out, prm.softmax_lse = ptr(out_partial), ptr(lse_partial), ptr(out), ptr(softmax_lse)
    for i, n in enumerate(("split", "batch", "row", "head")):
        setattr(prm, f"op_{n}_stride", out_partial.stride(i))
        setattr(prm, f"lp_{n}_stride", lse_partial.stride(i))
    for i, n in enumerate(("batch", "row", "head")):
        setattr(prm, f"o_{n}_stride", out.stride(i))
        setattr(prm, f"lse_{n}_stride", softmax_lse.stride(i))
    prm.num_splits, prm.b, prm.seqlen, prm.h, prm.d = out_partial.shape
    prm.out_dtype = _COMBINE_DT[out.dtype]
    st = lib.fa_fwd_combine(ctypes.byref(prm), ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
    if st != 0:
        raise RuntimeError(f"fa_fwd_combine failed ({st}): {_lib.strerror(st)}")
