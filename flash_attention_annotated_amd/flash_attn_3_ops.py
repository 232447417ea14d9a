"""`torch.ops.flash_attn_3.*`: the FA3 operator library (hopper/flash_api.cpp:1671-1768), defined with the reference's
schema strings so that `hopper/flash_attn_interface.py:66` (`flash_attn_3_gpu.fwd(...)`) and exported graphs bind
unchanged.  hopper/test_flash_attn.py::test_flash3_bw_compatibility (:1163-1201) pins these schemas: arguments may only
be appended with defaults.  Implementations are registered for the GPU dispatch key (HIP devices use `CUDA` in
PyTorch-ROCm) and route to flash_attn_3_cuda.fwd / flash_attn_2_cuda.bwd -> the C-ABI.
"""
import torch

from . import flash_attn_2_cuda, flash_attn_3_cuda

FWD_SCHEMA = (
    "fwd(Tensor q, Tensor k, Tensor v, Tensor(k_new!)? k_new = None, Tensor(v_new!)? v_new = None, Tensor? q_v = None, "
    "Tensor(out!)? out = None, Tensor? cu_seqlens_q = None, Tensor? cu_seqlens_k = None, Tensor? cu_seqlens_k_new = None, "
    "Tensor? seqused_q = None, Tensor? seqused_k = None, int? max_seqlen_q = None, int? max_seqlen_k = None, "
    "Tensor? page_table = None, Tensor? kv_batch_idx = None, Tensor? leftpad_k = None, Tensor? rotary_cos = None, "
    "Tensor? rotary_sin = None, Tensor? seqlens_rotary = None, Tensor? q_descale = None, Tensor? k_descale = None, "
    "Tensor? v_descale = None, float? softmax_scale = None, bool is_causal = False, int window_size_left = -1, "
    "int window_size_right = -1, int attention_chunk = 0, float softcap = 0.0, bool is_rotary_interleaved = False, "
    "Tensor? scheduler_metadata = None, int num_splits = 0, bool? pack_gqa = None, int sm_margin = 0) "
    "-> (Tensor(out!), Tensor, Tensor, Tensor)")
BWD_SCHEMA = (
    "bwd(Tensor dout, Tensor q, Tensor k, Tensor v, Tensor out, Tensor softmax_lse, Tensor(dq!)? dq = None, "
    "Tensor(dk!)? dk = None, Tensor(dv!)? dv = None, Tensor? cu_seqlens_q = None, Tensor? cu_seqlens_k = None, "
    "Tensor? seqused_q = None, Tensor? seqused_k = None, int? max_seqlen_q = None, int? max_seqlen_k = None, "
    "float? softmax_scale = None, bool is_causal = False, int window_size_left = -1, int window_size_right = -1, "
    "float softcap = 0.0, bool deterministic = False, int sm_margin = 0) "
    "-> (Tensor(dq!), Tensor(dk!), Tensor(dv!), Tensor, Tensor, Tensor, Tensor, Tensor)")
COMBINE_SCHEMA = ("fwd_combine(Tensor out_partial, Tensor lse_partial, Tensor(out!)? out = None, ScalarType? out_dtype = None) "
                  "-> (Tensor(out!), Tensor)")
METADATA_SCHEMA = (
    "get_scheduler_metadata(int batch_size, int max_seqlen_q, int max_seqlen_k, int num_heads, int num_heads_k, int headdim, "
    "int headdim_v, ScalarType qkv_dtype, Tensor seqused_k, Tensor? cu_seqlens_q = None, Tensor? cu_seqlens_k = None, "
    "Tensor? cu_seqlens_k_new = None, Tensor? seqused_q = None, Tensor? leftpad_k = None, int? page_size = None, "
    "int max_seqlen_k_new = 0, bool is_causal = False, int window_size_left = -1, int window_size_right = -1, "
    "int attention_chunk = 0, bool has_softcap = False, int num_splits = 0, bool? pack_gqa = None, int sm_margin = 0) -> Tensor")

_lib = torch.library.Library("flash_attn_3", "DEF")
for _schema in (FWD_SCHEMA, BWD_SCHEMA, COMBINE_SCHEMA, METADATA_SCHEMA):
    _lib.define(_schema)


def _fwd(q, k, v, k_new=None, v_new=None, q_v=None, out=None, cu_seqlens_q=None, cu_seqlens_k=None, cu_seqlens_k_new=None,
         seqused_q=None, seqused_k=None, max_seqlen_q=None, max_seqlen_k=None, page_table=None, kv_batch_idx=None,
         leftpad_k=None, rotary_cos=None, rotary_sin=None, seqlens_rotary=None, q_descale=None, k_descale=None,
         v_descale=None, softmax_scale=None, is_causal=False, window_size_left=-1, window_size_right=-1,
         attention_chunk=0, softcap=0.0, is_rotary_interleaved=False, scheduler_metadata=None, num_splits=0,
         pack_gqa=None, sm_margin=0):
    o, lse, _, _ = flash_attn_3_cuda.fwd(
        q, k, v, k_new, v_new, q_v, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new, seqused_q, seqused_k, max_seqlen_q,
        max_seqlen_k, page_table, kv_batch_idx, leftpad_k, rotary_cos, rotary_sin, seqlens_rotary, q_descale, k_descale,
        v_descale, softmax_scale, is_causal, window_size_left, window_size_right, attention_chunk, softcap,
        is_rotary_interleaved, scheduler_metadata, num_splits, pack_gqa, sm_margin)
    # out_accum / softmax_lse_accum: empty when the result was not produced by the split path (hopper/flash_api.cpp:1196)
    return o, lse, torch.empty(0, dtype=torch.float32, device=q.device), torch.empty(0, dtype=torch.float32, device=q.device)


def _bwd(dout, q, k, v, out, softmax_lse, dq=None, dk=None, dv=None, cu_seqlens_q=None, cu_seqlens_k=None, seqused_q=None,
         seqused_k=None, max_seqlen_q=None, max_seqlen_k=None, softmax_scale=None, is_causal=False, window_size_left=-1,
         window_size_right=-1, softcap=0.0, deterministic=False, sm_margin=0):
    """mha_bwd, hopper/flash_api.cpp:1259-1570, on the FA2-shaped backward of this build (16-bit types)."""
    if seqused_q is not None or seqused_k is not None:
        raise RuntimeError("This flash attention build does not support seqused_q / seqused_k in the backward.")
    if v.shape[-1] != q.shape[-1]:
        return _bwd_own_dv(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                           softmax_scale, is_causal, window_size_left, window_size_right, softcap, deterministic)
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    # window normalisation of the FA3 entry points (hopper/flash_api.cpp:1360-1361, as :796-797 of the forward): a side that
    # cannot mask anything becomes -1 = unbounded and stays unbounded (FA_FLAG_FA3_WINDOW)
    sq_max = int(max_seqlen_q) if cu_seqlens_q is not None else q.shape[1]
    sk_max = int(max_seqlen_k) if cu_seqlens_q is not None else k.shape[1]
    if window_size_left >= sk_max - 1:
        window_size_left = -1
    if window_size_right >= sq_max - 1:
        window_size_right = -1
    if is_causal:
        window_size_right = 0
    with flash_attn_2_cuda.fa3_window_rule():
        if cu_seqlens_q is not None:
            dq, dk, dv, sd = flash_attn_2_cuda.varlen_bwd(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k,
                                                         None, int(max_seqlen_q), int(max_seqlen_k), 0.0, softmax_scale, False,
                                                         is_causal, window_size_left, window_size_right, softcap,
                                                         deterministic, None, None)
        else:
            dq, dk, dv, sd = flash_attn_2_cuda.bwd(dout, q, k, v, out, softmax_lse, dq, dk, dv, None, 0.0, softmax_scale,
                                                   is_causal, window_size_left, window_size_right, softcap, deterministic,
                                                   None, None)
    e = torch.empty(0, dtype=torch.float32, device=q.device)  # softmax_lse_log2, dq_accum, dk_accum, dv_accum: none here
    return dq, dk, dv, sd, e, e.clone(), e.clone(), e.clone()


def _bwd_own_dv(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                softmax_scale, is_causal, window_size_left, window_size_right, softcap, deterministic):
    """mha_bwd with a V head dim of its own (hopper/flash_api.cpp:1345-1369, 1399-1412, 1462-1464: v / out / dout / dv carry
    head_size_v, the kernels round both dims to the larger).  Built for the wide tile: max(d, dv) in (128, 256]."""
    from . import _dispatch

    def check(cond, msg):
        if not cond:
            raise RuntimeError(msg)
    d, d_v = q.shape[-1], v.shape[-1]
    check(q.dtype in (torch.float16, torch.bfloat16), "FlashAttention only support fp16 and bf16 data type")
    for t, n in ((k, "key"), (v, "value"), (out, "out"), (dout, "dout")):
        check(t.dtype == q.dtype, f"query and {n} must have the same dtype")
    check(d % 8 == 0, "head_size should be a multiple of 8")
    check(d_v % 8 == 0, "head_size_v should be a multiple of 8")
    check(max(d, d_v) <= 256, "FlashAttention backward only supports head dimension at most 256")
    check(max(d, d_v) > 128, "This flash attention build supports a V headdim different from the Q/K headdim in the backward "
                             "only when the larger of the two is above 128.")
    check(k.shape[-1] == d and tuple(v.shape[:-1]) == tuple(k.shape[:-1]), "k / v shapes do not match")
    check(tuple(out.shape) == tuple(q.shape[:-1]) + (d_v,) and tuple(dout.shape) == tuple(out.shape), "out / dout must be (..., head_size_v)")
    varlen = cu_seqlens_q is not None
    if softmax_scale is None:
        softmax_scale = d ** (-0.5)
    sq_max = int(max_seqlen_q) if varlen else q.shape[1]
    sk_max = int(max_seqlen_k) if varlen else k.shape[1]
    if window_size_left >= sk_max - 1:
        window_size_left = -1
    if window_size_right >= sq_max - 1:
        window_size_right = -1
    if is_causal:
        window_size_right = 0

    def grad(given, like, name):
        if given is None:
            return torch.empty_like(like)
        check(given.dtype == like.dtype and given.is_cuda and given.stride(-1) == 1 and tuple(given.shape) == tuple(like.shape),
              f"{name} must have the dtype, device and shape of its tensor")
        return given
    dq, dk, dv = grad(dq, q, "dq"), grad(dk, k, "dk"), grad(dv, v, "dv")
    batch = cu_seqlens_q.numel() - 1 if varlen else q.shape[0]
    h = q.shape[-2]
    with torch.cuda.device(q.device):
        rows = q.shape[0] if varlen else (sq_max + 127) // 128 * 128
        softmax_d = torch.empty(((h, rows + 128 * batch) if varlen else (batch, h, rows)), dtype=torch.float32, device=q.device)
        if q.numel() > 0 and k.numel() > 0:
            ins = [x if _dispatch.aligned(x) else x.contiguous() for x in (dout, q, k, v, out)]
            outs = [x if _dispatch.aligned(x) else torch.empty_like(x, memory_format=torch.contiguous_format) for x in (dq, dk, dv)]
            lse = softmax_lse if softmax_lse.is_contiguous() else softmax_lse.contiguous()
            _dispatch.launch_bwd(*ins, lse, *outs, softmax_d, varlen=varlen, batch=batch, max_seqlen_q=sq_max, max_seqlen_k=sk_max,
                                 softmax_scale=softmax_scale, causal=is_causal, window_left=window_size_left,
                                 window_right=window_size_right, softcap=softcap, cu_seqlens_q=cu_seqlens_q,
                                 cu_seqlens_k=cu_seqlens_k, deterministic=deterministic, fa3_window=True)
            for dst, src in zip((dq, dk, dv), outs):
                if dst is not src:
                    dst.copy_(src)
        else:
            dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_()
    e = torch.empty(0, dtype=torch.float32, device=q.device)
    return dq, dk, dv, softmax_d, e, e.clone(), e.clone(), e.clone()


def _fwd_combine(out_partial, lse_partial, out=None, out_dtype=None):
    """mha_combine, hopper/flash_api.cpp:1569-1670: merge caller-held split-KV partials.  out_partial
    (num_splits, b, seqlen, h, d) fp32, lse_partial (num_splits, b, seqlen, h) fp32 -> (out, softmax_lse (b, seqlen, h))."""
    def check(cond, msg):
        if not cond:
            raise RuntimeError(msg)
    check(out_partial.dtype == torch.float32, "Attention combine function only support fp32 data type")
    check(lse_partial.dtype == torch.float32, "Attention combine function only support fp32 data type")
    check(out_partial.is_cuda and lse_partial.is_cuda, "out_partial must be on CUDA")
    check(out_partial.stride(-1) == 1, "Input tensor must have contiguous last dimension")
    check(lse_partial.stride(-2) == 1, "LSE tensor must be contiguous in the seqlen dimension")
    check(out_partial.dim() == 5, "out_partial must have shape (num_splits, batch_size, seqlen, num_heads, head_size)")
    num_splits, batch_size, seqlen, num_heads, head_size = out_partial.shape
    check(num_splits <= 256, "FlashAttention combine only supports num_splits at most 256")
    check(tuple(lse_partial.shape) == (num_splits, batch_size, seqlen, num_heads),
          "lse_partial must have shape (num_splits, batch_size, seqlen, num_heads)")
    out_type = out_dtype if out_dtype is not None else out_partial.dtype
    check(out_type in (torch.float32, torch.float16, torch.bfloat16), "Output type must be FP32, FP16 or BF16")
    if out is not None:
        check(out.dtype == out_type, "out must have the requested output type")
        check(out.is_cuda, "out must be on CUDA")
        check(out.stride(-1) == 1, "Output tensor must have contiguous last dimension")
        check(tuple(out.shape) == (batch_size, seqlen, num_heads, head_size),
              "out must have shape (batch_size, seqlen, num_heads, head_size)")
    else:
        out = torch.empty((batch_size, seqlen, num_heads, head_size), dtype=out_type, device=out_partial.device)
    with torch.cuda.device(out_partial.device):
        softmax_lse = torch.empty((batch_size, num_heads, seqlen), dtype=torch.float32,
                                  device=out_partial.device).transpose(1, 2)  # (:1632)
        if seqlen > 0 and batch_size > 0:
            from . import _dispatch
            _dispatch.combine(out_partial, lse_partial, out, softmax_lse)
    return out, softmax_lse


def _get_scheduler_metadata(batch_size, max_seqlen_q, max_seqlen_k, num_heads, num_heads_k, headdim, headdim_v, qkv_dtype,
                            seqused_k, cu_seqlens_q=None, cu_seqlens_k=None, cu_seqlens_k_new=None, seqused_q=None,
                            leftpad_k=None, page_size=None, max_seqlen_k_new=0, is_causal=False, window_size_left=-1,
                            window_size_right=-1, attention_chunk=0, has_softcap=False, num_splits=0, pack_gqa=None,
                            sm_margin=0):
    """The tile schedule of this build is computed inside the kernel (decode_tile): the metadata tensor is opaque to
    callers and empty here; `fwd` ignores it (hopper/flash_api.cpp:520-669 builds a semaphore + per-batch split table)."""
    return torch.zeros(1, dtype=torch.int32, device=seqused_k.device)


_lib.impl("fwd", _fwd, "CUDA")
_lib.impl("bwd", _bwd, "CUDA")
_lib.impl("fwd_combine", _fwd_combine, "CUDA")
_lib.impl("get_scheduler_metadata", _get_scheduler_metadata, "CUDA")
