// torch_binding.cpp — compiled `flash_attn_2_cuda` surface over the C-ABI (include/fa_fwd.h, include/fa_bwd.h).
//
// The pybind module of csrc/flash_attn/flash_api.cpp:1478-1485: fwd / varlen_fwd / bwd / varlen_bwd / fwd_kvcache with the
// reference's positional argument lists, doing the host work of mha_fwd (:350-512), mha_varlen_fwd (:514-755), mha_bwd
// (:767-971), mha_varlen_bwd (:973-1200) and mha_fwd_kvcache (:1202-1476) -- TORCH_CHECKs with the reference's texts, output /
// LSE / softmax_d allocation, the params struct -- and enqueueing the gfx950 kernels on torch's current stream.  Host code
// only: built by plain g++ against the torch headers (no hipify, no device code here), linked to libfa_fwd_gfx950.so.
// The Python module flash_attn_2_cuda.py states the same logic and stays as the fallback binding (ctypes).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cmath>
#include <limits>
#include <vector>

#include "fa_bwd.h"
#include "fa_fwd.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<at::Tensor>;

thread_local bool g_fa3_window = false;  // FA3 window rule for bwd / varlen_bwd (flash_attn_3_ops._bwd)

#define CHECK_DEVICE(x, name) TORCH_CHECK((x).is_cuda(), name " must be on CUDA")
#define CHECK_SHAPE(x, name, ...) \
    TORCH_CHECK((x).sizes() == c10::IntArrayRef({__VA_ARGS__}), name " must have shape (" #__VA_ARGS__ ")")
#define CHECK_LAST_CONTIGUOUS(x, msg) TORCH_CHECK((x).stride(-1) == 1, msg)

int dtype_code(const Tensor &t) {
    if (t.scalar_type() == at::kHalf) return FA_DTYPE_FP16;
    if (t.scalar_type() == at::kBFloat16) return FA_DTYPE_BF16;
    TORCH_CHECK(false, "FlashAttention only support fp16 and bf16 data type");
    return -1;
}

// the kernels move 16-byte vectors: bases and the non-unit strides must keep rows aligned (views that do not are copied)
bool aligned(const Tensor &t) {
    if (reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 != 0) return false;
    for (int64_t i = 0; i + 1 < t.dim(); ++i)
        if (t.stride(i) % 8 != 0) return false;
    return true;
}
Tensor aligned_or_copy(const Tensor &t) { return aligned(t) ? t : t.contiguous(); }

void *ptr(const OptTensor &t) { return t.has_value() ? t->data_ptr() : nullptr; }
void *ptr(const Tensor &t) { return t.data_ptr(); }

int64_t round128(int64_t x) { return (x + 127) / 128 * 128; }

// kBlockN of the reference's forward for this head dim (flash_attn/flash_attn_interface.py:23-46, a device that is neither
// sm8x nor sm90): the key-block width behind the running maxima of S_dmask
int sdmask_block_n(int64_t head_dim, bool is_dropout, bool /*is_causal*/) {
    if (head_dim <= 32) return 128;
    if (head_dim <= 64) return is_dropout ? 64 : 128;
    if (head_dim <= 96) return 64;
    if (head_dim <= 128) return is_dropout ? 32 : 64;
    return 64;
}

void check_dropout(double p_dropout, bool return_softmax) {
    TORCH_CHECK(p_dropout >= 0.0 && p_dropout < 1.0, "p_dropout must be in [0, 1)");
    if (return_softmax) TORCH_CHECK(p_dropout > 0.0, "return_softmax is only supported when p_dropout > 0.0");
}

// (seed, offset) of this call, drawn ON THE DEVICE from gen_ / the default generator (role of philox_cuda_state, :486-493)
Tensor dropout_state(double p_dropout, const c10::optional<at::Generator> &gen, const Tensor &like) {
    auto opts = like.options().dtype(at::kLong);
    if (p_dropout <= 0.0) return at::zeros({2}, opts);
    return at::randint(-(int64_t(1) << 62), int64_t(1) << 62, {2}, gen, opts);
}

OptTensor check_alibi(const OptTensor &a, int64_t batch_size, int64_t num_heads) {
    if (!a.has_value()) return a;
    TORCH_CHECK(a->scalar_type() == at::kFloat, "ALiBi slopes must have dtype fp32");
    CHECK_DEVICE(*a, "alibi_slopes");
    TORCH_CHECK(a->stride(-1) == 1, "ALiBi slopes tensor must have contiguous last dimension");
    TORCH_CHECK(a->sizes() == c10::IntArrayRef({num_heads}) || a->sizes() == c10::IntArrayRef({batch_size, num_heads}),
                "alibi_slopes must have shape (num_heads) or (batch_size, num_heads)");
    return a;
}

void check_leftpad(const OptTensor &lp, int64_t batch_size, bool paged) {
    if (!lp.has_value()) return;
    TORCH_CHECK(!paged, "We don't support Paged KV and leftpad_k running at the same time yet");
    TORCH_CHECK(lp->scalar_type() == at::kInt, "leftpad_k must have dtype int32");
    CHECK_DEVICE(*lp, "leftpad_k");
    TORCH_CHECK(lp->is_contiguous(), "leftpad_k must be contiguous");
    CHECK_SHAPE(*lp, "leftpad_k", batch_size);
}

// returns (page_block_size, max_num_blocks_per_seq)
std::pair<int64_t, int64_t> check_block_table(const Tensor &bt, const Tensor &kcache, int64_t batch_size, int64_t page_multiple) {
    CHECK_DEVICE(bt, "block_table");
    TORCH_CHECK(bt.scalar_type() == at::kInt, "block_table must have dtype torch.int32");
    TORCH_CHECK(bt.stride(-1) == 1, "block_table must have contiguous last dimension");
    TORCH_CHECK(kcache.dim() == 4, "paged k/v must have shape (num_blocks, page_block_size, num_heads_k, head_size)");
    const int64_t page = kcache.size(1);
    TORCH_CHECK(page % page_multiple == 0, "Paged KV cache block size must be divisible by ", page_multiple);
    TORCH_CHECK(bt.dim() == 2 && bt.size(0) == batch_size, "block_table must have shape (batch_size, max_num_blocks_per_seq)");
    return {page, bt.size(1)};
}

hipStream_t current_stream(const Tensor &t) {
    return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

struct FwdArgs {
    bool varlen = false;
    int64_t batch = 0, max_seqlen_q = 0, max_seqlen_k = 0;
    double softmax_scale = 1.0, softcap = 0.0, p_dropout = 0.0;
    bool causal = false;
    int64_t window_left = -1, window_right = -1;
    OptTensor cu_seqlens_q, cu_seqlens_k, seqused_k, alibi, kv_batch_idx, block_table, leftpad_k, rng_state, s_dmask;
    int num_splits = 1, s_dmask_block_n = 0;
};

// torch tensors -> fa_fwd_params -> fa_fwd on torch's current stream (q/k/v/out: last stride 1, aligned())
void launch_fwd(const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out, const Tensor &lse, const FwdArgs &a) {
    fa_fwd_params p{};
    p.abi_version = FA_ABI_VERSION;
    p.struct_size = sizeof(fa_fwd_params);
    p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr(); p.o = out.data_ptr();
    p.softmax_lse = static_cast<float *>(lse.data_ptr());
    const bool paged = a.block_table.has_value();
    if (a.varlen) {
        p.q_row_stride = q.stride(0); p.q_head_stride = q.stride(1);
        p.o_row_stride = out.stride(0); p.o_head_stride = out.stride(1);
        if (paged) {  // k, v: (num_blocks, page_block_size, h_k, d)
            p.k_batch_stride = k.stride(0); p.k_row_stride = k.stride(1); p.k_head_stride = k.stride(2);
            p.v_batch_stride = v.stride(0); p.v_row_stride = v.stride(1); p.v_head_stride = v.stride(2);
        } else {
            p.k_row_stride = k.stride(0); p.k_head_stride = k.stride(1);
            p.v_row_stride = v.stride(0); p.v_head_stride = v.stride(1);
        }
        p.total_q = (int32_t)q.size(0);
        p.total_k = paged ? 0 : (int32_t)k.size(0);
        p.h = (int32_t)q.size(1); p.h_k = (int32_t)k.size(-2); p.d = (int32_t)q.size(2);
    } else {
        p.q_batch_stride = q.stride(0); p.q_row_stride = q.stride(1); p.q_head_stride = q.stride(2);
        p.k_batch_stride = k.stride(0); p.k_row_stride = k.stride(1); p.k_head_stride = k.stride(2);
        p.v_batch_stride = v.stride(0); p.v_row_stride = v.stride(1); p.v_head_stride = v.stride(2);
        p.o_batch_stride = out.stride(0); p.o_row_stride = out.stride(1); p.o_head_stride = out.stride(2);
        p.h = (int32_t)q.size(2); p.h_k = (int32_t)k.size(2); p.d = (int32_t)q.size(3);
    }
    p.b = (int32_t)a.batch; p.seqlen_q = (int32_t)a.max_seqlen_q; p.seqlen_k = (int32_t)a.max_seqlen_k;
    p.dtype = dtype_code(q);
    p.cu_seqlens_q = static_cast<const int32_t *>(ptr(a.cu_seqlens_q));
    p.cu_seqlens_k = static_cast<const int32_t *>(ptr(a.cu_seqlens_k));
    p.seqused_k = static_cast<const int32_t *>(ptr(a.seqused_k));
    p.softmax_scale = (float)a.softmax_scale;
    p.softcap = (float)a.softcap;
    p.is_causal = a.causal ? 1 : 0;
    p.window_size_left = (int32_t)a.window_left; p.window_size_right = (int32_t)a.window_right;
    if (a.alibi.has_value()) {  // (h) or (b, h) fp32
        p.alibi_slopes = static_cast<const float *>(a.alibi->data_ptr());
        p.alibi_slopes_batch_stride = a.alibi->dim() == 2 ? a.alibi->stride(0) : 0;
    }
    p.kv_batch_idx = static_cast<const int32_t *>(ptr(a.kv_batch_idx));
    p.leftpad_k = static_cast<const int32_t *>(ptr(a.leftpad_k));
    p.p_dropout = (float)a.p_dropout;
    p.rng_state = static_cast<const uint64_t *>(ptr(a.rng_state));
    p.s_dmask = static_cast<uint8_t *>(ptr(a.s_dmask));
    p.flags = 0;
    if (a.s_dmask.has_value() && a.s_dmask_block_n > 0) {
        p.flags |= FA_FLAG_SDMASK_SIGNED;
        p.s_dmask_rows = (int32_t)a.s_dmask->size(-2);
        p.s_dmask_cols = (int32_t)a.s_dmask->size(-1);
        p.s_dmask_block_n = a.s_dmask_block_n;
    }
    p.num_splits = a.num_splits;
    if (paged) {
        p.block_table = static_cast<const int32_t *>(a.block_table->data_ptr());
        p.block_table_batch_stride = a.block_table->stride(0);
        p.page_block_size = (int32_t)k.size(1);
    }
    Tensor workspace;
    const int64_t need = fa_fwd_workspace_size(&p);
    TORCH_CHECK(need >= 0, "fa_fwd_workspace_size failed (", need, "): ", fa_strerror((int)need));
    if (need > 0) {  // split-KV partials: scratch from torch's caching allocator (the callee never allocates)
        workspace = at::empty({need + 256}, q.options().dtype(at::kByte));
        const uintptr_t base = (reinterpret_cast<uintptr_t>(workspace.data_ptr()) + 255) / 256 * 256;
        p.workspace = reinterpret_cast<void *>(base);
        p.workspace_bytes = (uint64_t)need;
    }
    const int st = fa_fwd(&p, current_stream(q));
    TORCH_CHECK(st == 0, "fa_fwd failed (", st, "): ", fa_strerror(st));
}

std::vector<Tensor> mha_fwd(Tensor &q, const Tensor &k, const Tensor &v, OptTensor &out_, OptTensor &alibi_slopes_,
                            const double p_dropout, const double softmax_scale, bool is_causal, int64_t window_size_left,
                            int64_t window_size_right, const double softcap, const bool return_softmax,
                            c10::optional<at::Generator> gen_) {
    const auto q_dtype = q.scalar_type();
    TORCH_CHECK(q_dtype == at::kHalf || q_dtype == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
    TORCH_CHECK(k.scalar_type() == q_dtype, "query and key must have the same dtype");
    TORCH_CHECK(v.scalar_type() == q_dtype, "query and value must have the same dtype");
    CHECK_DEVICE(q, "q"); CHECK_DEVICE(k, "k"); CHECK_DEVICE(v, "v");
    CHECK_LAST_CONTIGUOUS(q, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(k, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(v, "Input tensor must have contiguous last dimension");
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must have 4 dimensions");
    const int64_t batch_size = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_size = q.size(3);
    const int64_t seqlen_k = k.size(1), num_heads_k = k.size(2);
    TORCH_CHECK(batch_size > 0, "batch size must be positive");
    TORCH_CHECK(head_size <= 256, "FlashAttention forward only supports head dimension at most 256");
    TORCH_CHECK(head_size % 8 == 0, "query, key, value, and out_ must have a head_size that is a multiple of 8");
    TORCH_CHECK(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query");
    if (softcap > 0.0) TORCH_CHECK(p_dropout == 0.0, "Softcapping does not support dropout for now");
    check_dropout(p_dropout, return_softmax);
    const OptTensor alibi = check_alibi(alibi_slopes_, batch_size, num_heads);
    if (seqlen_q == 1 && !alibi_slopes_.has_value()) is_causal = false;  // causal=true is the same as causal=false here (:402)
    CHECK_SHAPE(q, "q", batch_size, seqlen_q, num_heads, head_size);
    CHECK_SHAPE(k, "k", batch_size, seqlen_k, num_heads_k, head_size);
    CHECK_SHAPE(v, "v", batch_size, seqlen_k, num_heads_k, head_size);
    Tensor out;
    if (out_.has_value()) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q_dtype, "Output must have the same dtype as inputs");
        CHECK_DEVICE(out, "out");
        TORCH_CHECK(out.stride(-1) == 1, "Output tensor must have contiguous last dimension");
        CHECK_SHAPE(out, "out", batch_size, seqlen_q, num_heads, head_size);
    } else {
        out = at::empty_like(q);
    }
    c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
    auto opts = q.options();
    Tensor softmax_lse = at::empty({batch_size, num_heads, seqlen_q}, opts.dtype(at::kFloat));
    // return_softmax: S_dmask (b, h, seqlen_q r128, seqlen_k r128) in the input dtype (csrc/flash_attn/flash_api.cpp:436-449)
    Tensor p = return_softmax ? at::zeros({batch_size, num_heads, round128(seqlen_q), round128(seqlen_k)}, opts) : at::empty({0}, opts);
    Tensor rng_state = dropout_state(p_dropout, gen_, q);
    if (seqlen_k > 0 && seqlen_q > 0) {
        const Tensor qc = aligned_or_copy(q), kc = aligned_or_copy(k), vc = aligned_or_copy(v);
        Tensor oc = aligned(out) ? out : at::empty_like(qc);
        FwdArgs a;
        a.batch = batch_size; a.max_seqlen_q = seqlen_q; a.max_seqlen_k = seqlen_k;
        a.softmax_scale = softmax_scale; a.causal = is_causal; a.window_left = window_size_left; a.window_right = window_size_right;
        a.softcap = softcap; a.alibi = alibi; a.p_dropout = p_dropout;
        if (p_dropout > 0) a.rng_state = rng_state;
        if (return_softmax) a.s_dmask = p;
        a.s_dmask_block_n = sdmask_block_n(head_size, p_dropout > 0, is_causal);
        a.num_splits = p_dropout == 0.0 ? 0 : 1;  // the split heuristic runs whenever there is no dropout (:453-456)
        launch_fwd(qc, kc, vc, oc, softmax_lse, a);
        if (!oc.is_same(out)) out.copy_(oc);
    } else if (seqlen_q > 0) {
        out.zero_();  // seqlen_k == 0: empty attention (:499-504)
        softmax_lse.fill_(std::numeric_limits<float>::infinity());
    }
    return {out, softmax_lse, p, rng_state};
}

std::vector<Tensor> mha_varlen_fwd(Tensor &q, const Tensor &k, const Tensor &v, OptTensor &out_, const Tensor &cu_seqlens_q,
                                   const Tensor &cu_seqlens_k, OptTensor &seqused_k, OptTensor &leftpad_k_,
                                   OptTensor &block_table_, OptTensor &alibi_slopes_, int64_t max_seqlen_q,
                                   const int64_t max_seqlen_k, const double p_dropout, const double softmax_scale,
                                   const bool zero_tensors, bool is_causal, int64_t window_size_left, int64_t window_size_right,
                                   const double softcap, const bool return_softmax, c10::optional<at::Generator> gen_) {
    const auto q_dtype = q.scalar_type();
    TORCH_CHECK(q_dtype == at::kHalf || q_dtype == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
    TORCH_CHECK(k.scalar_type() == q_dtype, "query and key must have the same dtype");
    TORCH_CHECK(v.scalar_type() == q_dtype, "query and value must have the same dtype");
    TORCH_CHECK(cu_seqlens_q.scalar_type() == at::kInt, "cu_seqlens_q must have dtype int32");
    TORCH_CHECK(cu_seqlens_k.scalar_type() == at::kInt, "cu_seqlens_k must have dtype int32");
    CHECK_DEVICE(q, "q"); CHECK_DEVICE(k, "k"); CHECK_DEVICE(v, "v");
    CHECK_DEVICE(cu_seqlens_q, "cu_seqlens_q"); CHECK_DEVICE(cu_seqlens_k, "cu_seqlens_k");
    const bool paged = block_table_.has_value();
    CHECK_LAST_CONTIGUOUS(q, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(k, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(v, "Input tensor must have contiguous last dimension");
    TORCH_CHECK(cu_seqlens_q.is_contiguous(), "cu_seqlens_q must be contiguous");
    TORCH_CHECK(cu_seqlens_k.is_contiguous(), "cu_seqlens_k must be contiguous");
    TORCH_CHECK(q.dim() == 3, "q must have shape (total_q, num_heads, head_size)");
    const int64_t total_q = q.size(0), num_heads = q.size(1), head_size = q.size(2);
    const int64_t batch_size = cu_seqlens_q.numel() - 1;
    int64_t total_k = 0, num_heads_k = 0;
    if (paged) {  // k, v: (num_blocks, page_block_size, h_k, d), rows found through block_table (:554-560, :608-612)
        check_block_table(*block_table_, k, batch_size, 256);
        num_heads_k = k.size(2);
    } else {
        TORCH_CHECK(k.dim() == 3, "k must have shape (total_k, num_heads_k, head_size)");
        total_k = k.size(0); num_heads_k = k.size(1);
    }
    TORCH_CHECK(batch_size > 0, "batch size must be positive");
    TORCH_CHECK(head_size <= 256, "FlashAttention forward only supports head dimension at most 256");
    TORCH_CHECK(head_size % 8 == 0, "query, key, value, and out_ must have a head_size that is a multiple of 8");
    TORCH_CHECK(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query");
    if (softcap > 0.0) TORCH_CHECK(p_dropout == 0.0, "Softcapping does not support dropout for now");
    check_dropout(p_dropout, return_softmax);
    if (p_dropout > 0.0)
        TORCH_CHECK(!paged && !leftpad_k_.has_value(), "dropout is not supported with a paged or left-padded KV cache");
    const OptTensor alibi = check_alibi(alibi_slopes_, batch_size, num_heads);
    if (max_seqlen_q == 1 && !alibi_slopes_.has_value()) is_causal = false;  // (:590)
    CHECK_SHAPE(q, "q", total_q, num_heads, head_size);
    if (paged) {
        CHECK_SHAPE(k, "k", k.size(0), k.size(1), num_heads_k, head_size);
        CHECK_SHAPE(v, "v", k.size(0), k.size(1), num_heads_k, head_size);
    } else {
        CHECK_SHAPE(k, "k", total_k, num_heads_k, head_size);
        CHECK_SHAPE(v, "v", total_k, num_heads_k, head_size);
    }
    CHECK_SHAPE(cu_seqlens_q, "cu_seqlens_q", batch_size + 1);
    CHECK_SHAPE(cu_seqlens_k, "cu_seqlens_k", batch_size + 1);
    check_leftpad(leftpad_k_, batch_size, paged);
    if (seqused_k.has_value()) {
        TORCH_CHECK(seqused_k->scalar_type() == at::kInt, "seqused_k must have dtype int32");
        CHECK_DEVICE(*seqused_k, "seqused_k");
        TORCH_CHECK(seqused_k->is_contiguous(), "seqused_k must be contiguous");
        CHECK_SHAPE(*seqused_k, "seqused_k", batch_size);
    }
    Tensor out;
    if (out_.has_value()) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q_dtype, "Output must have the same dtype as inputs");
        CHECK_DEVICE(out, "out");
        TORCH_CHECK(out.stride(-1) == 1, "Output tensor must have contiguous last dimension");
        CHECK_SHAPE(out, "out", total_q, num_heads, head_size);
    } else {
        out = at::empty_like(q);
    }
    c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
    auto opts = q.options();
    Tensor softmax_lse = at::empty({num_heads, total_q}, opts.dtype(at::kFloat));
    Tensor p = return_softmax ? at::zeros({batch_size, num_heads, round128(max_seqlen_q), round128(max_seqlen_k)}, opts)
                              : at::empty({0}, opts);
    Tensor rng_state = dropout_state(p_dropout, gen_, q);
    if (zero_tensors) {
        out.zero_();
        softmax_lse.fill_(-std::numeric_limits<float>::infinity());
    }
    if (max_seqlen_k > 0 && total_q > 0 && max_seqlen_q > 0) {
        const Tensor qc = aligned_or_copy(q), kc = aligned_or_copy(k), vc = aligned_or_copy(v);
        Tensor oc = aligned(out) ? out : at::empty_like(qc);
        FwdArgs a;
        a.varlen = true; a.batch = batch_size; a.max_seqlen_q = max_seqlen_q; a.max_seqlen_k = max_seqlen_k;
        a.softmax_scale = softmax_scale; a.causal = is_causal; a.window_left = window_size_left; a.window_right = window_size_right;
        a.softcap = softcap; a.cu_seqlens_q = cu_seqlens_q; a.cu_seqlens_k = cu_seqlens_k; a.seqused_k = seqused_k;
        a.alibi = alibi; a.block_table = block_table_; a.leftpad_k = leftpad_k_; a.p_dropout = p_dropout;
        if (p_dropout > 0) a.rng_state = rng_state;
        if (return_softmax) a.s_dmask = p;
        a.s_dmask_block_n = sdmask_block_n(head_size, p_dropout > 0, is_causal);
        launch_fwd(qc, kc, vc, oc, softmax_lse, a);
        if (!oc.is_same(out)) out.copy_(oc);
    } else if (total_q > 0) {
        out.zero_();
        softmax_lse.fill_(std::numeric_limits<float>::infinity());
    }
    return {out, softmax_lse, p, rng_state};
}

Tensor grad_out(const OptTensor &given, const Tensor &like, const char *name, c10::IntArrayRef shape) {
    if (!given.has_value()) return at::empty_like(like);
    TORCH_CHECK(given->scalar_type() == like.scalar_type(), name, " must have the same dtype as q");
    TORCH_CHECK(given->is_cuda(), name, " must be on CUDA");
    TORCH_CHECK(given->stride(-1) == 1, name, " must have contiguous last dimension");
    TORCH_CHECK(given->sizes() == shape, name, " must have shape ", shape);
    return given.value();
}

struct BwdArgs {
    bool varlen = false;
    int64_t batch = 0, max_seqlen_q = 0, max_seqlen_k = 0;
    double softmax_scale = 1.0, softcap = 0.0, p_dropout = 0.0;
    bool causal = false, deterministic = false;
    int64_t window_left = -1, window_right = -1;
    OptTensor cu_seqlens_q, cu_seqlens_k, alibi, rng_state;
};

void launch_bwd(const Tensor &dout, const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out, const Tensor &lse,
                const Tensor &dq, const Tensor &dk, const Tensor &dv, const Tensor &softmax_d, const BwdArgs &a) {
    fa_bwd_params p{};
    p.abi_version = FA_ABI_VERSION;
    p.struct_size = sizeof(fa_bwd_params);
    p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr(); p.o = out.data_ptr(); p.dout = dout.data_ptr();
    p.softmax_lse = static_cast<const float *>(lse.data_ptr());
    p.softmax_d = static_cast<float *>(softmax_d.data_ptr());
    p.dq = dq.data_ptr(); p.dk = dk.data_ptr(); p.dv = dv.data_ptr();
    const int o = a.varlen ? 0 : 1;  // index of the row stride
#define FA_SET_STRIDES(name, t)                                                                   \
    p.name##_batch_stride = a.varlen ? 0 : (t).stride(0);                                         \
    p.name##_row_stride = (t).stride(o);                                                          \
    p.name##_head_stride = (t).stride(o + 1);
    FA_SET_STRIDES(q, q) FA_SET_STRIDES(k, k) FA_SET_STRIDES(v, v) FA_SET_STRIDES(o, out) FA_SET_STRIDES(do, dout)
    FA_SET_STRIDES(dq, dq) FA_SET_STRIDES(dk, dk) FA_SET_STRIDES(dv, dv)
#undef FA_SET_STRIDES
    if (a.varlen) {
        p.total_q = (int32_t)q.size(0); p.total_k = (int32_t)k.size(0);
        p.h = (int32_t)q.size(1); p.h_k = (int32_t)k.size(1); p.d = (int32_t)q.size(2);
    } else {
        p.h = (int32_t)q.size(2); p.h_k = (int32_t)k.size(2); p.d = (int32_t)q.size(3);
    }
    p.softmax_d_row_len = softmax_d.size(-1);
    p.b = (int32_t)a.batch; p.seqlen_q = (int32_t)a.max_seqlen_q; p.seqlen_k = (int32_t)a.max_seqlen_k;
    p.dtype = dtype_code(q);
    p.cu_seqlens_q = static_cast<const int32_t *>(ptr(a.cu_seqlens_q));
    p.cu_seqlens_k = static_cast<const int32_t *>(ptr(a.cu_seqlens_k));
    p.softmax_scale = (float)a.softmax_scale; p.softcap = (float)a.softcap;
    p.is_causal = a.causal ? 1 : 0;
    p.window_size_left = (int32_t)a.window_left; p.window_size_right = (int32_t)a.window_right;
    if (a.alibi.has_value()) {
        p.alibi_slopes = static_cast<const float *>(a.alibi->data_ptr());
        p.alibi_slopes_batch_stride = a.alibi->dim() == 2 ? a.alibi->stride(0) : 0;
    }
    p.flags = g_fa3_window ? FA_FLAG_FA3_WINDOW : 0;
    p.deterministic = a.deterministic ? 1 : 0;
    p.p_dropout = (float)a.p_dropout;
    p.rng_state = static_cast<const uint64_t *>(ptr(a.rng_state));
    const int st = fa_bwd(&p, current_stream(q));
    TORCH_CHECK(st == 0, "fa_bwd failed (", st, "): ", fa_strerror(st));
}

void bwd_common_checks(const Tensor &dout, const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out,
                       const Tensor &softmax_lse) {
    const auto q_dtype = q.scalar_type();
    TORCH_CHECK(q_dtype == at::kHalf || q_dtype == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
    TORCH_CHECK(k.scalar_type() == q_dtype, "query and key must have the same dtype");
    TORCH_CHECK(v.scalar_type() == q_dtype, "query and value must have the same dtype");
    TORCH_CHECK(out.scalar_type() == q_dtype, "query and out must have the same dtype");
    TORCH_CHECK(dout.scalar_type() == q_dtype, "query and dout must have the same dtype");
    CHECK_DEVICE(q, "q"); CHECK_DEVICE(k, "k"); CHECK_DEVICE(v, "v"); CHECK_DEVICE(out, "out"); CHECK_DEVICE(dout, "dout");
    CHECK_DEVICE(softmax_lse, "softmax_lse");
    CHECK_LAST_CONTIGUOUS(q, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(k, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(v, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(out, "out tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(dout, "dout tensor must have contiguous last dimension");
}

OptTensor bwd_rng_state(double p_dropout, c10::optional<at::Generator> &gen_, OptTensor &rng_state, const Tensor &q) {
    TORCH_CHECK(p_dropout >= 0.0 && p_dropout < 1.0, "p_dropout must be in [0, 1)");
    if (p_dropout <= 0.0) return c10::nullopt;
    // the forward's (seed, offset); without it a fresh pair is drawn like the reference (:895-910)
    Tensor rs = rng_state.has_value() ? rng_state.value() : dropout_state(p_dropout, gen_, q);
    TORCH_CHECK(rs.scalar_type() == at::kLong && rs.numel() == 2 && rs.is_cuda() && rs.is_contiguous(),
                "rng_state must be a contiguous int64 CUDA tensor with 2 elements");
    return rs;
}

std::vector<Tensor> mha_bwd(const Tensor &dout, const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out,
                            const Tensor &softmax_lse, OptTensor &dq_, OptTensor &dk_, OptTensor &dv_, OptTensor &alibi_slopes_,
                            const double p_dropout, const double softmax_scale, const bool is_causal, int64_t window_size_left,
                            int64_t window_size_right, const double softcap, const bool deterministic,
                            c10::optional<at::Generator> gen_, OptTensor &rng_state) {
    bwd_common_checks(dout, q, k, v, out, softmax_lse);
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4, "q, k must have 4 dimensions");
    const int64_t batch_size = q.size(0), seqlen_q = q.size(1), num_heads = q.size(2), head_size = q.size(3);
    const int64_t seqlen_k = k.size(1), num_heads_k = k.size(2);
    TORCH_CHECK(batch_size > 0, "batch size must be positive");
    TORCH_CHECK(head_size % 8 == 0, "head_size should be a multiple of 8");
    TORCH_CHECK(head_size <= 256, "FlashAttention backward only supports head dimension at most 256");
    TORCH_CHECK(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query");
    if (softcap > 0.0) TORCH_CHECK(p_dropout == 0.0, "Softcapping does not support dropout for now");
    const OptTensor rs = bwd_rng_state(p_dropout, gen_, rng_state, q);
    const OptTensor alibi = check_alibi(alibi_slopes_, batch_size, num_heads);
    CHECK_SHAPE(q, "q", batch_size, seqlen_q, num_heads, head_size);
    CHECK_SHAPE(k, "k", batch_size, seqlen_k, num_heads_k, head_size);
    CHECK_SHAPE(v, "v", batch_size, seqlen_k, num_heads_k, head_size);
    CHECK_SHAPE(out, "out", batch_size, seqlen_q, num_heads, head_size);
    CHECK_SHAPE(dout, "dout", batch_size, seqlen_q, num_heads, head_size);
    Tensor dq = grad_out(dq_, q, "dq", {batch_size, seqlen_q, num_heads, head_size});
    Tensor dk = grad_out(dk_, k, "dk", {batch_size, seqlen_k, num_heads_k, head_size});
    Tensor dv = grad_out(dv_, v, "dv", {batch_size, seqlen_k, num_heads_k, head_size});
    c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
    Tensor softmax_d = at::empty({batch_size, num_heads, round128(seqlen_q)}, q.options().dtype(at::kFloat));
    if (seqlen_q > 0 && seqlen_k > 0) {
        const Tensor doc = aligned_or_copy(dout), qc = aligned_or_copy(q), kc = aligned_or_copy(k), vc = aligned_or_copy(v),
                     oc = aligned_or_copy(out);
        Tensor dqc = aligned(dq) ? dq : at::empty_like(dq, at::MemoryFormat::Contiguous);
        Tensor dkc = aligned(dk) ? dk : at::empty_like(dk, at::MemoryFormat::Contiguous);
        Tensor dvc = aligned(dv) ? dv : at::empty_like(dv, at::MemoryFormat::Contiguous);
        const Tensor lse = softmax_lse.is_contiguous() ? softmax_lse : softmax_lse.contiguous();
        BwdArgs a;
        a.batch = batch_size; a.max_seqlen_q = seqlen_q; a.max_seqlen_k = seqlen_k; a.softmax_scale = softmax_scale;
        a.causal = is_causal; a.window_left = window_size_left; a.window_right = window_size_right; a.softcap = softcap;
        a.alibi = alibi; a.deterministic = deterministic; a.p_dropout = p_dropout; a.rng_state = rs;
        launch_bwd(doc, qc, kc, vc, oc, lse, dqc, dkc, dvc, softmax_d, a);
        if (!dqc.is_same(dq)) dq.copy_(dqc);
        if (!dkc.is_same(dk)) dk.copy_(dkc);
        if (!dvc.is_same(dv)) dv.copy_(dvc);
    } else {
        dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_();  // (:953-958)
    }
    return {dq, dk, dv, softmax_d};
}

std::vector<Tensor> mha_varlen_bwd(const Tensor &dout, const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out,
                                   const Tensor &softmax_lse, OptTensor &dq_, OptTensor &dk_, OptTensor &dv_,
                                   const Tensor &cu_seqlens_q, const Tensor &cu_seqlens_k, OptTensor &alibi_slopes_,
                                   const int64_t max_seqlen_q, const int64_t max_seqlen_k, const double p_dropout,
                                   const double softmax_scale, const bool zero_tensors, const bool is_causal,
                                   int64_t window_size_left, int64_t window_size_right, const double softcap,
                                   const bool deterministic, c10::optional<at::Generator> gen_, OptTensor &rng_state) {
    bwd_common_checks(dout, q, k, v, out, softmax_lse);
    TORCH_CHECK(cu_seqlens_q.scalar_type() == at::kInt, "cu_seqlens_q must have dtype int32");
    TORCH_CHECK(cu_seqlens_k.scalar_type() == at::kInt, "cu_seqlens_k must have dtype int32");
    CHECK_DEVICE(cu_seqlens_q, "cu_seqlens_q"); CHECK_DEVICE(cu_seqlens_k, "cu_seqlens_k");
    TORCH_CHECK(cu_seqlens_q.is_contiguous(), "cu_seqlens_q must be contiguous");
    TORCH_CHECK(cu_seqlens_k.is_contiguous(), "cu_seqlens_k must be contiguous");
    TORCH_CHECK(q.dim() == 3 && k.dim() == 3, "q, k must have 3 dimensions");
    const int64_t total_q = q.size(0), num_heads = q.size(1), head_size = q.size(2);
    const int64_t batch_size = cu_seqlens_q.numel() - 1;
    const int64_t total_k = k.size(0), num_heads_k = k.size(1);
    TORCH_CHECK(batch_size > 0, "batch size must be positive");
    TORCH_CHECK(head_size % 8 == 0, "head_size should be a multiple of 8");
    TORCH_CHECK(head_size <= 256, "FlashAttention backward only supports head dimension at most 256");
    TORCH_CHECK(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query");
    if (softcap > 0.0) TORCH_CHECK(p_dropout == 0.0, "Softcapping does not support dropout for now");
    const OptTensor rs = bwd_rng_state(p_dropout, gen_, rng_state, q);
    const OptTensor alibi = check_alibi(alibi_slopes_, batch_size, num_heads);
    CHECK_SHAPE(q, "q", total_q, num_heads, head_size);
    CHECK_SHAPE(k, "k", total_k, num_heads_k, head_size);
    CHECK_SHAPE(v, "v", total_k, num_heads_k, head_size);
    CHECK_SHAPE(out, "out", total_q, num_heads, head_size);
    CHECK_SHAPE(dout, "dout", total_q, num_heads, head_size);
    CHECK_SHAPE(cu_seqlens_q, "cu_seqlens_q", batch_size + 1);
    CHECK_SHAPE(cu_seqlens_k, "cu_seqlens_k", batch_size + 1);
    Tensor dq = grad_out(dq_, q, "dq", {total_q, num_heads, head_size});
    Tensor dk = grad_out(dk_, k, "dk", {total_k, num_heads_k, head_size});
    Tensor dv = grad_out(dv_, v, "dv", {total_k, num_heads_k, head_size});
    c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
    Tensor softmax_d = at::empty({num_heads, total_q + 128 * batch_size}, q.options().dtype(at::kFloat));
    if (zero_tensors) { dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_(); }
    if (max_seqlen_q > 0 && total_q > 0 && total_k > 0) {
        const Tensor doc = aligned_or_copy(dout), qc = aligned_or_copy(q), kc = aligned_or_copy(k), vc = aligned_or_copy(v),
                     oc = aligned_or_copy(out);
        Tensor dqc = aligned(dq) ? dq : at::empty_like(dq, at::MemoryFormat::Contiguous);
        Tensor dkc = aligned(dk) ? dk : at::empty_like(dk, at::MemoryFormat::Contiguous);
        Tensor dvc = aligned(dv) ? dv : at::empty_like(dv, at::MemoryFormat::Contiguous);
        const Tensor lse = softmax_lse.is_contiguous() ? softmax_lse : softmax_lse.contiguous();
        BwdArgs a;
        a.varlen = true; a.batch = batch_size; a.max_seqlen_q = max_seqlen_q; a.max_seqlen_k = max_seqlen_k;
        a.softmax_scale = softmax_scale; a.causal = is_causal; a.window_left = window_size_left; a.window_right = window_size_right;
        a.softcap = softcap; a.cu_seqlens_q = cu_seqlens_q; a.cu_seqlens_k = cu_seqlens_k; a.alibi = alibi;
        a.deterministic = deterministic; a.p_dropout = p_dropout; a.rng_state = rs;
        launch_bwd(doc, qc, kc, vc, oc, lse, dqc, dkc, dvc, softmax_d, a);
        if (!dqc.is_same(dq)) dq.copy_(dqc);
        if (!dkc.is_same(dk)) dk.copy_(dkc);
        if (!dvc.is_same(dv)) dv.copy_(dvc);
    } else {
        dq.zero_(); dk.zero_(); dv.zero_(); softmax_d.zero_();
    }
    return {dq, dk, dv, softmax_d};
}

void rotary_apply(const Tensor &src, const Tensor &dst, const Tensor &cos, const Tensor &sin, const Tensor &seqlen_offsets,
                  bool interleaved, bool per_row_positions) {
    fa_rotary_params p{};
    p.abi_version = FA_ABI_VERSION;
    p.struct_size = sizeof(fa_rotary_params);
    p.src = src.data_ptr(); p.dst = dst.data_ptr();
    p.src_batch_stride = src.stride(0); p.src_row_stride = src.stride(1); p.src_head_stride = src.stride(2);
    p.dst_batch_stride = dst.stride(0); p.dst_row_stride = dst.stride(1); p.dst_head_stride = dst.stride(2);
    p.b = (int32_t)src.size(0); p.s = (int32_t)src.size(1); p.h = (int32_t)src.size(2); p.d = (int32_t)src.size(3);
    p.dtype = dtype_code(src);
    p.rotary_dim = (int32_t)cos.size(1) * 2;
    p.rotary_interleaved = interleaved ? 1 : 0;
    p.per_row_positions = per_row_positions ? 1 : 0;
    p.rotary_cos = cos.data_ptr(); p.rotary_sin = sin.data_ptr();
    p.seqlen_offsets = static_cast<const int32_t *>(seqlen_offsets.data_ptr());
    const int st = fa_rotary_apply(&p, current_stream(src));
    TORCH_CHECK(st == 0, "fa_rotary_apply failed (", st, "): ", fa_strerror(st));
}

void kvcache_append(const Tensor &k_new, const Tensor &v_new, const Tensor &k_cache, const Tensor &v_cache,
                    const Tensor &cache_seqlens, const OptTensor &cache_batch_idx, const OptTensor &block_table,
                    const OptTensor &rotary_cos, const OptTensor &rotary_sin, bool rotary_interleaved,
                    const OptTensor &rotary_seqlens = c10::nullopt) {
    fa_kvcache_append_params p{};
    p.abi_version = FA_ABI_VERSION;
    p.struct_size = sizeof(fa_kvcache_append_params);
    p.k_new = k_new.data_ptr(); p.v_new = v_new.data_ptr(); p.k_cache = k_cache.data_ptr(); p.v_cache = v_cache.data_ptr();
    p.knew_batch_stride = k_new.stride(0); p.knew_row_stride = k_new.stride(1); p.knew_head_stride = k_new.stride(2);
    p.vnew_batch_stride = v_new.stride(0); p.vnew_row_stride = v_new.stride(1); p.vnew_head_stride = v_new.stride(2);
    p.kcache_batch_stride = k_cache.stride(0); p.kcache_row_stride = k_cache.stride(1); p.kcache_head_stride = k_cache.stride(2);
    p.vcache_batch_stride = v_cache.stride(0); p.vcache_row_stride = v_cache.stride(1); p.vcache_head_stride = v_cache.stride(2);
    p.b = (int32_t)k_new.size(0); p.seqlen_new = (int32_t)k_new.size(1); p.h_k = (int32_t)k_new.size(2); p.d = (int32_t)k_new.size(3);
    p.seqlen_cache = (int32_t)k_cache.size(1);
    if (block_table.has_value()) {
        p.block_table = static_cast<const int32_t *>(block_table->data_ptr());
        p.block_table_batch_stride = block_table->stride(0);
        p.page_block_size = (int32_t)k_cache.size(1);
        p.seqlen_cache = (int32_t)(block_table->size(1) * k_cache.size(1));
    }
    p.cache_seqlens = static_cast<const int32_t *>(cache_seqlens.data_ptr());
    p.cache_batch_idx = static_cast<const int32_t *>(ptr(cache_batch_idx));
    p.dtype = dtype_code(k_new);
    if (rotary_cos.has_value()) {
        p.rotary_cos = rotary_cos->data_ptr(); p.rotary_sin = rotary_sin->data_ptr();
        p.rotary_dim = (int32_t)rotary_cos->size(1) * 2;
        p.rotary_interleaved = rotary_interleaved ? 1 : 0;
        p.rotary_seqlens = static_cast<const int32_t *>(ptr(rotary_seqlens));   // FA3 seqlens_rotary (NULL: the cache fill levels)
    }
    const int st = fa_kvcache_append(&p, current_stream(k_new));
    TORCH_CHECK(st == 0, "fa_kvcache_append failed (", st, "): ", fa_strerror(st));
}

// mha_fwd_kvcache, csrc/flash_attn/flash_api.cpp:1202-1476 (+ the page-size rule of the calling surface: FA2 256, FA3 any)
std::vector<Tensor> fwd_kvcache_impl(Tensor q, const Tensor &kcache, const Tensor &vcache, OptTensor k_, OptTensor v_,
                                     OptTensor seqlens_k_, OptTensor rotary_cos_, OptTensor rotary_sin_,
                                     OptTensor cache_batch_idx_, OptTensor leftpad_k_, OptTensor block_table_,
                                     OptTensor alibi_slopes_, OptTensor out_, const double softmax_scale, bool is_causal,
                                     int64_t window_size_left, int64_t window_size_right, const double softcap,
                                     bool is_rotary_interleaved, int64_t num_splits, int64_t page_multiple,
                                     OptTensor seqlens_rotary_) {
    const auto q_dtype = q.scalar_type();
    TORCH_CHECK(q_dtype == at::kHalf || q_dtype == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
    TORCH_CHECK(kcache.scalar_type() == q_dtype, "query and key must have the same dtype");
    TORCH_CHECK(vcache.scalar_type() == q_dtype, "query and value must have the same dtype");
    CHECK_DEVICE(q, "q"); CHECK_DEVICE(kcache, "kcache"); CHECK_DEVICE(vcache, "vcache");
    CHECK_LAST_CONTIGUOUS(q, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(kcache, "Input tensor must have contiguous last dimension");
    CHECK_LAST_CONTIGUOUS(vcache, "Input tensor must have contiguous last dimension");
    const bool paged = block_table_.has_value();
    if (paged) TORCH_CHECK(!cache_batch_idx_.has_value(), "Paged KVcache does not support cache_batch_idx");
    TORCH_CHECK(q.dim() == 4 && kcache.dim() == 4, "q, kcache must have 4 dimensions");
    const int64_t batch_size = q.size(0), head_size_og = q.size(3);
    int64_t seqlen_q = q.size(1), num_heads = q.size(2);
    int64_t batch_size_c = kcache.size(0), seqlen_k = kcache.size(1);
    const int64_t num_heads_k = kcache.size(2);
    int64_t page_block_size = 0;
    if (paged) {
        auto pr = check_block_table(*block_table_, kcache, batch_size, page_multiple);
        page_block_size = pr.first;
        seqlen_k = pr.second * page_block_size; batch_size_c = batch_size;  // (:1266-1268)
    }
    TORCH_CHECK(batch_size > 0, "batch size must be positive");
    TORCH_CHECK(head_size_og <= 256, "FlashAttention forward only supports head dimension at most 256");
    TORCH_CHECK(head_size_og % 8 == 0, "This flash attention build needs head_size to be a multiple of 8 in fwd_kvcache");
    TORCH_CHECK(num_heads % num_heads_k == 0, "Number of heads in key/value must divide number of heads in query");
    const OptTensor alibi = check_alibi(alibi_slopes_, batch_size, num_heads);
    if (seqlen_q == 1 && !alibi.has_value()) is_causal = false;  // (:1270)
    if (is_causal) window_size_right = 0;
    // (b, 1, (h_k ngroups), d) -> (b, ngroups, h_k, d): one pass over the cache serves the whole GQA group (:1272-1285)
    const bool swapped = seqlen_q == 1 && num_heads > num_heads_k && window_size_left < 0 && window_size_right < 0 && !alibi.has_value();
    if (swapped) {
        const int64_t ngroups = num_heads / num_heads_k;
        q = q.reshape({batch_size, num_heads_k, ngroups, head_size_og}).transpose(1, 2);
        seqlen_q = ngroups; num_heads = num_heads_k;
    }
    CHECK_SHAPE(q, "q", batch_size, seqlen_q, num_heads, head_size_og);
    if (paged) {
        CHECK_SHAPE(kcache, "kcache", kcache.size(0), page_block_size, num_heads_k, head_size_og);
        CHECK_SHAPE(vcache, "vcache", kcache.size(0), page_block_size, num_heads_k, head_size_og);
    } else {
        CHECK_SHAPE(kcache, "kcache", batch_size_c, seqlen_k, num_heads_k, head_size_og);
        CHECK_SHAPE(vcache, "vcache", batch_size_c, seqlen_k, num_heads_k, head_size_og);
    }
    Tensor out;
    if (out_.has_value() && !swapped) {
        out = out_.value();
        TORCH_CHECK(out.scalar_type() == q_dtype, "Output must have the same dtype as inputs");
        CHECK_DEVICE(out, "out");
        TORCH_CHECK(out.stride(-1) == 1, "Output tensor must have contiguous last dimension");
        CHECK_SHAPE(out, "out", batch_size, seqlen_q, num_heads, head_size_og);
    } else {
        out = at::empty({batch_size, seqlen_q, num_heads, head_size_og}, q.options());
    }
    int64_t seqlen_knew = 0;
    if (k_.has_value()) {
        TORCH_CHECK(v_.has_value(), "If key is supplied, value must also be passed in");
        TORCH_CHECK(seqlens_k_.has_value(), "If key is supplied, seqlens_k must also be passed in");
        TORCH_CHECK(seqlen_q <= seqlen_k, "If key is supplied, it must have seqlen <= the seqlen of the KV cache");
        TORCH_CHECK(k_->scalar_type() == q_dtype, "Key must have the same dtype as query");
        TORCH_CHECK(v_->scalar_type() == q_dtype, "Value must have the same dtype as query");
        CHECK_DEVICE(*k_, "k"); CHECK_DEVICE(*v_, "v");
        TORCH_CHECK(k_->stride(-1) == 1, "Key tensor must have contiguous last dimension");
        TORCH_CHECK(v_->stride(-1) == 1, "Value tensor must have contiguous last dimension");
        seqlen_knew = k_->size(1);
        CHECK_SHAPE(*k_, "k", batch_size, seqlen_knew, num_heads_k, head_size_og);
        CHECK_SHAPE(*v_, "v", batch_size, seqlen_knew, num_heads_k, head_size_og);
    }
    if (seqlens_k_.has_value()) {
        TORCH_CHECK(seqlens_k_->scalar_type() == at::kInt, "seqlens_k must have dtype int32");
        CHECK_DEVICE(*seqlens_k_, "seqlens_k");
        TORCH_CHECK(seqlens_k_->is_contiguous(), "seqlens_k must be contiguous");
        CHECK_SHAPE(*seqlens_k_, "seqlens_k", batch_size);
    }
    check_leftpad(leftpad_k_, batch_size, paged);
    if (cache_batch_idx_.has_value()) {
        CHECK_DEVICE(*cache_batch_idx_, "cache_batch_idx");
        TORCH_CHECK(cache_batch_idx_->is_contiguous(), "cache_batch_idx must be contiguous");
        TORCH_CHECK(cache_batch_idx_->scalar_type() == at::kInt, "cache_batch_idx must have dtype int32");
    } else {
        TORCH_CHECK(batch_size_c >= batch_size, "the KV cache must have at least batch_size entries");
    }
    TORCH_CHECK(aligned(kcache) && aligned(vcache),
                "the KV cache must be 16-byte aligned with row/head/batch strides that are multiples of 8");
    const bool rotary = rotary_cos_.has_value();
    if (rotary) {  // (:1404-1428)
        TORCH_CHECK(k_.has_value(), "If rotary cos/sin are provided, new key / value to be appended to KV cache must also be provided");
        CHECK_DEVICE(*rotary_cos_, "rotary_cos");
        const int64_t rotary_dim = rotary_cos_->size(1) * 2;
        TORCH_CHECK(rotary_dim <= head_size_og, "rotary_dim must be <= headdim");
        TORCH_CHECK(rotary_dim % 16 == 0, "Only rotary dimensions divisible by 16 are currently supported");
        const int64_t seqlen_ro = rotary_cos_->size(0);
        TORCH_CHECK(seqlen_ro >= seqlen_k, "cos/sin seqlen must be at least the seqlen of KV cache");
        CHECK_SHAPE(*rotary_cos_, "rotary_cos", seqlen_ro, rotary_dim / 2);
        TORCH_CHECK(rotary_cos_->is_contiguous(), "rotary_cos must be contiguous");
        TORCH_CHECK(rotary_cos_->scalar_type() == q_dtype, "rotary_cos must have the same dtype as query");
        TORCH_CHECK(rotary_sin_.has_value(), "If rotary cos is provided, rotary sin must also be provided");
        CHECK_DEVICE(*rotary_sin_, "rotary_sin");
        CHECK_SHAPE(*rotary_sin_, "rotary_sin", seqlen_ro, rotary_dim / 2);
        TORCH_CHECK(rotary_sin_->is_contiguous(), "rotary_sin must be contiguous");
        TORCH_CHECK(rotary_sin_->scalar_type() == q_dtype, "rotary_cos must have the same dtype as query");
    }
    c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
    Tensor softmax_lse = at::empty({batch_size, num_heads, seqlen_q}, q.options().dtype(at::kFloat));
    OptTensor seqused = seqlens_k_;
    if (seqlen_knew > 0) {  // "Append_KV": new rows land at [seqlens_k, seqlens_k + seqlen_knew) of each cache entry
        const Tensor kn = aligned_or_copy(*k_), vn = aligned_or_copy(*v_);
        kvcache_append(kn, vn, kcache, vcache, *seqlens_k_, cache_batch_idx_, block_table_, rotary_cos_, rotary_sin_, is_rotary_interleaved,
                       seqlens_rotary_);
        seqused = *seqlens_k_ + seqlen_knew;
    }
    Tensor qc = aligned_or_copy(q);
    if (rotary) {
        // causal / local: query row i sits at position seqlens_k + i; otherwise every row at seqlens_k
        // (flash_attn/flash_attn_interface.py:1516-1524, src/flash_fwd_kernel.h:753-775)
        const bool per_row = is_causal || window_size_left >= 0 || window_size_right >= 0;
        Tensor q_ro = at::empty_like(qc, at::MemoryFormat::Contiguous);
        rotary_apply(qc, q_ro, *rotary_cos_, *rotary_sin_, seqlens_rotary_.has_value() ? *seqlens_rotary_ : *seqlens_k_,
                     is_rotary_interleaved, per_row);
        qc = q_ro;
    }
    Tensor oc = aligned(out) ? out : at::empty_like(out, at::MemoryFormat::Contiguous);
    if (seqlen_k > 0) {
        FwdArgs a;
        a.batch = batch_size; a.max_seqlen_q = seqlen_q; a.max_seqlen_k = seqlen_k; a.softmax_scale = softmax_scale;
        a.causal = is_causal; a.window_left = window_size_left; a.window_right = window_size_right; a.softcap = softcap;
        a.seqused_k = seqused; a.alibi = alibi; a.kv_batch_idx = cache_batch_idx_; a.block_table = block_table_;
        a.num_splits = (int)num_splits; a.leftpad_k = leftpad_k_;
        launch_fwd(qc, kcache, vcache, oc, softmax_lse, a);
        if (!oc.is_same(out)) out.copy_(oc);
    } else {
        out.zero_();
        softmax_lse.fill_(std::numeric_limits<float>::infinity());
    }
    if (swapped) {
        out = out.transpose(1, 2).reshape({batch_size, 1, num_heads_k * seqlen_q, head_size_og});
        softmax_lse = softmax_lse.reshape({batch_size, num_heads_k * seqlen_q, 1});
        if (out_.has_value()) {
            out_->copy_(out);
            out = out_.value();
        }
    }
    return {out, softmax_lse};
}

std::vector<Tensor> mha_fwd_kvcache(Tensor &q, const Tensor &kcache, const Tensor &vcache, OptTensor &k_, OptTensor &v_,
                                    OptTensor &seqlens_k_, OptTensor &rotary_cos_, OptTensor &rotary_sin_,
                                    OptTensor &cache_batch_idx_, OptTensor &leftpad_k_, OptTensor &block_table_,
                                    OptTensor &alibi_slopes_, OptTensor &out_, const double softmax_scale, bool is_causal,
                                    int64_t window_size_left, int64_t window_size_right, const double softcap,
                                    bool is_rotary_interleaved, int64_t num_splits) {
    return fwd_kvcache_impl(q, kcache, vcache, k_, v_, seqlens_k_, rotary_cos_, rotary_sin_, cache_batch_idx_, leftpad_k_,
                            block_table_, alibi_slopes_, out_, softmax_scale, is_causal, window_size_left, window_size_right,
                            softcap, is_rotary_interleaved, num_splits, 256, c10::nullopt);  // the reference's page rule (:1265)
}

bool set_fa3_window_rule(bool on) {
    const bool prev = g_fa3_window;
    g_fa3_window = on;
    return prev;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "FlashAttention (MI355X / gfx950 native kernels behind the flash_attn_2_cuda surface)";
    m.def("fwd", &mha_fwd, "Forward pass");
    m.def("varlen_fwd", &mha_varlen_fwd, "Forward pass (variable length)");
    m.def("bwd", &mha_bwd, "Backward pass");
    m.def("varlen_bwd", &mha_varlen_bwd, "Backward pass (variable length)");
    m.def("fwd_kvcache", &mha_fwd_kvcache, "Forward pass, with KV-cache");
    m.def("_fwd_kvcache_impl", &fwd_kvcache_impl, "fwd_kvcache with the page-size rule of the calling surface (+ FA3 seqlens_rotary)",
          py::arg("q"), py::arg("kcache"), py::arg("vcache"), py::arg("k"), py::arg("v"), py::arg("seqlens_k"), py::arg("rotary_cos"),
          py::arg("rotary_sin"), py::arg("cache_batch_idx"), py::arg("leftpad_k"), py::arg("block_table"), py::arg("alibi_slopes"),
          py::arg("out"), py::arg("softmax_scale"), py::arg("is_causal"), py::arg("window_size_left"), py::arg("window_size_right"),
          py::arg("softcap"), py::arg("is_rotary_interleaved"), py::arg("num_splits"), py::arg("page_multiple"),
          py::arg("seqlens_rotary") = py::none());
    m.def("_set_fa3_window_rule", &set_fa3_window_rule, "FA3 window rule for the backward entry points (returns the previous value)");
}
