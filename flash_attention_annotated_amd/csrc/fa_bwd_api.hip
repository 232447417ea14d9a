// fa_bwd_api.hip — C-ABI entry points declared in include/fa_bwd.h.
//
// Host-side role of mha_bwd / mha_varlen_bwd (csrc/flash_attn/flash_api.cpp:767-971, 973-1200),
// set_params_dgrad (:161-221) and run_mha_bwd (:757-765): validate, fill the kernel params, launch the three
// kernels on the caller's stream.  No allocation, no synchronisation.
#include "fa_bwd.h"
#include "fa_bwd_kernel.h"

#include <algorithm>
#include <atomic>
#include <cmath>

namespace {

int head_dim_tile_b(int d) {
    if (d <= 64) return 64;
    if (d <= 128) return 128;
    return 256;
}

template <typename K>
int launch_kernel(K kernel, int smem, std::atomic<uint64_t> &attr_set, int grid, int threads, const fa::BParams &bp,
                  hipStream_t stream) {
    // the > 64 KiB dynamic-LDS opt-in is a per-device attribute of the kernel: one bit per device ordinal
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (smem > 65536 && !(attr_set.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            (void)hipGetLastError();
            return FA_ERR_LAUNCH;
        }
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), smem, stream, bp);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

// grid of decode_block(): whole units of `blocks` workgroups for as many (batch, head) units as deal evenly over the 8 XCDs,
// the remaining heads block by block (any head count loads the XCDs equally; fewer than 16 units: everything block by block)
int64_t unit_grid(int64_t tiles, int blocks, int32_t &whole_slots) {
    const int64_t units = tiles / blocks;
    const int64_t ws = units >= 16 ? units / 8 * blocks : 0;
    whole_slots = (int32_t)ws;
    return 8 * (ws + (tiles - ws * 8 + 7) / 8);
}

template <typename T, int D, bool SOFTCAP, bool DROPOUT = false>
int run_bwd(fa::BParams bp, int rows_q_max, int rows_k_max, hipStream_t stream) {
    // 32-wide blocks per wave (every LDS fragment feeds NB MFMAs): two wherever accumulators + resident operands fit
    // the 512-register budget of a lone wave.  dQ: 2 for D <= 128; dK/dV (two accumulator sets + K and V): 2 for D = 64
    // (at D = 128 two blocks would fill all 256 AGPRs with accumulators; hipcc then rotates the whole AGPR file
    //  through v_accvgpr_mov to find temporaries -- measured 3x slower and not worth fighting).
    //  At D = 64 the plain problem -- no softcap / dropout / ALiBi -- takes the one-block form as well: that is the shape of the
    //  generated tile loop, tools/gen_bwd_loop.py.)
    constexpr int NBQ = D <= 128 ? 2 : 1;
    constexpr int NBK2 = D <= 64 ? 2 : 1;
    const bool one_block = D == 64 && !SOFTCAP && !DROPOUT && !bp.alibi;
    // 1. D = rowsum(dO * O)
    {
        const int64_t rows = bp.cu_seqlens_q ? (int64_t)bp.total_q : (int64_t)bp.b * bp.seqlen_q;
        const int64_t items = rows * bp.h;
        constexpr int LPR = D / 8 > 32 ? 32 : D / 8;  // lanes per (row, head)
        const int per_block = 256 / LPR;
        const int grid = (int)std::min<int64_t>((items + per_block - 1) / per_block, 256 * 16);
        if (items > 0) {
            hipLaunchKernelGGL((fa::bwd_dot_kernel<T, LPR>), dim3(grid), dim3(256), 0, stream, bp);
            if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
        }
    }
    // 2. dK, dV
    {
        static std::atomic<uint64_t> attr{0}, attr1{0};
        const int NBK = one_block ? 1 : NBK2;
        bp.num_blocks = (rows_k_max + 128 * NBK - 1) / (128 * NBK);
        const int64_t tiles = (int64_t)bp.num_blocks * bp.h_k * bp.b;
        if (tiles > 0x7fffffff) return FA_ERR_BAD_SHAPE;
        if (tiles > 0) {
            bp.num_tiles = (int32_t)tiles;
            const int64_t grid = unit_grid(tiles, bp.num_blocks, bp.whole_slots);
            if (grid > 0x7fffffff) return FA_ERR_BAD_SHAPE;
            bp.grid = (int32_t)grid;
            int st;
            if constexpr (D == 128 && !SOFTCAP && !DROPOUT) {
                // head dims <= 96 on the 128-wide tiles: the instantiation whose generated loop skips the zero padding
                static std::atomic<uint64_t> attr96{0};
                st = bp.d <= 96 ? launch_kernel(fa::bwd_dkdv_kernel<T, D, 1, false, false, 96>, fa::smem_bytes_dkdv<D>(), attr96, bp.grid, 256, bp, stream)
                                : launch_kernel(fa::bwd_dkdv_kernel<T, D, 1, false, false>, fa::smem_bytes_dkdv<D>(), attr, bp.grid, 256, bp, stream);
            } else if constexpr (D == 256) {
                // head-dim tile 256: dV and dK by a launch each (PART 1 / 2, fa_bwd_kernel.h) -- one accumulator set per sweep;
                // head dims <= 160 / <= 192 on the instantiations that skip the zero padding (DEFF)
                auto two = [&](auto deff_c) {
                    constexpr int DEFF = decltype(deff_c)::value;
                    static std::atomic<uint64_t> attr_dv{0}, attr_dk{0};
                    int s2 = launch_kernel(fa::bwd_dkdv_kernel<T, D, 1, SOFTCAP, DROPOUT, DEFF, 1>, fa::smem_bytes_dkdv<D>(), attr_dv, bp.grid, 256, bp, stream);
                    if (s2 == FA_OK)
                        s2 = launch_kernel(fa::bwd_dkdv_kernel<T, D, 1, SOFTCAP, DROPOUT, DEFF, 2>, fa::smem_bytes_dkdv<D>(), attr_dk, bp.grid, 256, bp, stream);
                    return s2;
                };
                if constexpr (!SOFTCAP && !DROPOUT) {
                    const int w = std::max(bp.d, bp.d_v);
                    st = w <= 160 ? two(std::integral_constant<int, 160>{}) : w <= 192 ? two(std::integral_constant<int, 192>{}) : two(std::integral_constant<int, 256>{});
                } else {
                    st = two(std::integral_constant<int, 256>{});
                }
            } else {
                st = one_block ? launch_kernel(fa::bwd_dkdv_kernel<T, D, 1, SOFTCAP, DROPOUT>, fa::smem_bytes_dkdv<D>(), attr1, bp.grid, 256, bp, stream)
                               : launch_kernel(fa::bwd_dkdv_kernel<T, D, NBK2, SOFTCAP, DROPOUT>, fa::smem_bytes_dkdv<D>(), attr, bp.grid, 256, bp, stream);
            }
            if (st != FA_OK) return st;
        }
    }
    // 3. dQ
    {
        static std::atomic<uint64_t> attr{0};
        bp.num_blocks = (rows_q_max + 128 * NBQ - 1) / (128 * NBQ);
        const int64_t tiles = (int64_t)bp.num_blocks * bp.h * bp.b;
        if (tiles > 0x7fffffff) return FA_ERR_BAD_SHAPE;
        if (tiles > 0) {
            bp.num_tiles = (int32_t)tiles;
            const int64_t grid = unit_grid(tiles, bp.num_blocks, bp.whole_slots);
            if (grid > 0x7fffffff) return FA_ERR_BAD_SHAPE;
            bp.grid = (int32_t)grid;
            int st;
            if constexpr (D == 128 && !SOFTCAP && !DROPOUT) {
                static std::atomic<uint64_t> attr96{0};
                st = bp.d <= 96 ? launch_kernel(fa::bwd_dq_kernel<T, D, NBQ, false, false, 96>, fa::smem_bytes_dq<D>(), attr96, bp.grid, 256, bp, stream)
                                : launch_kernel(fa::bwd_dq_kernel<T, D, NBQ, false, false>, fa::smem_bytes_dq<D>(), attr, bp.grid, 256, bp, stream);
            } else if constexpr (D == 256 && !SOFTCAP && !DROPOUT) {
                static std::atomic<uint64_t> attr160{0}, attr192{0};
                const int w = std::max(bp.d, bp.d_v);
                st = w <= 160 ? launch_kernel(fa::bwd_dq_kernel<T, D, NBQ, false, false, 160>, fa::smem_bytes_dq<D>(), attr160, bp.grid, 256, bp, stream)
                   : w <= 192 ? launch_kernel(fa::bwd_dq_kernel<T, D, NBQ, false, false, 192>, fa::smem_bytes_dq<D>(), attr192, bp.grid, 256, bp, stream)
                                 : launch_kernel(fa::bwd_dq_kernel<T, D, NBQ, false, false>, fa::smem_bytes_dq<D>(), attr, bp.grid, 256, bp, stream);
            } else {
                st = launch_kernel(fa::bwd_dq_kernel<T, D, NBQ, SOFTCAP, DROPOUT>, fa::smem_bytes_dq<D>(), attr, bp.grid, 256, bp, stream);
            }
            if (st != FA_OK) return st;
        }
    }
    return FA_OK;
}

template <typename T>
int dispatch_bwd(const fa::BParams &bp, bool softcap, int sq, int sk, hipStream_t stream) {
    const bool drop = bp.rp_dropout != 1.f;  // p > 0 (never together with softcap: fa_bwd_validate)
    switch (head_dim_tile_b(std::max(bp.d, bp.d_v))) {
        case 64:
            if (drop) return run_bwd<T, 64, false, true>(bp, sq, sk, stream);
            return softcap ? run_bwd<T, 64, true>(bp, sq, sk, stream) : run_bwd<T, 64, false>(bp, sq, sk, stream);
        case 128:
            if (drop) return run_bwd<T, 128, false, true>(bp, sq, sk, stream);
            return softcap ? run_bwd<T, 128, true>(bp, sq, sk, stream) : run_bwd<T, 128, false>(bp, sq, sk, stream);
        default:
            if (drop) return run_bwd<T, 256, false, true>(bp, sq, sk, stream);
            return softcap ? run_bwd<T, 256, true>(bp, sq, sk, stream) : run_bwd<T, 256, false>(bp, sq, sk, stream);
    }
}

}  // namespace

extern "C" {

uint32_t fa_bwd_params_size(void) { return (uint32_t)sizeof(fa_bwd_params); }

int fa_bwd_validate(const fa_bwd_params *p) {
    if (!p) return FA_ERR_NULL_POINTER;
    if (p->abi_version != FA_ABI_VERSION || p->struct_size != sizeof(fa_bwd_params)) return FA_ERR_BAD_ABI;
    if (p->dtype != FA_DTYPE_FP16 && p->dtype != FA_DTYPE_BF16) return FA_ERR_BAD_DTYPE;
    if (!(p->p_dropout >= 0.f && p->p_dropout < 1.f)) return FA_ERR_BAD_SHAPE;
    if (p->p_dropout > 0.f && (!p->rng_state || reinterpret_cast<uintptr_t>(p->rng_state) % 8 != 0)) return FA_ERR_NULL_POINTER;
    if (p->p_dropout > 0.f && p->softcap > 0.f) return FA_ERR_UNSUPPORTED;  // "Softcapping does not support dropout for now"
    if (p->b <= 0 || p->h <= 0 || p->h_k <= 0 || p->seqlen_q < 0 || p->seqlen_k < 0) return FA_ERR_BAD_SHAPE;
    if (p->d <= 0 || p->d > 256 || p->d % 8 != 0) return FA_ERR_BAD_HEAD_DIM;
    if (p->d_v < 0 || p->d_v % 8 != 0) return FA_ERR_BAD_HEAD_DIM;
    if (p->d_v > 0 && p->d_v != p->d && (p->d_v > 256 || std::max(p->d, p->d_v) <= 128)) return FA_ERR_UNSUPPORTED;  // wide tile only
    if (p->h % p->h_k != 0) return FA_ERR_BAD_HEADS;
    if ((p->cu_seqlens_q == nullptr) != (p->cu_seqlens_k == nullptr)) return FA_ERR_BAD_SHAPE;
    if (p->cu_seqlens_q && (p->total_q < 0 || p->total_k < 0)) return FA_ERR_BAD_SHAPE;
    const bool no_q = (p->seqlen_q == 0) || (p->cu_seqlens_q && p->total_q == 0);
    const bool no_k = (p->seqlen_k == 0) || (p->cu_seqlens_q && p->total_k == 0);
    if (!no_q && (!p->q || !p->o || !p->dout || !p->softmax_lse || !p->dq || !p->softmax_d)) return FA_ERR_NULL_POINTER;
    if (!no_k && (!p->k || !p->v || !p->dk || !p->dv)) return FA_ERR_NULL_POINTER;
    const int64_t strides[] = {p->q_row_stride, p->q_head_stride, p->k_row_stride, p->k_head_stride, p->v_row_stride,
                               p->v_head_stride, p->o_row_stride, p->o_head_stride, p->do_row_stride, p->do_head_stride,
                               p->dq_row_stride, p->dq_head_stride, p->dk_row_stride, p->dk_head_stride,
                               p->dv_row_stride, p->dv_head_stride};
    for (int64_t s : strides)
        if (s % 8 != 0) return FA_ERR_BAD_STRIDE;
    if (!p->cu_seqlens_q) {
        const int64_t bs[] = {p->q_batch_stride, p->k_batch_stride, p->v_batch_stride, p->o_batch_stride,
                              p->do_batch_stride, p->dq_batch_stride, p->dk_batch_stride, p->dv_batch_stride};
        for (int64_t s : bs)
            if (s % 8 != 0) return FA_ERR_BAD_STRIDE;
    }
    // tiles are addressed as 64-bit tile base + 32-bit (row * stride) lane offset
    if (p->q_row_stride < 0 || p->do_row_stride < 0 || p->q_row_stride >= (1 << 24) || p->do_row_stride >= (1 << 24) ||
        p->k_row_stride < 0 || p->v_row_stride < 0 || p->k_row_stride >= (1 << 24) || p->v_row_stride >= (1 << 24))
        return FA_ERR_BAD_STRIDE;
    const void *ptrs[] = {p->q, p->k, p->v, p->o, p->dout, p->dq, p->dk, p->dv};
    for (const void *ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return FA_ERR_BAD_STRIDE;
    const int64_t need = p->cu_seqlens_q ? (int64_t)p->total_q : (int64_t)p->seqlen_q;
    if (p->softmax_d_row_len < need) return FA_ERR_BAD_SHAPE;
    if (p->softcap < 0.f || std::isnan(p->softcap) || std::isnan(p->softmax_scale)) return FA_ERR_BAD_SHAPE;
    if (p->alibi_slopes && (reinterpret_cast<uintptr_t>(p->alibi_slopes) % 4 != 0 || p->alibi_slopes_batch_stride < 0 ||
                            p->alibi_slopes_batch_stride > 0x7fffffff))
        return FA_ERR_BAD_STRIDE;
    return FA_OK;
}

int fa_bwd(const fa_bwd_params *p, void *stream_) {
    const int st = fa_bwd_validate(p);
    if (st != FA_OK) return st;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool no_q = (p->seqlen_q == 0) || (p->cu_seqlens_q && p->total_q == 0);
    const bool no_k = (p->seqlen_k == 0) || (p->cu_seqlens_q && p->total_k == 0);
    if (no_q && no_k) return FA_OK;

    fa::BParams bp{};
    bp.q = p->q; bp.k = p->k; bp.v = p->v; bp.o = p->o; bp.dout = p->dout; bp.lse = p->softmax_lse;
    bp.dq = p->dq; bp.dk = p->dk; bp.dv = p->dv; bp.dsum = p->softmax_d;
    bp.cu_seqlens_q = p->cu_seqlens_q; bp.cu_seqlens_k = p->cu_seqlens_k;
    bp.q_batch_stride = p->q_batch_stride; bp.q_row_stride = p->q_row_stride; bp.q_head_stride = p->q_head_stride;
    bp.k_batch_stride = p->k_batch_stride; bp.k_row_stride = p->k_row_stride; bp.k_head_stride = p->k_head_stride;
    bp.v_batch_stride = p->v_batch_stride; bp.v_row_stride = p->v_row_stride; bp.v_head_stride = p->v_head_stride;
    bp.o_batch_stride = p->o_batch_stride; bp.o_row_stride = p->o_row_stride; bp.o_head_stride = p->o_head_stride;
    bp.do_batch_stride = p->do_batch_stride; bp.do_row_stride = p->do_row_stride; bp.do_head_stride = p->do_head_stride;
    bp.dq_batch_stride = p->dq_batch_stride; bp.dq_row_stride = p->dq_row_stride; bp.dq_head_stride = p->dq_head_stride;
    bp.dk_batch_stride = p->dk_batch_stride; bp.dk_row_stride = p->dk_row_stride; bp.dk_head_stride = p->dk_head_stride;
    bp.dv_batch_stride = p->dv_batch_stride; bp.dv_row_stride = p->dv_row_stride; bp.dv_head_stride = p->dv_head_stride;
    bp.dsum_row_len = p->softmax_d_row_len;
    bp.b = p->b; bp.seqlen_q = p->seqlen_q; bp.seqlen_k = p->seqlen_k; bp.h = p->h; bp.h_k = p->h_k; bp.d = p->d;
    bp.d_v = (p->d_v > 0) ? p->d_v : p->d;
    bp.total_q = p->total_q;
    bp.h_ratio = p->h / p->h_k;

    // window normalisation exactly as the forward (fa_fwd_api.hip; csrc/flash_attn/flash_api.cpp:790,836-837)
    int wl = p->window_size_left, wr = p->window_size_right;
    if (p->is_causal) wr = 0;
    if (!(p->flags & FA_FLAG_FA3_WINDOW)) {  // (FA3 rule: a negative side is unbounded, include/fa_fwd.h)
        if (wl >= p->seqlen_k) wl = -1;
        if (wr >= p->seqlen_k) wr = -1;
        if (p->is_causal) wr = 0;
        if (wl >= 0 && wr < 0) wr = p->seqlen_k;
    }
    bp.window_left = wl;
    bp.window_right = wr;

    const bool softcap = p->softcap > 0.f;
    constexpr float kLog2e = 1.4426950408889634f;
    if (softcap) {
        bp.softcap_pre = p->softmax_scale / p->softcap;
        bp.scale_log2 = p->softcap * kLog2e;
    } else {
        bp.softcap_pre = 0.f;
        bp.scale_log2 = p->softmax_scale * kLog2e;
    }
    bp.out_scale = p->softmax_scale;
    bp.alibi = p->alibi_slopes;
    bp.alibi_bs = (int32_t)p->alibi_slopes_batch_stride;
    bp.drop_thr = p->p_dropout > 0.f ? (int)std::floor(255.0 * (1.0 - (double)p->p_dropout)) : 255;  // as fa_fwd
    bp.rp_dropout = p->p_dropout > 0.f ? 1.f / (1.f - p->p_dropout) : 1.f;
    bp.rng_state = p->rng_state;

    // seqlen_q == 0: dK = dV = 0 is written by the dK/dV pass (no query tile is visible); seqlen_k == 0: dQ = 0 likewise
    if (p->dtype == FA_DTYPE_BF16) return dispatch_bwd<__bf16>(bp, softcap, p->seqlen_q, p->seqlen_k, stream);
    return dispatch_bwd<_Float16>(bp, softcap, p->seqlen_q, p->seqlen_k, stream);
}

}  // extern "C"
