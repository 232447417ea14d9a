// fa_fwd_kernel_d256.h — gfx950 FlashAttention forward for head dims 129 .. 256 (head-dim tile 256), round 3.
//
// The role hopper/tile_size.h:20-45 plays for d192 / d256 (tiles of their own): the 256-row / two-q-blocks-per-wave shape of
// fa_fwd_kernel_w64.h does not fit -- O for two q-blocks alone would be 256 accumulator registers -- and the compiler-scheduled
// fwd_kernel<T, 256, 4> (register-staged K/V, serial QK -> softmax -> PV per tile) ran at 0.15 - 0.23 of peak.  This kernel:
//
//   * workgroup = 4 waves (one per SIMD), BLOCK_M = 128: a wave owns ONE 32-row q-block; O (8 x 16) and Q (16 x 4) live in
//     AGPRs, the scores / probabilities of a 32-key half-step in arch VGPRs;
//   * the 64-key K/V tile (32 KiB each, 512-byte rows, the XOR swizzle of lds_off<256>) is consumed as two 32-key half-steps:
//         phase 1   MFMA: S(j+1) = K.Q^T                 (16 k-steps)
//         phase 2   MFMA: O += V(j)^T.P(j)  (16 steps)   VALU: exp / sum / pack of S(j+1) -> P(j+1)
//     (the VALU per MFMA is half that of the 128-wide loop: matrix-bound);
//   * K/V tiles by LDS-DMA through raw buffer descriptors (rows past the end of the sequence land as zeros), 2-deep K and V
//     rings [K0 K1 V0 V1] = 128 KiB: K tile t+2 and V tile t+1 are requested during tile t and have landed at its end (one
//     barrier per tile); K tiles are staged shifted by 32 keys so that both score halves formed during V tile t come from K
//     tile t+1;
//   * the steady state (runs of half-steps that need no mask) is the generated asm block fa::FastLoop256<T>
//     (fa_fwd_loop_d256_gen.h, tools/gen_fwd_loop_d256.py); masks, tails and guard trips go through generic_half below, which
//     shares the LDS images and the pipeline state with it.
// Features: dense / varlen / seqused / leftpad, causal and sliding windows, GQA, softcap (SOFTCAP instantiations), ALiBi (ALIBI
// instantiations), a V head dim of its own.  Dropout, ALiBi together with softcap, paged caches and split-KV keep the fwd_kernel<T, 256, 4> instantiations (fa_fwd_api.hip).
#pragma once

#include "fa_fwd_kernel_w64.h"
#ifdef FA_LOOP_D256_GEN_HEADER  /* developer-only: a timing-ablation variant of the generated loop */
#include FA_LOOP_D256_GEN_HEADER
#else
#include "fa_fwd_loop_d256_gen.h"
#endif

namespace fa {

// eight consecutive 1-KiB LDS-DMA pieces of one tile (this wave's share): pieces 0..3 at M0 = lds, 4..7 at M0 = lds + 4096; the
// instruction offset steps the LDS target inside a group and enters the source address too (voff carries -1024 per step)
__device__ __forceinline__ void dma_tile_d256(uint32_t lds, u32x4 desc, uint32_t soff, const uint32_t (&voff)[8]) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %5, %1, %4 offen lds\n\t"
                 "buffer_load_dwordx4 %6, %1, %4 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %7, %1, %4 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %8, %1, %4 offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %9, %1, %4 offen lds\n\t"
                 "buffer_load_dwordx4 %10, %1, %4 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %11, %1, %4 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %12, %1, %4 offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(desc), "s"(lds), "s"(lds + 4096), "s"(soff), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]),
                   "v"(voff[4]), "v"(voff[5]), "v"(voff[6]), "v"(voff[7])
                 : "memory");
}

// DEFF: 256, or 192 / 160 when the head dim is <= 192 / <= 160 -- the k-steps and O blocks of the zero padding are skipped
// (12 + 12 / 10 + 10 instead of 16 + 16 MFMAs per half-step; the LDS images keep their 512-byte rows)
// SOFTCAP: scores = softcap * tanh(q.k * softmax_scale / softcap) (set_params_fprop csrc/flash_attn/flash_api.cpp:103-117; Gemma-2's
// head dim 256 + softcap): the generated block caps the fresh scores in place (FastLoop256<T, DEFF, true>), the generic
// half-step right behind its score product.
// ALIBI: bias -slope |row + sk - sq - key| on the score (csrc/flash_attn/src/alibi.h:18-71), in the block by three VALU per score
// (FastLoop256<T, DEFF, false, true>), in the generic half-step between the cap and the mask (src/mask.h order).  Not with SOFTCAP.
template <typename T, int DEFF, bool SOFTCAP = false, bool ALIBI = false>
__global__ __launch_bounds__(256, 1) void fwd_kernel_d256(const KParams p) {
    static_assert(!(SOFTCAP && ALIBI), "one of the two");
    constexpr int D = 256;
    constexpr int KS_EFF = DEFF / 16, DB_EFF = DEFF / 32;
    constexpr int BLOCK_M = 128;
    constexpr int KSTEPS = D / 16;
    constexpr int DBLOCKS = D / 32;
    constexpr int CH_PER_ROW = D / 8;
    constexpr int ROWB = D * 2;
    constexpr int TILE_BYTES = BLOCK_N * ROWB;   // 32 KiB
    constexpr int O_ROW_BYTES = D * 2 + 16;
    constexpr float THR = (float)FA_RESCALE_THR;
    constexpr float LIM = (float)(1u << (int)THR);

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [K0 | K1 | V0 | V1]; Q is staged through V0 / V1, O through all

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int hh = lane >> 5;

    int m_block, head, batch, split;
    if (!decode_tile(p, m_block, head, batch, split)) return;
    const int kv_head = head / p.h_ratio;

    int sq, sk;
    int64_t q_base, k_base, v_base, o_base, lse_base;
    if (p.cu_seqlens_q) {
        const int q0 = p.cu_seqlens_q[batch];
        sq = p.seqused_q ? p.seqused_q[batch] : p.cu_seqlens_q[batch + 1] - q0;
        q_base = (int64_t)q0 * p.q_row_stride;
        o_base = (int64_t)q0 * p.o_row_stride;
        lse_base = (int64_t)head * p.total_q + q0;
    } else {
        sq = p.seqused_q ? p.seqused_q[batch] : p.seqlen_q;
        q_base = (int64_t)batch * p.q_batch_stride;
        o_base = (int64_t)batch * p.o_batch_stride;
        lse_base = ((int64_t)batch * p.h + head) * p.seqlen_q;
    }
    if (p.cu_seqlens_k) {
        const int k0 = p.cu_seqlens_k[batch];
        sk = p.seqused_k ? p.seqused_k[batch] : p.cu_seqlens_k[batch + 1] - k0;
        k_base = (int64_t)k0 * p.k_row_stride;
        v_base = (int64_t)k0 * p.v_row_stride;
    } else {
        sk = p.seqused_k ? p.seqused_k[batch] : p.seqlen_k;
        const int kv_batch = p.kv_batch_idx ? p.kv_batch_idx[batch] : batch;
        k_base = (int64_t)kv_batch * p.k_batch_stride;
        v_base = (int64_t)kv_batch * p.v_batch_stride;
    }
    const int row_lo = m_block * BLOCK_M;
    if (row_lo >= sq) return;
    if (p.leftpad_k) {
        const int lp = p.leftpad_k[batch];
        sk = max(sk - lp, 0);
        k_base += (int64_t)lp * p.k_row_stride;
        v_base += (int64_t)lp * p.v_row_stride;
    }
    const T *qp = (const T *)p.q + q_base + (int64_t)head * p.q_head_stride;
    const T *kp = (const T *)p.k + k_base + (int64_t)kv_head * p.k_head_stride;
    const T *vp = (const T *)p.v + v_base + (int64_t)kv_head * p.v_head_stride;
    T *op = (T *)p.o + o_base + (int64_t)head * p.o_head_stride;

    const Scales sc = load_scales(p, batch, kv_head);
    float csc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.scale_log2)));
    float scale_e = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.scale)));
    float vdesc_e = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.v_descale)));
    int k_rs = (int)p.k_row_stride, v_rs = (int)p.v_row_stride;
    asm volatile("" : "+s"(csc), "+s"(scale_e), "+s"(vdesc_e), "+s"(k_rs), "+s"(v_rs));

    const int shift = sk - sq;
    const int row_hi = min(sq, row_lo + BLOCK_M);
    int key_hi = sk, key_lo = 0;
    if (p.window_right >= 0) key_hi = min(sk, row_hi + shift + p.window_right);
    if (p.window_left >= 0) key_lo = max(0, row_lo + shift - p.window_left);
    const int n_min = key_lo / BLOCK_N;
    const int n_max = key_hi > 0 ? (key_hi + BLOCK_N - 1) / BLOCK_N : 0;

    const int wrow = row_lo + wave * 32;  // first row of this wave
    // half-steps [0, jend) this wave computes (half-step j = keys [64 n_min + 32 j, +32)); later ones are fully masked
    int jend = 2 * (n_max - n_min);
    if (p.window_right >= 0) {
        const int last_key = min(sk - 1, wrow + 31 + shift + p.window_right);
        jend = min(jend, last_key >= n_min * BLOCK_N ? (last_key - n_min * BLOCK_N) / 32 + 1 : 0);
    }
    if (wrow >= sq || n_min >= n_max) jend = 0;
    jend = __builtin_amdgcn_readfirstlane(jend);
    const int J = 2 * (n_max - n_min);

    // ---- K/V staging: raw buffer descriptors, lane offsets with the source-side swizzle ------------------------------------
    auto make_desc = [&](const T *base, int rs, int rows, int width) {
        const uint64_t b = (uint64_t)(uintptr_t)base;
        u32x4 dsc;
        dsc[0] = (uint32_t)b;
        dsc[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0: raw buffer
        dsc[2] = rows > 0 ? (uint32_t)(((int64_t)(rows - 1) * rs + width) * 2) : 0u;  // bytes to the end of the last valid row
        dsc[3] = 0x00020000u;
#pragma unroll
        for (int i = 0; i < 4; ++i) dsc[i] = __builtin_amdgcn_readfirstlane(dsc[i]);
        return dsc;
    };
    const int dw = min(p.d, D);
    // (V and O may have a head dim of their own, p.dv: FA3 headdim_v, e.g. 192 / 128 -- its chunks past dv read as zeros, the
    //  O columns they would produce are not stored; p.dv = p.d everywhere else)
    const u32x4 kdesc = make_desc(kp, k_rs, sk, dw), vdesc = make_desc(vp, v_rs, sk, min(p.dv, D));
    // piece i of this wave = LDS bytes [wave * 8 KiB + 1024 i, +1024) of a tile image = rows 16 wave + 2 i, + 1; lane l holds the
    // 16-byte slot l of the piece; head-dim chunks past d read as zeros (offset pushed past num_records)
    uint32_t koff[8], voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int slot = wave * 512 + i * 64 + lane;
        const int row = slot / CH_PER_ROW;
        const int ch = (slot % CH_PER_ROW) ^ (((row & 3) << 2) | ((row >> 2) & 3));   // inverse of lds_off<256>
        const bool in = ch * 8 < p.d, in_v = ch * 8 < p.dv;
        koff[i] = (in ? (uint32_t)(row * k_rs + ch * 8) * 2u : 0x7ffffff0u) - 1024u * (i & 3);
        voff[i] = (in_v ? (uint32_t)(row * v_rs + ch * 8) * 2u : 0x7ffffff0u) - 1024u * (i & 3);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t lds_wave = lds0 + wave * 8192;
    // K tile m = keys [64 m - 32, 64 m + 32).  Tile 0 starts 32 rows in front of the sequence: its first half is never read,
    // its lanes get an out-of-range offset (zeros) instead of a negative soffset
    auto load_k = [&](int m, int buf) {
        const int k0 = m * BLOCK_N - 32;
        if (k0 >= 0) {
            dma_tile_d256(lds_wave + buf * TILE_BYTES, kdesc, (uint32_t)(k0 * k_rs * 2), koff);
        } else {
            uint32_t ko[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int slot = wave * 512 + i * 64 + lane;
                const int row = slot / CH_PER_ROW;
                const int ch = (slot % CH_PER_ROW) ^ (((row & 3) << 2) | ((row >> 2) & 3));
                const bool in = ch * 8 < p.d && row + k0 >= 0;
                ko[i] = (in ? (uint32_t)((row + k0) * k_rs + ch * 8) * 2u : 0x7ffffff0u) - 1024u * (i & 3);
            }
            dma_tile_d256(lds_wave + buf * TILE_BYTES, kdesc, 0u, ko);
        }
    };
    auto load_v = [&](int n, int buf) {
        dma_tile_d256(lds_wave + (2 + buf) * TILE_BYTES, vdesc, (uint32_t)(n * BLOCK_N * v_rs * 2), voff);
    };

    // ---- lane parts of the LDS fragment addresses ------------------------------------------------------------------------
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int kbase = lds_off<D>(r, hh);                                                             // ^ 32 ks
    const int vbase = lds_off<D>(4 * hh + (i16 >> 2), 2 * g1 + ((i16 >> 1) & 1)) + 8 * (i16 & 1);   // ^ (64 db + 32 j2)

    u32x4 qf[KSTEPS];
    f32x16 oa[DBLOCKS];
    float m_run = -INFINITY, l_run = 0.f;

    auto drain_o = [&]() {
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(oa[2]), "+a"(oa[3]), "+a"(oa[4]), "+a"(oa[5]),
                     "+a"(oa[6]), "+a"(oa[7]));
    };
    auto qk_half = [&](int kbuf, int kh, f32x16 &s) {
        const char *base = smem + kbuf * TILE_BYTES + kh * (32 * ROWB);
#pragma unroll
        for (int ks = 0; ks < KS_EFF; ++ks) {
            const u32x4 kf = *(const u32x4 *)(base + (kbase ^ (32 * ks)));
            if (ks == 0) Mfma<T>::s_first_pad(s, kf, qf[ks]);
            else Mfma<T>::s_acc_pad(s, kf, qf[ks]);
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s));  // asm MFMA results -> VALU readers
    };
    auto pv_half = [&](int vbuf, int kh, const u32x4 (&pf)[2]) {
        const char *base = smem + (2 + vbuf) * TILE_BYTES + kh * (32 * ROWB);
#pragma unroll
        for (int db = 0; db < DB_EFF; ++db)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4 vf;
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
                        base + (vbase ^ (64 * db + 32 * j2)) + (16 * st + 8 * j2) * ROWB));
                    const u32x2 t2 = __builtin_bit_cast(u32x2, t);
                    vf[2 * j2] = t2[0];
                    vf[2 * j2 + 1] = t2[1];
                }
                Mfma<T>::o_acc_pad(oa[db], vf, pf[st]);
            }
        drain_o();
    };
    auto half_needs_mask = [&](int j) -> bool {
        const int k0 = n_min * BLOCK_N + 32 * j;
        bool need = (k0 + 32 > sk);
        if (p.window_right >= 0) need = need || (k0 + 31 > wrow + shift + p.window_right);
        if (p.window_left >= 0) need = need || (k0 < wrow + 31 + shift - p.window_left);
        return need;
    };
    auto cap_scores = [&](f32x16 &s) {  // softcap before the mask, as everywhere (src/mask.h order)
        if constexpr (SOFTCAP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = fast_tanh(s[i] * sc.softcap_pre);
        }
    };
    // masked body of the block: this lane's last visible key minus the key base of half-step jn's scores in its lane half
    auto mask_limit = [&](int jn) {
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int rr = wrow + (ln & 31) + shift;
        const int last = p.window_right >= 0 ? min(sk - 1, rr + p.window_right) : sk - 1;
        return last - (n_min * BLOCK_N + 32 * jn + 4 * (ln >> 5));
    };
    const float alibi_raw = ALIBI ? load_alibi(p, sc, batch, head) : 0.f;   // slope in units of the raw score
    auto alibi_scores = [&](int j, f32x16 &s) {
        if constexpr (ALIBI) {
            const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const int rel0 = wrow + (ln & 31) + shift - (n_min * BLOCK_N + 32 * j + 4 * (ln >> 5));
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] -= alibi_raw * fabsf((float)(rel0 - ((i & 3) + 8 * (i >> 2))));
        }
    };
    auto mask_scores = [&](int j, f32x16 &s) {
        if (!half_needs_mask(j)) return;
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int k0 = n_min * BLOCK_N + 32 * j + 4 * (ln >> 5);
        const int rr = wrow + (ln & 31) + shift;  // diagonal key of this lane's row
        int hi = sk, lo = 0;                      // [lo, hi) visible
        if (p.window_right >= 0) hi = min(sk, rr + p.window_right + 1);
        if (p.window_left >= 0) lo = max(0, rr - p.window_left);
        hi -= k0; lo -= k0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2);
            if (key >= hi || key < lo) s[i] = -INFINITY;
        }
    };
    // online softmax of one 32 x 32 score block (lane = query row): P fragments, (m, l); alpha = what O still owes
    auto softmax = [&](f32x16 &s, u32x4 (&pf)[2], float &alpha, bool &moved) {
        float mxa, mxb;
        rowmax16(s, m_run, mxa, mxb);
        const float m_new = half_swap_max(fmaxf(mxa, mxb));
        moved = __any((m_new - m_run) * csc > THR);  // -inf -> finite counts as moved
        const float m_eff = moved ? m_new : m_run;
        const float mc = (m_eff == -INFINITY ? 0.f : m_eff) * csc;
        alpha = __builtin_amdgcn_exp2f(m_run * csc - mc);
        m_run = m_eff;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) pf[i >> 3][(i & 7) >> 1] = Exp2Pair<T>::run(s[i], s[i + 1], csc, mc, ps0, ps1);
        l_run = l_run * alpha + (ps0 + ps1);
    };
    auto rescale = [&](float alpha) {
        drain_o();
#pragma unroll
        for (int db = 0; db < DB_EFF; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) oa[db][i] *= alpha;
        drain_o();
    };

    // ---- prologue: K tiles n_min (slot 0) and n_min + 1 (slot 1); Q through the V slots; first scores ---------------------
    if (n_min < n_max) {
        load_k(n_min, 0);
        load_k(n_min + 1, 1);
    }
    {   // this wave's 32 rows of Q (16 KiB) as a half-tile image at V region + wave * 16 KiB: the K fragment reads read it
        const int rows_here = min(sq - wrow, 32);
        const u32x4 qdesc = make_desc(qp + (int64_t)wrow * p.q_row_stride, (int)p.q_row_stride, rows_here, dw);
        const int q_rs = (int)p.q_row_stride;
        uint32_t qoff[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int slot = i * 64 + lane;
            const int row = slot / CH_PER_ROW;
            const int ch = (slot % CH_PER_ROW) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            qoff[i] = ch * 8 < p.d ? (uint32_t)(row * q_rs + ch * 8) * 2u : 0x7ffffff0u;
        }
        const uint32_t q_img = lds0 + 2 * TILE_BYTES + wave * (32 * ROWB);
#pragma unroll
        for (int i = 0; i < 16; ++i) lds_dma_buf1(q_img + i * 1024, qdesc, qoff[i]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
        const char *qimg = smem + 2 * TILE_BYTES + wave * (32 * ROWB);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *(const u32x4 *)(qimg + (kbase ^ (32 * ks)));
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("; pin Q" : "+a"(qf[ks]));
    {
        const u32x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int db = 0; db < DBLOCKS; ++db) Mfma<T>::o_zero(oa[db], z4);
        drain_o();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave holds its Q fragments: the V slots are free
    if (n_min < n_max) load_v(n_min, 0);

    // Pipeline state in front of half-step j: pc = P(j), (m, l) through j, O owes alpha when moved; after a guard trip
    // (redo) s still holds S(j) and P(j) / l have to be redone with a fresh max.
    f32x16 s;
    u32x4 pc[2], pn[2];
    float alpha = 1.f, l_saved = 0.f;
    bool moved = false, redo = false;
    if (jend > 0) {
        qk_half(0, 1, s);  // half-step 0 = second half of the shifted K tile n_min
        cap_scores(s);
        alibi_scores(0, s);
        mask_scores(0, s);
        softmax(s, pc, alpha, moved);
        moved = false;     // O is still zero
    }
    // V tile n_min has landed (requested above, behind the Q barrier); every wave has read K tile n_min before tile 0 requests
    // K tile n_min + 2 into its slot
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    auto generic_half = [&](int j) {
        const int i = j >> 1, kb = j & 1, slot = i & 1, n = n_min + i;
        if (kb == 0) {  // K tile n+2 over K tile n (last read during tile n-1), V tile n+1 over V tile n-1
            load_k(n + 2, slot);
            load_v(n + 1, slot ^ 1);
        }
        if (j < jend) {
            if (redo) {  // s still holds S(j)
                l_run = l_saved;
                softmax(s, pc, alpha, moved);
                redo = false;
            }
            if (moved) rescale(alpha);
            moved = false;
            if (j + 1 < jend) {
                qk_half(slot ^ 1, kb, s);
                cap_scores(s);
                alibi_scores(j + 1, s);
                mask_scores(j + 1, s);
            }
            pv_half(slot, kb, pc);
            if (j + 1 < jend) {
                softmax(s, pn, alpha, moved);
                pc[0] = pn[0];
                pc[1] = pn[1];
            }
        }
        if (kb == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // last half-step index whose scores need no mask for this wave / first one without a left-window mask
    int fast_last = -1, fast_first = 0;
    if (jend > 0) {
        int nomask = (sk - n_min * BLOCK_N) / 32 - 1;
        if (p.window_right >= 0) {
            const int t = wrow + shift + p.window_right - 31 - n_min * BLOCK_N;
            nomask = min(nomask, t >= 0 ? t / 32 : -1);
        }
        fast_last = min(nomask, jend - 1);
        if (p.window_left >= 0) {
            const int t = wrow + 31 + shift - p.window_left - n_min * BLOCK_N;
            fast_first = t > 0 ? (t + 31) / 32 : 0;
        }
    }
    fast_last = __builtin_amdgcn_readfirstlane(fast_last);
    fast_first = __builtin_amdgcn_readfirstlane(fast_first);
    const bool addr32 = (int64_t)sk * k_rs < (1ll << 30) && (int64_t)sk * v_rs < (1ll << 30);

    int j = 0;
    while (j < J) {
        // the generated block: a run of whole tiles, entered at a tile boundary, whose fresh scores S(j+1) .. S(j + 2 count)
        // all need no mask and are all needed by this wave
        if ((j & 1) == 0 && !moved && !redo && addr32 && j + 1 >= fast_first) {
            int count = (min(fast_last, jend - 1) - j) >> 1;
            // masked body of the block (round 3; not under ALiBi): what is left of this wave's range behind its last mask-free tile
            // -- the diagonal tiles under a causal / right-window mask, the tail tile of a sequence that is not a multiple of 64, the
            // last tile of any sweep (its look-ahead scores lie behind the end) -- as whole tiles: a trailing half-step the wave
            // does not need is fully masked (P = 0, row sums 0)
            const bool masked_run = !ALIBI && count < 1 && j < jend;
            if (masked_run) count = (jend - j + 1) >> 1;
            if (count >= 1 && !__any(m_run == -INFINITY)) {
                const int n_cur = n_min + (j >> 1);
                uint32_t ktile = (uint32_t)(((n_cur + 2) * BLOCK_N - 32) * k_rs * 2);
                uint32_t vtile = (uint32_t)((n_cur + 1) * BLOCK_N * v_rs * 2);
                int done = 0;
                uint64_t redo_mask = 0;
                const uint32_t kstep_ = (uint32_t)(BLOCK_N * k_rs * 2), vstep_ = (uint32_t)(BLOCK_N * v_rs * 2);
                const int slot0_ = (j >> 1) & 1;
                const float mc_ = m_run * csc;
                const int masked_ = __builtin_amdgcn_readfirstlane(masked_run ? 1 : 0);   // (an SGPR operand of the block)
                if constexpr (SOFTCAP) {
                    float cap2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.softcap_pre * 2.885390081777927f)));
                    FastLoop256<T, DEFF, true>::run(oa, qf, s, pc, pn, l_run, l_saved, mc_, (uint32_t)kbase, (uint32_t)vbase, koff, voff,
                                        csc, LIM, kdesc, vdesc, ktile, vtile, kstep_, vstep_, lds0, lds_wave, slot0_, count, done, redo_mask, cap2,
                                        masked_, mask_limit(j + 1));
                } else if constexpr (ALIBI) {
                    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    const int rrel = wrow + (ln & 31) + shift - (n_min * BLOCK_N + 32 * (j + 1) + 4 * (ln >> 5));
                    float aslope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, alibi_raw)));
                    FastLoop256<T, DEFF, false, true>::run(oa, qf, s, pc, pn, l_run, l_saved, mc_, (uint32_t)kbase, (uint32_t)vbase, koff, voff,
                                        csc, LIM, kdesc, vdesc, ktile, vtile, kstep_, vstep_, lds0, lds_wave, slot0_, count, done, redo_mask, aslope, rrel);
                } else {
                    FastLoop256<T, DEFF>::run(oa, qf, s, pc, pn, l_run, l_saved, mc_, (uint32_t)kbase, (uint32_t)vbase, koff, voff,
                                        csc, LIM, kdesc, vdesc, ktile, vtile, kstep_, vstep_, lds0, lds_wave, slot0_, count, done, redo_mask,
                                        masked_, mask_limit(j + 1));
                }
                j += done;
                redo = redo_mask != 0;
                if (done & 1) {  // the fresh P sits in the odd buffer
                    pc[0] = pn[0];
                    pc[1] = pn[1];
                    // the tile's first half-step is done (its LDS-DMA requests are out): finish the tile generically
                }
                if (done > 0) continue;
            }
        }
        generic_half(j);
        ++j;
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // all LDS-DMA landed: the rings can be reused
    drain_o();
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int r_e = lane_e & 31, hh_e = lane_e >> 5;
    const int row_e = wrow + r_e;
    const float lt = half_swap_sum(l_run);
    const bool empty = (lt == 0.f) || (lt != lt);
    const float inv = (empty ? 1.f : 1.f / lt) * vdesc_e;
    const bool wave_active = wrow < sq;
    if (wave_active) {
        if (hh_e == 0 && row_e < sq) p.lse[lse_base + row_e] = empty ? INFINITY : m_run * scale_e + __logf(lt);
        char *obuf = smem + wave * (32 * O_ROW_BYTES);
#pragma unroll
        for (int db = 0; db < DB_EFF; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w[0] = Elem<T>::pack2(oa[db][4 * g4] * inv, oa[db][4 * g4 + 1] * inv);
                w[1] = Elem<T>::pack2(oa[db][4 * g4 + 2] * inv, oa[db][4 * g4 + 3] * inv);
                *(u32x2 *)(obuf + r_e * O_ROW_BYTES + (db * 32 + 8 * g4 + 4 * hh_e) * 2) = w;
            }
        // (a wave reads back only its own 32 staged rows: LDS operations of one wave are in order)
        constexpr int NCH = (32 * CH_PER_ROW) / 64;
        u32x4 val[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane_e + i * 64;
            val[i] = *(const u32x4 *)(obuf + (c / CH_PER_ROW) * O_ROW_BYTES + (c % CH_PER_ROW) * 16);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane_e + i * 64;
            const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
            if (wrow + row < sq && ch * 8 < p.dv) *(u32x4 *)(op + (int64_t)(wrow + row) * p.o_row_stride + ch * 8) = val[i];
        }
    }
}

constexpr int smem_bytes_d256() { return 4 * BLOCK_N * 256 * 2; }  // K and V rings (2 x 32 KiB each); Q and O staging alias them

}  // namespace fa
