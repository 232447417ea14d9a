// fa_fwd_kernel_fp8.h — gfx950 FlashAttention forward on e4m3 inputs, native: both products on the block-scaled fp8 MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4, unit scales, 2x the bf16 rate per clock), K/V/Q read from HBM as they are (1 byte per
// element, no expansion pass), bf16 output.  BASELINE config 5; role of the fp8 branch of
// hopper/mainloop_fwd_sm90_tma_gmma_ws.hpp (V transpose :702-739, P re-quantised :1157-1160) and hopper/softmax.h:67-69,146-151
// (max offset), descales folded as hopper/flash_fwd_kernel_sm90.h:408-415.
//
// Shape: as fwd_kernel_w64 -- 4 waves (one per SIMD), BLOCK_M = 256, each wave two 32-row q-blocks A / B, 64-key tiles,
// 3-deep K and V rings in LDS, one barrier per tile -- with the 64-key tile as the pipeline step (the PV product contracts
// over 64 keys per MFMA):
//     phase 1   MFMA: S_A(n+1), S_B(n+1) = K(n+1).Q^T            VALU: softmax of S_B(n)    -> P_B(n)  (e4m3)
//     phase 2   MFMA: O_A += V(n)^T P_A(n), O_B += V(n)^T P_B(n)   VALU: softmax of S_A(n+1)  -> P_A(n+1)
// The steady state is the generated asm block of fa_fwd_loop_fp8_gen.h (tools/gen_fwd_loop_fp8.py, which documents the
// operand maps); this file is the prologue, the boundary tiles (masks, tails, rescales) and the epilogue around it.
//
// Layouts:
//   * LDS tile image: 64 rows x 128 B, 16-byte chunk c of row r at r * 128 + 16 * (c ^ swz(r)), swz(r) = (r & 6) ^ ((r >> 3) & 1):
//     both the ds_read_b128 row reads of the K operand and the ds_read_b64_tr_b8 transposed reads of the V operand spread
//     over all banks.
//   * score block beta of a tile = keys 32 hm + 16 beta + 4 a + b for MFMA row m = 8 a + 4 hm + b: register i of lane half h
//     is key 32 h + 16 beta + i, so the packed probabilities are the PV product's B operand in natural key order.
//   * K/V tiles arrive by buffer_load_dwordx4 ... lds through raw buffer descriptors: rows past the end of the sequence read
//     as zeros (no clamping anywhere in this kernel).
// Numerics: P' = exp2(s c - m c + OFF) with OFF = 4 and stale-max threshold THR = 4 (P' <= 2^8 < 448 = e4m3 max); l carries
// 2^OFF as well and it cancels in O / l; LSE = m scale + log(l) - OFF ln 2.  (Round 2 ran OFF 5 / THR 3: on N(0,1) data a row's
// maximum outgrows its first tile's by more than 2^3 in ~4 % of the rows, i.e. nearly every wave left the generated block 4 - 5
// times per sweep -- counted with -DFA_F8_DEBUG, tools/f8_debug.py -- and every exit stalls the workgroup's other waves at the
// tile barrier: 11 - 16 % of the kernel.  At THR 4 that is 0.5 exits per wave; probabilities below 2^-13 of the row's stale
// maximum now flush to zero instead of 2^-14, against the 2^-9 of an unshifted e4m3 P as the reference's oracle rounds it.)
#pragma once

#include "fa_fwd_kernel_w64.h"

namespace fa {
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
}

#ifdef FA_LOOP_FP8_GEN_HEADER  /* developer-only: a timing-ablation variant of the generated loop */
#include FA_LOOP_FP8_GEN_HEADER
#else
#include "fa_fwd_loop_fp8_gen.h"
#endif

namespace fa {

#ifndef FA_FP8_OFF
#define FA_FP8_OFF 4
#endif
#ifndef FA_FP8_THR
#define FA_FP8_THR 4
#endif

__device__ __forceinline__ int swz8(int row) { return (row & 6) ^ ((row >> 3) & 1); }

struct MfmaF8 {
    // hipcc pads nothing around asm MFMAs: s_nop 1 in front covers VALU-written operands, callers drain before VALU reads results
    static __device__ __forceinline__ void s_first(f32x16 &d, u32x8 k, const u32x8 &q, uint32_t one) {
        asm("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]" : "=&v"(d) : "v"(k), "v"(q), "v"(one));
    }
    static __device__ __forceinline__ void s_acc(f32x16 &d, u32x8 k, const u32x8 &q, uint32_t one) {
        asm("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(k), "v"(q), "v"(one));
    }
    static __device__ __forceinline__ void o_acc(f32x16 &o, u32x8 vf, u32x8 pf, uint32_t one) {
        asm("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(o) : "v"(vf), "v"(pf), "v"(one));
    }
};

// two e4m3 bytes into the low / high half of a dword (RNE, saturating at +-448)
__device__ __forceinline__ void cvt_pk_fp8_lo(uint32_t &d, float a, float b) { asm("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); }
__device__ __forceinline__ void cvt_pk_fp8_hi(uint32_t &d, float a, float b) { asm("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(d) : "v"(a), "v"(b)); }

// one tile's LDS-DMA pieces of this wave: piece i -> LDS m0 + 1024 i, source base + soff + voff[i] + 1024 i (the instruction
// offset enters both addresses: voff carries -1024 i, tools/probe_bufload.hip)
__device__ __forceinline__ void dma_tile_f8(uint32_t lds, u32x4 desc, uint32_t soff, const uint32_t (&voff)[2]) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t"
                 "buffer_load_dwordx4 %5, %1, %3 offen offset:1024 lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(desc), "s"(lds), "s"(soff), "v"(voff[0]), "v"(voff[1]) : "memory");
}

#ifdef FA_F8_DEBUG  /* developer-only (tools/f8_debug.py): [0] runs of the generated block, [1] tiles they completed, [2] runs ended by pend, [3] by tripb */
__device__ unsigned long long fa_f8_dbg[8];
#endif

__global__ __launch_bounds__(256, 1) void fwd_kernel_fp8(const KParams p) {
    constexpr int D = 128;
    constexpr int BLOCK_M = 256;
    constexpr int TILE_BYTES = BLOCK_N * 128;   // 8 KiB
    constexpr int O_ROW_BYTES = D * 2 + 16;
    constexpr float THR = (float)FA_FP8_THR, OFF = (float)FA_FP8_OFF;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [K0 K1 K2 V0 V1 V2]; the epilogue stages O over it

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
        k_base = (int64_t)batch * p.k_batch_stride;
        v_base = (int64_t)batch * p.v_batch_stride;
    }
    const int row_lo = m_block * BLOCK_M;
    if (row_lo >= sq) return;

    const uint8_t *qp = (const uint8_t *)p.q + q_base + (int64_t)head * p.q_head_stride;
    const uint8_t *kp = (const uint8_t *)p.k + k_base + (int64_t)kv_head * p.k_head_stride;
    const uint8_t *vp = (const uint8_t *)p.v + v_base + (int64_t)kv_head * p.v_head_stride;
    __bf16 *op = (__bf16 *)p.o + o_base + (int64_t)head * p.o_head_stride;

    const Scales sc = load_scales(p, batch, kv_head);
    float csc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.scale_log2)));
    float scale_e = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.scale)));
    float vdesc_e = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.v_descale)));
    int k_rs = (int)p.k_row_stride, v_rs = (int)p.v_row_stride;
    asm volatile("" : "+s"(csc), "+s"(scale_e), "+s"(vdesc_e), "+s"(k_rs), "+s"(v_rs));

    const int shift = sk - sq;
    const int row_hi = min(sq, row_lo + BLOCK_M);
    int key_hi = sk;
    if (p.window_right >= 0) key_hi = min(sk, row_hi + shift + p.window_right);
    const int n_max = key_hi > 0 ? (key_hi + BLOCK_N - 1) / BLOCK_N : 0;  // tiles [0, n_max) (no left window on this path)

    const int wrow = row_lo + wave * 64;
    const int row_a = wrow + r, row_b = wrow + 32 + r;
    // tiles [0, tend) this wave computes; later ones are fully masked for all of its 64 rows
    int tend = n_max;
    if (p.window_right >= 0) {
        const int last_key = min(sk - 1, wrow + 63 + shift + p.window_right);
        tend = min(tend, last_key >= 0 ? last_key / BLOCK_N + 1 : 0);
    }
    if (wrow >= sq) tend = 0;
    tend = __builtin_amdgcn_readfirstlane(tend);

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (row r, half hh): bytes 64 st + 32 hh .. + 32 of its row -------------
    u32x8 qa[2], qb[2];
    {
        const uint8_t *qra = qp + (int64_t)min(row_a, sq - 1) * p.q_row_stride + 32 * hh;
        const uint8_t *qrb = qp + (int64_t)min(row_b, sq - 1) * p.q_row_stride + 32 * hh;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const u32x4 a0 = *(const u32x4 *)(qra + 64 * st), a1 = *(const u32x4 *)(qra + 64 * st + 16);
            const u32x4 b0 = *(const u32x4 *)(qrb + 64 * st), b1 = *(const u32x4 *)(qrb + 64 * st + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qa[st][e] = a0[e]; qa[st][4 + e] = a1[e];
                qb[st][e] = b0[e]; qb[st][4 + e] = b1[e];
            }
        }
    }
    f32x16 oa[4], ob[4];
    {
        const u32x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            Mfma<__bf16>::o_zero(oa[db], z4);
            Mfma<__bf16>::o_zero(ob[db], z4);
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(oa[2]), "+a"(oa[3]), "+a"(ob[0]), "+a"(ob[1]), "+a"(ob[2]), "+a"(ob[3]));
    }
    float m_a = -INFINITY, m_b = -INFINITY, l_a = 0.f, l_b = 0.f;
    uint32_t one = 0x7f7f7f7fu;  // E8M0 block scales 2^0
    asm volatile("" : "+v"(one));

    // ---- K/V staging: raw buffer descriptors, lane offsets with the source-side swizzle ----------------------------------
    auto make_desc = [&](const uint8_t *base, int rs) {
        const uint64_t b = (uint64_t)(uintptr_t)base;
        u32x4 dsc;
        dsc[0] = (uint32_t)b;
        dsc[1] = (uint32_t)(b >> 32) & 0xffffu;
        dsc[2] = sk > 0 ? (uint32_t)((int64_t)(sk - 1) * rs + D) : 0u;  // bytes up to the end of the last valid row
        dsc[3] = 0x00020000u;
#pragma unroll
        for (int i = 0; i < 4; ++i) dsc[i] = __builtin_amdgcn_readfirstlane(dsc[i]);
        return dsc;
    };
    const u32x4 kdesc = make_desc(kp, k_rs), vdesc = make_desc(vp, v_rs);
    uint32_t koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (2 * wave + i) + (lane >> 3);
        const int ch = (lane & 7) ^ swz8(row);
        koff[i] = (uint32_t)(row * k_rs + 16 * ch) - 1024u * i;
        voff[i] = (uint32_t)(row * v_rs + 16 * ch) - 1024u * i;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t lds_wave = lds0 + wave * 2048;
    auto load_k = [&](int n, int buf) { dma_tile_f8(lds_wave + buf * TILE_BYTES, kdesc, (uint32_t)(n * BLOCK_N * k_rs), koff); };
    auto load_v = [&](int n, int buf) { dma_tile_f8(lds_wave + (3 + buf) * TILE_BYTES, vdesc, (uint32_t)(n * BLOCK_N * v_rs), voff); };

    // ---- lane parts of the LDS fragment addresses ----------------------------------------------------------------------
    const int krow0 = 32 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);              // key row of MFMA row r in score block 0
    const int kbase = krow0 * 128 + 16 * ((2 * hh) ^ swz8(krow0));              // ^ (64 st + 16 e), + 2048 beta
    const int i16 = lane & 15, gi = (lane >> 4) & 1, vb = i16 >> 1, vp8 = i16 & 1;
    const int vbase = (32 * hh + vb) * 128 + 16 * (gi ^ (vb & 6)) + 8 * vp8;     // ^ 16 (2 db | (t & 1)), + 1024 t

    auto qk_tile = [&](int kbuf, f32x16 (&sa_)[2], f32x16 (&sb_)[2]) {
        const char *base = smem + kbuf * TILE_BYTES;
#pragma unroll
        for (int beta = 0; beta < 2; ++beta)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const u32x4 k0 = *(const u32x4 *)(base + 2048 * beta + (kbase ^ (64 * st)));
                const u32x4 k1 = *(const u32x4 *)(base + 2048 * beta + (kbase ^ (64 * st + 16)));
                u32x8 kf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { kf[e] = k0[e]; kf[4 + e] = k1[e]; }
                if (st == 0) { MfmaF8::s_first(sa_[beta], kf, qa[0], one); MfmaF8::s_first(sb_[beta], kf, qb[0], one); }
                else { MfmaF8::s_acc(sa_[beta], kf, qa[1], one); MfmaF8::s_acc(sb_[beta], kf, qb[1], one); }
                __builtin_amdgcn_sched_barrier(0);  // (boundary code: keep hipcc from hoisting every fragment read -> spills)
            }
    };
    auto pv_tile = [&](int vbuf, const u32x8 &pa, const u32x8 &pb) {
        const char *base = smem + (3 + vbuf) * TILE_BYTES;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            u32x8 vf;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const auto x = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2_t *)(
                    base + (vbase ^ (16 * (2 * db + (t & 1)))) + 1024 * t));
                vf[2 * t] = (uint32_t)x[0];
                vf[2 * t + 1] = (uint32_t)x[1];
            }
            MfmaF8::o_acc(oa[db], vf, pa, one);
            MfmaF8::o_acc(ob[db], vf, pb, one);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto drain_scores = [&](f32x16 (&s0)[2], f32x16 (&s1)[2]) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(s0[0]), "+v"(s0[1]), "+v"(s1[0]), "+v"(s1[1]));
    };
    auto drain_all = [&]() {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(oa[2]), "+a"(oa[3]), "+a"(ob[0]), "+a"(ob[1]), "+a"(ob[2]), "+a"(ob[3]));
    };
    auto rescale = [&](f32x16 (&o)[4], float alpha) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]));
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]));
    };
    // mask of the scores of tile n (keys 64 n + 32 hh + 16 beta + i in register i of block beta); boundary tiles only
    auto tile_needs_mask = [&](int n) -> bool {
        const int k0 = n * BLOCK_N;
        bool need = (k0 + BLOCK_N > sk);
        if (p.window_right >= 0) need = need || (k0 + BLOCK_N - 1 > wrow + shift + p.window_right);
        return need;
    };
    auto mask_scores = [&](int n, f32x16 (&sa_)[2], f32x16 (&sb_)[2]) {
        if (!tile_needs_mask(n)) return;
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int k0 = n * BLOCK_N + 32 * (ln >> 5);
        const int ra = wrow + (ln & 31) + shift, rb = ra + 32;
        int hi_a = sk, hi_b = sk;
        if (p.window_right >= 0) {
            hi_a = min(sk, ra + p.window_right + 1);
            hi_b = min(sk, rb + p.window_right + 1);
        }
        hi_a -= k0; hi_b -= k0;
#pragma unroll
        for (int beta = 0; beta < 2; ++beta)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 16 * beta + i;
                if (key >= hi_a) sa_[beta][i] = -INFINITY;
                if (key >= hi_b) sb_[beta][i] = -INFINITY;
            }
    };
    auto rowmax32 = [&](const f32x16 (&s)[2], float m) -> float {
        float x0, x1, y0, y1;
        rowmax16(s[0], m, x0, x1);
        rowmax16(s[1], m, y0, y1);
        return half_swap_max(fmaxf(fmaxf(x0, x1), fmaxf(y0, y1)));
    };
    // online softmax of one 64-key score tile of a q-block (lane = query row, 32 of its keys); P' packed to e4m3
    auto softmax = [&](f32x16 (&s)[2], u32x8 &pf, float &m_run, float &l_run, float &alpha, bool &moved) {
        const float m_new = rowmax32(s, m_run);
        moved = __any((m_new - m_run) * csc > THR);  // (-inf -> finite counts as moved)
        const float m_eff = moved ? m_new : m_run;
        const float mc = (m_eff == -INFINITY ? 0.f : m_eff) * csc - OFF;
        alpha = __builtin_amdgcn_exp2f(m_run * csc - (m_eff == -INFINITY ? 0.f : m_eff) * csc);
        m_run = m_eff;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int beta = 0; beta < 2; ++beta)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_exp2f(s[beta][4 * g + e] * csc - mc);
                ps0 += t[0] + t[2];
                ps1 += t[1] + t[3];
                uint32_t w = 0;
                cvt_pk_fp8_lo(w, t[0], t[1]);
                cvt_pk_fp8_hi(w, t[2], t[3]);
                pf[4 * beta + g] = w;
                __builtin_amdgcn_sched_barrier(0);
            }
        l_run = l_run * alpha + (ps0 + ps1);
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------
    if (n_max > 0) {
        load_k(0, 0);
        load_v(0, 0);
        load_k(1, 1);
        load_k(2, 2);
        load_v(1, 1);
    }
    {
        const u32x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            qa[st] = row_a < sq ? qa[st] : z8;
            qb[st] = row_b < sq ? qb[st] : z8;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // The pipeline state between tiles -- S_B(n) (32 raw scores per lane) and P_A(n) (8 dwords) -- lives in a per-wave LDS
    // hand-off area behind the K/V rings (10 x 16 B per lane, chunk c of lane l at st + 1024 c), never in loop-carried
    // registers: the generated block reads / writes it there, the boundary code loads it where it needs it.  (As 40
    // loop-carried registers it was spilled to scratch around every entry of the block, whose register map is fixed.)
    char *st = smem + 6 * TILE_BYTES + wave * (10 * 1024) + lane * 16;
    auto state_store = [&](const f32x16 (&sb)[2], const u32x8 &pa) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            u32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(sb[c >> 2][4 * (c & 3) + e]);
            *(u32x4 *)(st + 1024 * c) = w;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            u32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pa[4 * c + e];
            *(u32x4 *)(st + 1024 * (8 + c)) = w;
        }
    };
    auto state_load_sb = [&](f32x16 (&sb)[2]) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const u32x4 w = *(const u32x4 *)(st + 1024 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) sb[c >> 2][4 * (c & 3) + e] = __uint_as_float(w[e]);
        }
    };
    auto state_load_pa = [&](u32x8 &pa) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u32x4 w = *(const u32x4 *)(st + 1024 * (8 + c));
#pragma unroll
            for (int e = 0; e < 4; ++e) pa[4 * c + e] = w[e];
        }
    };
    float alpha_a = 1.f, alpha_b = 1.f;
    bool moved_a = false, moved_b = false;
    if (tend > 0) {
        f32x16 sa[2], sb[2];
        u32x8 pa;
        qk_tile(0, sa, sb);
        drain_scores(sa, sb);
        mask_scores(0, sa, sb);
        softmax(sa, pa, m_a, l_a, alpha_a, moved_a);
        moved_a = false;  // O_A is still zero
        m_b = rowmax32(sb, m_b);  // B's running max starts at the max of its first tile (l_b stays 0)
        state_store(sb, pa);
    }
    __syncthreads();  // every wave has read K tile 0 before tile 0's DMA overwrites its slot

    // ---- generic tile: masks, the last tiles of the wave, rescales (serial: softmax B, P.V, next scores, softmax A) -----
    auto generic_tile = [&](int i) {
        const int slot = i % 3, slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot == 0 ? 2 : slot - 1;
        load_k(i + 3, slot);    // K tile i+3 over K tile i (last read during tile i-1)
        load_v(i + 2, slot2);   // V tile i+2 over V tile i-1
        if (i < tend) {
            if (moved_a) rescale(oa, alpha_a);
            moved_a = false;
            {
                f32x16 sb[2];
                u32x8 pa, pb;
                state_load_sb(sb);
                softmax(sb, pb, m_b, l_b, alpha_b, moved_b);
                if (moved_b) rescale(ob, alpha_b);
                state_load_pa(pa);
                pv_tile(slot, pa, pb);
                drain_all();
            }
            if (i + 1 < tend) {
                f32x16 sa[2], sb[2];
                u32x8 pa;
                qk_tile(slot1, sa, sb);
                drain_scores(sa, sb);
                mask_scores(i + 1, sa, sb);
                softmax(sa, pa, m_a, l_a, alpha_a, moved_a);
                state_store(sb, pa);
            }
        }
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // last tile index whose scores need no mask for this wave; a fast tile i computes the scores of tile i+1
    int fast_last = -1;
    if (tend > 0) {
        int nomask = sk / BLOCK_N - 1;
        if (p.window_right >= 0) {
            const int t = wrow + shift + p.window_right - (BLOCK_N - 1);
            nomask = min(nomask, t >= 0 ? t / BLOCK_N : -1);
        }
        fast_last = min(nomask, tend - 1);
    }
    fast_last = __builtin_amdgcn_readfirstlane(fast_last);

    int i = 0;
    while (i < n_max) {
        // tiles i .. fast_last-1 (each computes the unmasked scores of its successor), plus -- when the wave's LAST tile is
        // itself unmasked -- that tile as a `phantom` call of its own: the block then forms S(tend) / P_A(tend) from
        // whatever K tile follows (real keys behind a causal diagonal, or the zeros that rows past the end read as) with the
        // look-ahead guards off (threshold +inf); nothing of that is used, and l_a is restored from l_a_saved.
        const bool last_unmasked = fast_last == tend - 1 && tend > 0;
        const bool phantom = last_unmasked && i == tend - 1;
        int count = phantom ? 1 : fast_last - i;
        if (count >= 1 && !moved_a && (int64_t)sk * k_rs < (1ll << 31) && (int64_t)sk * v_rs < (1ll << 31) &&
            !__any(m_a == -INFINITY || m_b == -INFINITY)) {
            {   // S_B(i) must be safe to exponentiate with the stale m_b (inside the block the look-ahead guarantees it)
                f32x16 sb[2];
                state_load_sb(sb);
                const float m_new = rowmax32(sb, m_b);
                if (__any((m_new - m_b) * csc > THR)) {
                    const float al = __builtin_amdgcn_exp2f((m_b - m_new) * csc);
                    rescale(ob, al);
                    l_b *= al;
                    m_b = m_new;
                }
            }
            int done = 0;
            uint64_t pend = 0, tripb = 0;
            float ala = 1.f, l_a_saved = l_a;
            FastLoopFp8::run(oa, ob, qa, qb, lds0 + 6 * TILE_BYTES + wave * (10 * 1024) + lane * 16, l_a, l_b, m_a, ala, l_a_saved,
                             m_a * csc - OFF, m_b * csc - OFF, m_b, (uint32_t)kbase, (uint32_t)vbase, koff, voff, csc,
                             phantom ? INFINITY : THR / csc, OFF, kdesc, vdesc,
                             (uint32_t)((i + 3) * BLOCK_N * k_rs), (uint32_t)((i + 2) * BLOCK_N * v_rs),
                             (uint32_t)(BLOCK_N * k_rs), (uint32_t)(BLOCK_N * v_rs), lds0, lds_wave, i % 3, count, done,
                             pend, tripb);
            i += done;
#ifdef FA_F8_DEBUG
            if (lane == 0) {
                atomicAdd(&fa_f8_dbg[0], 1ull);
                atomicAdd(&fa_f8_dbg[1], (unsigned long long)done);
                if (pend != 0) atomicAdd(&fa_f8_dbg[2], 1ull);
                if (tripb != 0) atomicAdd(&fa_f8_dbg[3], 1ull);
            }
#endif
            if (phantom) l_a = l_a_saved;
            if (pend != 0) rescale(oa, ala);
            (void)tripb;  // q-block B's new max is taken at the top of the next iteration (or by the generic tile)
            continue;
        }
        generic_tile(i);
        ++i;
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // all LDS-DMA landed: the rings can be reused
    if (moved_a) rescale(oa, alpha_a);
    drain_all();
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int r_e = lane_e & 31, hh_e = lane_e >> 5;
    const int row_a_e = wrow + r_e, row_b_e = wrow + 32 + r_e;
    const float lt_a = half_swap_sum(l_a), lt_b = half_swap_sum(l_b);
    const bool e_a = (lt_a == 0.f) || (lt_a != lt_a), e_b = (lt_b == 0.f) || (lt_b != lt_b);
    const float inv_a = (e_a ? 1.f : 1.f / lt_a) * vdesc_e, inv_b = (e_b ? 1.f : 1.f / lt_b) * vdesc_e;
    const bool wave_active = wrow < sq;
    constexpr float OFF_LN2 = OFF * 0.6931471805599453f;
    if (wave_active) {
        if (hh_e == 0) {
            if (row_a_e < sq) p.lse[lse_base + row_a_e] = e_a ? INFINITY : m_a * scale_e + __logf(lt_a) - OFF_LN2;
            if (row_b_e < sq) p.lse[lse_base + row_b_e] = e_b ? INFINITY : m_b * scale_e + __logf(lt_b) - OFF_LN2;
        }
        char *obuf = smem + wave * (64 * O_ROW_BYTES);
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 wa, wb;
                wa[0] = Elem<__bf16>::pack2(oa[db][4 * g4] * inv_a, oa[db][4 * g4 + 1] * inv_a);
                wa[1] = Elem<__bf16>::pack2(oa[db][4 * g4 + 2] * inv_a, oa[db][4 * g4 + 3] * inv_a);
                wb[0] = Elem<__bf16>::pack2(ob[db][4 * g4] * inv_b, ob[db][4 * g4 + 1] * inv_b);
                wb[1] = Elem<__bf16>::pack2(ob[db][4 * g4 + 2] * inv_b, ob[db][4 * g4 + 3] * inv_b);
                const int col = (db * 32 + 8 * g4 + 4 * hh_e) * 2;
                *(u32x2 *)(obuf + r_e * O_ROW_BYTES + col) = wa;
                *(u32x2 *)(obuf + (32 + r_e) * O_ROW_BYTES + col) = wb;
            }
    }
    __syncthreads();
    if (wave_active) {
        const char *obuf = smem + wave * (64 * O_ROW_BYTES);
        constexpr int CH_PER_ROW = D / 8, NCH = (64 * CH_PER_ROW) / 64;
        u32x4 val[NCH];
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) {
            const int c = lane_e + c2 * 64;
            val[c2] = *(const u32x4 *)(obuf + (c / CH_PER_ROW) * O_ROW_BYTES + (c % CH_PER_ROW) * 16);
        }
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) {
            const int c = lane_e + c2 * 64;
            const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
            if (wrow + row < sq) *(u32x4 *)(op + (int64_t)(wrow + row) * p.o_row_stride + ch * 8) = val[c2];
        }
    }
}

constexpr int smem_bytes_fp8() { return 6 * 64 * 128 + 4 * 10 * 1024; }  // K/V rings (48 KiB) + the state hand-off area (40 KiB); the O staging (68 KiB) reuses them

}  // namespace fa
