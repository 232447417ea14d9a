// fa_fwd_kernel_w64.h — gfx950 FlashAttention forward, "one wave per SIMD" mainloop.
//
// Same algorithm and LDS images as fa_fwd_kernel.h (transposed scores S^T = K.Q^T with the query row on
// the lane, O^T = V^T.P^T fed straight from the score accumulators, V^T through ds_read_b64_tr_b16), but a
// different decomposition, sized for the 512-entry register file a wave owns when it is alone on its SIMD:
//
//   * workgroup = 4 waves (one per SIMD), BLOCK_M = 256; each wave owns 64 query rows = two 32-row
//     q-blocks A and B.  Every K fragment (ds_read_b128) and every V^T fragment (2 x ds_read_b64_tr_b16)
//     feeds TWO MFMAs (one per q-block): half the LDS read traffic per FLOP of the 32-rows-per-wave shape.
//   * the 64-key K/V tile is consumed as two 32-key halves.  With nobody else on the SIMD the wave must
//     overlap its own MFMA and VALU work, so each half-step h is two phases, each one basic block with an
//     independent MFMA stream and VALU stream of equal weight (16 MFMAs vs one 32x32 softmax):
//         phase 1   MFMA: S_A(h+1), S_B(h+1) = K(h+1).Q^T         VALU: softmax of S_B(h)  -> P_B(h)
//         phase 2   MFMA: O_A += V(h)^T.P_A(h), O_B += V(h)^T.P_B(h)   VALU: softmax of S_A(h+1) -> P_A(h+1)
//     q-block A's softmax runs one half-step ahead of its PV product, so its O rescale is deferred to the
//     next phase 1 (after the PV MFMAs that still use the old scale have been issued).
//     Role of the intra-warpgroup overlap of hopper/mainloop_fwd_sm90_tma_gmma_ws.hpp:1170-1207.
//   * K tiles are staged SHIFTED by 32 keys (K tile m = keys [64m-32, 64m+32)) so that the two score halves
//     computed during V tile n both come from K tile n+1.  K and V each live in a ring of 3 LDS buffers
//     (tile t in buffer (t - n_min) % 3); the LDS-DMA for K(n+3)/V(n+2) is issued at the top of tile n and only
//     has to have landed at the END of tile n+1 (counted vmcnt), i.e. it has two tiles of compute to hide under;
//     one barrier per 64 keys.
#pragma once

#include "fa_fwd_kernel.h"
#ifdef FA_LOOP_GEN_HEADER  /* developer-only: a timing-ablation variant of the generated loop */
#include FA_LOOP_GEN_HEADER
#else
#include "fa_fwd_loop_gen.h"
#endif

#include <type_traits>

namespace fa {

// log2-domain growth of the running max below which O/l are NOT rescaled (the stale max is kept and P may
// reach 2^THR).  0 = rescale whenever any row's max moves: bit-identical to always rescaling.
#ifndef FA_RESCALE_THR
#define FA_RESCALE_THR 8  /* must be > 0: the fast path keeps a stale max and guards it with 2^THR */
#endif


// ---- softmax inner step for two scores in ONE asm statement --------------------------------------------------
//   x = exp2(s0*c - mc), y = exp2(s1*c - mc);  ps0 += x;  ps1 += y;  returns pack(x, y) in the input dtype (RNE)
// Written in asm as a unit because (a) gfx950 needs one wait state between a transcendental (v_exp) and a
// non-transcendental VALU reader of its result, which hipcc pads only for its own instructions -- the order
// exp,exp,add,add,cvt keeps one instruction between every producer/consumer pair; (b) it keeps hipcc from packing
// the adds into v_pk_add_f32 (slow beside MFMAs) and (c) one statement = one asm boundary pad.
// front/back: the same step split in two statements so one MFMA can be issued between them (the wave issues in
// order: [MFMA][fma fma exp exp][MFMA][add add cvt] keeps the matrix pipe fed every 32 cycles, while
// [MFMA MFMA][7 VALU] stalls the second MFMA on the pipe and then leaves the pipe idle behind the VALU block).
template <typename T> struct Exp2Pair;
template <> struct Exp2Pair<__bf16> {
    static __device__ __forceinline__ void front(float s0, float s1, float c, float mc, float &t0, float &t1) {
        asm("v_fma_f32 %0, %2, %4, -%5\n\t"
            "v_fma_f32 %1, %3, %4, -%5\n\t"
            "v_exp_f32 %0, %0\n\t"
            "v_exp_f32 %1, %1"
            : "=&v"(t0), "=&v"(t1)
            : "v"(s0), "v"(s1), "s"(c), "v"(mc));
    }
    static __device__ __forceinline__ uint32_t back(float t0, float t1, float &ps0, float &ps1) {
        uint32_t pk;
        asm("v_add_f32 %1, %1, %3\n\t"
            "v_add_f32 %2, %2, %4\n\t"
            "v_cvt_pk_bf16_f32 %0, %3, %4"
            : "=&v"(pk), "+v"(ps0), "+v"(ps1)
            : "v"(t0), "v"(t1));
        return pk;
    }
    static __device__ __forceinline__ uint32_t run(float s0, float s1, float c, float mc, float &ps0, float &ps1) {
        uint32_t pk;
        float t0, t1;
        asm("v_fma_f32 %3, %5, %7, -%8\n\t"
            "v_fma_f32 %4, %6, %7, -%8\n\t"
            "v_exp_f32 %3, %3\n\t"
            "v_exp_f32 %4, %4\n\t"
            "v_add_f32 %1, %1, %3\n\t"
            "v_add_f32 %2, %2, %4\n\t"
            "v_cvt_pk_bf16_f32 %0, %3, %4"
            : "=&v"(pk), "+v"(ps0), "+v"(ps1), "=&v"(t0), "=&v"(t1)
            : "v"(s0), "v"(s1), "s"(c), "v"(mc));
        return pk;
    }
};
template <> struct Exp2Pair<_Float16> {
    static __device__ __forceinline__ void front(float s0, float s1, float c, float mc, float &t0, float &t1) {
        asm("v_fma_f32 %0, %2, %4, -%5\n\t"
            "v_fma_f32 %1, %3, %4, -%5\n\t"
            "v_exp_f32 %0, %0\n\t"
            "v_exp_f32 %1, %1"
            : "=&v"(t0), "=&v"(t1)
            : "v"(s0), "v"(s1), "s"(c), "v"(mc));
    }
    static __device__ __forceinline__ uint32_t back(float t0, float t1, float &ps0, float &ps1) {
        uint32_t pk;
        float h0, h1;
        asm("v_add_f32 %1, %1, %5\n\t"
            "v_add_f32 %2, %2, %6\n\t"
            "v_cvt_f16_f32 %3, %5\n\t"
            "v_cvt_f16_f32 %4, %6\n\t"
            "v_pack_b32_f16 %0, %3, %4"
            : "=&v"(pk), "+v"(ps0), "+v"(ps1), "=&v"(h0), "=&v"(h1)
            : "v"(t0), "v"(t1));
        return pk;
    }
    static __device__ __forceinline__ uint32_t run(float s0, float s1, float c, float mc, float &ps0, float &ps1) {
        uint32_t pk;
        float t0, t1;
        asm("v_fma_f32 %3, %5, %7, -%8\n\t"
            "v_fma_f32 %4, %6, %7, -%8\n\t"
            "v_exp_f32 %3, %3\n\t"
            "v_exp_f32 %4, %4\n\t"
            "v_add_f32 %1, %1, %3\n\t"
            "v_add_f32 %2, %2, %4\n\t"
            "v_cvt_f16_f32 %3, %3\n\t"
            "v_cvt_f16_f32 %4, %4\n\t"
            "v_pack_b32_f16 %0, %3, %4"
            : "=&v"(pk), "+v"(ps0), "+v"(ps1), "=&v"(t0), "=&v"(t1)
            : "v"(s0), "v"(s1), "s"(c), "v"(mc));
        return pk;
    }
};

// MFMA wrappers with explicit register classes: S accumulators in arch VGPRs (the VALU reads them), Q (B operand
// of the score product) and the O accumulators in AGPRs.  Q is an in/out ("+a") operand of every score MFMA
// although it is only read: each statement then hands the NEXT one an AGPR-defined value, which keeps hipcc's
// register allocator from parking Q in VGPRs and copying it into AGPRs in front of every MFMA (unpadded copies).
// hipcc does not pad hazards of asm MFMAs: callers keep >= 1 independent MFMA (or a drain) between an asm
// MFMA and any VALU reader of its result.
// Row max of a 32x32 score block over this lane's 16 registers, seeded with m: two v_max3 chains in ONE asm
// statement (hipcc pads every asm statement boundary with s_nop; one statement = one pad).
__device__ __forceinline__ void rowmax16(const f32x16 &s, float m, float &mxa, float &mxb) {
    asm("v_max3_f32 %0, %2, %3, %18\n\t"
        "v_max3_f32 %1, %4, %5, %18\n\t"
        "v_max3_f32 %0, %0, %6, %7\n\t"
        "v_max3_f32 %1, %1, %8, %9\n\t"
        "v_max3_f32 %0, %0, %10, %11\n\t"
        "v_max3_f32 %1, %1, %12, %13\n\t"
        "v_max3_f32 %0, %0, %14, %15\n\t"
        "v_max3_f32 %1, %1, %16, %17"
        : "=&v"(mxa), "=&v"(mxb)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]),
          "v"(s[9]), "v"(s[10]), "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]), "v"(m));
}

template <typename T> struct Mfma;
template <> struct Mfma<__bf16> {
    // PAD variants (generic path): hipcc may put its own VALU / v_accvgpr writes of an operand right in front of
    // an asm MFMA and pads nothing for asm, so the padded forms carry the 2 wait states themselves.
    static __device__ __forceinline__ void s_first_pad(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %1, 0" : "=&v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void s_acc_pad(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %1, %0" : "+v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void o_acc_pad(f32x16 &o, u32x4 v, u32x4 pf) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pf));
    }
    static __device__ __forceinline__ void o_zero(f32x16 &o, u32x4 z) {  // O = 0*0 + 0: no VALU/accvgpr write involved
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(o) : "v"(z));
    }
    static __device__ __forceinline__ void s_first(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %2, %1, 0" : "=&v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void s_acc(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %2, %1, %0" : "+v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void o_acc(f32x16 &o, u32x4 v, u32x4 pf) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pf));
    }
    // first PV MFMA of a phase: pads the VALU->MFMA operand hazard in front, and carries the fresh score tiles
    // as dummy operands so that no VALU reader of them is scheduled before this (independent) MFMA
    static __device__ __forceinline__ void o_acc_fence(f32x16 &o, u32x4 v, u32x4 pf, f32x16 &s0, f32x16 &s1) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n\ts_nop 3"
            : "+a"(o), "+v"(s0), "+v"(s1) : "v"(v), "v"(pf));
    }
};
template <> struct Mfma<_Float16> {
    static __device__ __forceinline__ void s_first_pad(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, 0" : "=&v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void s_acc_pad(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void o_acc_pad(f32x16 &o, u32x4 v, u32x4 pf) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pf));
    }
    static __device__ __forceinline__ void o_zero(f32x16 &o, u32x4 z) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(o) : "v"(z));
    }
    static __device__ __forceinline__ void s_first(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("v_mfma_f32_32x32x16_f16 %0, %2, %1, 0" : "=&v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void s_acc(f32x16 &d, u32x4 k, u32x4 &q) {
        asm("v_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(d), "+a"(q) : "v"(k));
    }
    static __device__ __forceinline__ void o_acc(f32x16 &o, u32x4 v, u32x4 pf) {
        asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pf));
    }
    static __device__ __forceinline__ void o_acc_fence(f32x16 &o, u32x4 v, u32x4 pf, f32x16 &s0, f32x16 &s1) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\ts_nop 3"
            : "+a"(o), "+v"(s0), "+v"(s1) : "v"(v), "v"(pf));
    }
};

// ---- fused slice of the fast path: ONE asm statement = two MFMAs (q-blocks A and B, same LDS fragment) interleaved
// with the softmax step of two scores:
//     MFMA A ; t0 = exp2(s0*c - mc) ; t1 = exp2(s1*c - mc) ; MFMA B ; ps0 += t0 ; ps1 += t1 ; pk = pack(t0, t1)
// The wave is alone on its SIMD and issues in order, so this text order IS the schedule: ~24 cycles of VALU issue
// behind each MFMA keep the matrix pipe fed every 32 cycles.  One statement per slice also means a single hipcc
// asm-boundary pad per slice, and the exp -> add distance (one MFMA) satisfies gfx950's transcendental forwarding rule.
// score_slice<FIRST>: score tiles (D in VGPRs; B = Q pinned in AGPRs as in/out operands); FIRST: D = A.B + 0
// out_slice<FENCE>  : O accumulate (D in AGPRs, A = V^T fragment, B = P^T fragments in VGPRs); FENCE = first slice of a
//                     phase: s_nop 1 in front (VALU-written P -> MFMA operand) and s_nop 3 behind the first MFMA (the
//                     fma that follows reads score registers written by asm MFMAs of the previous phase)
#define FA_FUSED_SLICE(MN, CVT_S, CVT_O)                                                                             \
    template <bool FIRST>                                                                                            \
    static __device__ __forceinline__ uint32_t score_slice(f32x16 &da, u32x4 &qa, f32x16 &db, u32x4 &qb, u32x4 kf,    \
                                                           float s0, float s1, float c, float mc, float &ps0,        \
                                                           float &ps1) {                                             \
        uint32_t pk;                                                                                                 \
        float t0, t1;                                                                                                \
        if constexpr (FIRST)                                                                                         \
            asm(MN " %0, %9, %1, 0\n\tv_fma_f32 %7, %10, %12, -%13\n\tv_fma_f32 %8, %11, %12, -%13\n\t"              \
                   "v_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\t" MN " %2, %9, %3, 0\n\t"                                \
                   "v_add_f32 %5, %5, %7\n\tv_add_f32 %6, %6, %8\n\t" CVT_S                                          \
                : "=&v"(da), "+a"(qa), "=&v"(db), "+a"(qb), "=&v"(pk), "+v"(ps0), "+v"(ps1), "=&v"(t0), "=&v"(t1)     \
                : "v"(kf), "v"(s0), "v"(s1), "s"(c), "v"(mc));                                                       \
        else                                                                                                         \
            asm(MN " %0, %9, %1, %0\n\tv_fma_f32 %7, %10, %12, -%13\n\tv_fma_f32 %8, %11, %12, -%13\n\t"             \
                   "v_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\t" MN " %2, %9, %3, %2\n\t"                               \
                   "v_add_f32 %5, %5, %7\n\tv_add_f32 %6, %6, %8\n\t" CVT_S                                          \
                : "+v"(da), "+a"(qa), "+v"(db), "+a"(qb), "=&v"(pk), "+v"(ps0), "+v"(ps1), "=&v"(t0), "=&v"(t1)       \
                : "v"(kf), "v"(s0), "v"(s1), "s"(c), "v"(mc));                                                       \
        return pk;                                                                                                   \
    }                                                                                                                \
    template <bool FENCE>                                                                                            \
    static __device__ __forceinline__ uint32_t out_slice(f32x16 &oa, u32x4 pa, f32x16 &ob, u32x4 pb, u32x4 vf,        \
                                                         float s0, float s1, float c, float mc, float &ps0,          \
                                                         float &ps1) {                                               \
        uint32_t pk;                                                                                                 \
        float t0, t1;                                                                                                \
        if constexpr (FENCE)                                                                                         \
            asm("s_nop 1\n\t" MN " %0, %7, %8, %0\n\ts_nop 3\n\tv_fma_f32 %5, %10, %12, -%13\n\t"                    \
                "v_fma_f32 %6, %11, %12, -%13\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\t" MN " %1, %7, %9, %1\n\t"  \
                "v_add_f32 %3, %3, %5\n\tv_add_f32 %4, %4, %6\n\t" CVT_O                                             \
                : "+a"(oa), "+a"(ob), "=&v"(pk), "+v"(ps0), "+v"(ps1), "=&v"(t0), "=&v"(t1)                           \
                : "v"(vf), "v"(pa), "v"(pb), "v"(s0), "v"(s1), "s"(c), "v"(mc));                                     \
        else                                                                                                         \
            asm(MN " %0, %7, %8, %0\n\tv_fma_f32 %5, %10, %12, -%13\n\t"                                             \
                "v_fma_f32 %6, %11, %12, -%13\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\t" MN " %1, %7, %9, %1\n\t"  \
                "v_add_f32 %3, %3, %5\n\tv_add_f32 %4, %4, %6\n\t" CVT_O                                             \
                : "+a"(oa), "+a"(ob), "=&v"(pk), "+v"(ps0), "+v"(ps1), "=&v"(t0), "=&v"(t1)                           \
                : "v"(vf), "v"(pa), "v"(pb), "v"(s0), "v"(s1), "s"(c), "v"(mc));                                     \
        return pk;                                                                                                   \
    }

template <typename T> struct Fused;
template <> struct Fused<__bf16> {
    FA_FUSED_SLICE("v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32 %4, %7, %8", "v_cvt_pk_bf16_f32 %2, %5, %6")
};
template <> struct Fused<_Float16> {
    // fp16 pack: two round-to-nearest-even converts, the second one into the high half of the same register (SDWA)
    FA_FUSED_SLICE("v_mfma_f32_32x32x16_f16",
                   "v_cvt_f16_f32 %4, %7\n\tv_cvt_f16_f32_sdwa %4, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD",
                   "v_cvt_f16_f32 %2, %5\n\tv_cvt_f16_f32_sdwa %2, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD")
};
#undef FA_FUSED_SLICE

// LDS-DMA of N consecutive 1-KiB pieces: piece i = 64 lanes x 16 B from (base + off[i]) to LDS byte address
// lds + 1024 i.  Inline asm on purpose: hipcc cannot prove that the DMA destination does not alias later ds_reads
// and drains vmcnt(0) right behind a __builtin_amdgcn_global_load_lds, serialising every tile load; asm loads
// are invisible to its waitcnt pass, and the caller waits (s_waitcnt vmcnt) in front of the barrier that
// publishes the tile.  M0 (the LDS base of an LDS-DMA) is written and restored inside the statement.
template <int N>
__device__ __forceinline__ void lds_dma(uint32_t lds, const void *base, const uint32_t (&off)[N]) {
    uint32_t keep;
    static_assert(N == 2 || N == 4 || N == 8, "pieces per wave and tile");
    if constexpr (N == 8) {  // (head-dim tile 256: 512-byte rows, the backward's Q / dO / K / V tiles)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
            "s_add_u32 m0, %2, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
            "s_add_u32 m0, %2, 2048\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
            "s_add_u32 m0, %2, 3072\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
            "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %1\n\t"
            "s_add_u32 m0, %3, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %1\n\t"
            "s_add_u32 m0, %3, 2048\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %1\n\t"
            "s_add_u32 m0, %3, 3072\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %11, %1\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(base), "s"(lds), "s"(lds + 4096), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]),
              "v"(off[6]), "v"(off[7])
            : "memory", "scc");
    } else if constexpr (N == 4) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
            "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %1\n\t"
            "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %1\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(base), "s"(lds), "s"(lds + 1024), "s"(lds + 2048), "s"(lds + 3072), "v"(off[0]), "v"(off[1]),
              "v"(off[2]), "v"(off[3])
            : "memory");
    } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
            "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(base), "s"(lds), "s"(lds + 1024), "v"(off[0]), "v"(off[1])
            : "memory");
    }
}
// one piece (fast path: issued inside an MFMA slice so its ~50-100 issue cycles hide under the matrix pipe)
__device__ __forceinline__ void lds_dma1(uint32_t lds, const void *base, uint32_t off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(lds), "v"(off) : "memory");
}
// one 1-KiB piece through a raw buffer descriptor: lanes whose offset is past num_records land as zeros
__device__ __forceinline__ void lds_dma_buf1(uint32_t lds, u32x4 desc, uint32_t off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %1, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(desc), "s"(lds), "v"(off) : "memory");
}
// 64 lanes x 4 B (touches one cache line per lane): used to pull lines into the L2 ahead of time, the LDS copy is a dump
__device__ __forceinline__ void lds_dma_touch(uint32_t lds, const void *base, uint32_t off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %3, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "s"(lds), "v"(off) : "memory");
}
// End of a tile: all but this wave's N youngest LDS-DMA pieces have landed (the N issued at the top of THIS tile
// may stay in flight: they fill buffers nobody reads before the next-but-one barrier), then the workgroup barrier.
template <int N>
__device__ __forceinline__ void tile_barrier() {
    static_assert(N == 0 || N == 2 || N == 4 || N == 8, "pieces in flight");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void tile_barrier_n(int n_in_flight, int ld) {  // generic path: n in {0, ld, 2 ld}
    if (n_in_flight >= 2 * ld) { if (ld == 4) tile_barrier<8>(); else tile_barrier<4>(); }
    else if (n_in_flight >= ld) { if (ld == 4) tile_barrier<4>(); else tile_barrier<2>(); }
    else tile_barrier<0>();
}

__device__ __forceinline__ void drain_scores(f32x16 &s0, f32x16 &s1) {
    // asm MFMA results -> VALU readers (generic path only)
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s0), "+v"(s1));
}

// Developer-only timing ablations of the fast path (results are WRONG when non-zero; never shipped):
// 1 = no phase-1 VALU, 2 = no phase-2 VALU, 4 = no K/V loads, 8 = no barriers, 16 = no LDS fragment reads.
#ifndef FA_ABLATE
#define FA_ABLATE 0
#endif


// Developer-only phase timestamps (never shipped: -DFA_TIMING builds, tools/wg_phases.py): every wave drops the 100 MHz
// wall clock into 64 spare LDS bytes (per wave) behind the K/V rings (LDS traffic only: the loop's manual vmcnt accounting is not
// disturbed) and copies them to fa_timing_buf[workgroup][wave][8] at the very end.
#ifdef FA_TIMING
__device__ unsigned long long fa_timing_buf[32 * 4096];
// (every lane stores -- same address, same value per wave: a lane-0 branch would be jump-threaded across the phases by
//  hipcc and drag the wave-uniform LDS-DMA bases into VGPRs)
#ifndef FA_TMASK
#define FA_TMASK 255
#endif
#define FA_T(i) do { if constexpr ((FA_TMASK >> (i)) & 1) ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[(threadIdx.x >> 6) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FA_T(i)
#endif

// Developer-only cycle stamps of the fast loop (never shipped: -DFA_CYCLES builds, tools/loop_cycles.py): each wave appends
// s_memtime (shader clock) at every fast half-step entry and between its two phases -- up to 60 stamps -- plus the 100 MHz
// wall clock at the first and last stamp (slots 62, 63), into spare LDS; copied to fa_cycle_buf[workgroup][wave][64] at
// the end.  The stamp drains lgkmcnt (s_memtime returns through it), i.e. it perturbs the LDS prefetch by ~100 cycles.
#ifdef FA_CYCLES
#ifndef FA_CYCLES_WG0
#define FA_CYCLES_WG0 0   // first of the 256 workgroups whose stamps are kept (0: the launch's first round, cold caches)
#endif
#ifndef FA_CYCLES_STRIDE
#define FA_CYCLES_STRIDE 1  // keep every n-th workgroup id of the window (2: two launch rounds of the even XCDs -> CU hand-over gaps)
#endif
__device__ unsigned long long fa_cycle_buf[256 * 4 * 64];
#define FA_C() do { if (!PERSIST && cyc_n < 60) { unsigned long long *cb_ = (unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024) + (threadIdx.x >> 6) * 64; \
        if (cyc_n == 0) cb_[62] = __builtin_amdgcn_s_memrealtime(); cb_[cyc_n++] = __builtin_amdgcn_s_memtime(); cb_[63] = __builtin_amdgcn_s_memrealtime(); cb_[61] = cyc_n; } } while (0)
// (PERSIST keeps the next item's Q in that image during the sweep: its stamps go to fa_cycle_buf directly, FA_PSTAMP below)
#define FA_STAMP(k) do { if constexpr (!PERSIST) ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[(threadIdx.x >> 6) * 64 + (k)] = __builtin_amdgcn_s_memrealtime(); \
                         else FA_PSTAMP(k); } while (0)
// (the stamp area lies inside the Q staging image: stamps taken before the Q fragments are in registers wait in scalars)
#define FA_STAMP_VAR(k) const unsigned long long fa_st_##k = __builtin_amdgcn_s_memrealtime()
#define FA_STAMP_FLUSH(k) do { if constexpr (!PERSIST) ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[(threadIdx.x >> 6) * 64 + (k)] = fa_st_##k; } while (0)
// persistent kernel: the workgroup's item number `fa_item` (0, 1, ...) selects the row: stamps of items 1 and 2 are kept
#define FA_PSTAMP(k) do { if (fa_item >= 1 && fa_item <= 2 && blockIdx.x < 128) \
    fa_cycle_buf[((blockIdx.x * 2 + fa_item - 1) * 4 + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FA_STAMP_ENTRY() const unsigned long long fa_st_entry = __builtin_amdgcn_s_memrealtime()
#define FA_STAMP_ENTRY_FLUSH(k) do { if constexpr (!PERSIST) ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[(threadIdx.x >> 6) * 64 + (k)] = fa_st_entry; } while (0)
// XCC_ID (hwreg 20) << 32 | HW_ID (hwreg 4: cu_id [11:8], sh_id [12], se_id [15:13]): which CU ran this workgroup
#define FA_STAMP_HWID(k) do { if constexpr (D == 128 && !PERSIST) ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[(threadIdx.x >> 6) * 64 + (k)] = \
    ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11)); } while (0)
#else
#define FA_C()
#define FA_STAMP(k)
#define FA_STAMP_VAR(k)
#define FA_STAMP_FLUSH(k)
#define FA_PSTAMP(k)
#define FA_STAMP_HWID(k)
#define FA_STAMP_ENTRY()
#define FA_STAMP_ENTRY_FLUSH(k)
#endif

// DEFF (head-dim tile 128 only): 96 when the head dim is <= 96 -- the generated loop then skips the k-steps and O blocks
// of the zero padding (6 + 6 instead of 8 + 8 MFMA pairs per half-step; the LDS images keep their 256-byte rows).
//
// PERSIST (head-dim tile 128, dense batches, plain features: the host decides, fa_fwd_api.hip persist_ok()): one workgroup per
// CU walks a chain of work items (role of hopper/tile_scheduler.hpp:140-214 + flash_fwd_kernel_sm90.h:308-359).  The K / V
// look-ahead stream of an item's last three tiles fetches the NEXT item's first tiles instead of rows past the end, its Q
// lands in the Q staging image (dead once the fragments are in registers) during the sweep, and the epilogue stages O in
// that image -- so an item switch costs the epilogue and the first scores, not the loads in front of them.
template <typename T, int D, bool SOFTCAP, int DEFF = D, bool PERSIST = false>
__global__ __launch_bounds__(256, 1) void fwd_kernel_w64(const KParams p) {
    static_assert(!PERSIST || (D == 128 && DEFF == 128 && !SOFTCAP), "the persistent form exists for the plain head-dim-128 kernel");
    constexpr int NT = 256;
    constexpr int BLOCK_M = 256;
    constexpr int KSTEPS = D / 16;
    constexpr int DBLOCKS = D / 32;
    constexpr int CH_PER_ROW = D / 8;
    constexpr int ROWB = D * 2;                  // bytes per LDS row
    constexpr int TILE_BYTES = BLOCK_N * ROWB;
    constexpr int LD_PER_THREAD = BLOCK_N * CH_PER_ROW / NT;
    constexpr int O_ROW_BYTES = D * 2 + 16;
    constexpr float THR = (float)FA_RESCALE_THR;
    static_assert(D == 64 || D == 128, "w64 shape is built for head-dim tiles 64 and 128");

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [K0 | K1 | K2 | V0 | V1 | V2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int hh = lane >> 5;

    FA_STAMP_ENTRY();  // kernel entry (developer builds)
    int m_block, head, batch;
    int split;
    if (!decode_tile(p, m_block, head, batch, split)) return;  // whole workgroup (padding)
    int kv_head = head / p.h_ratio;

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
    int row_lo = m_block * BLOCK_M;
    if (row_lo >= sq) return;
    FA_T(0);

    o_base += split * p.o_split_stride;  // split-KV: partial results of split s (0 when off)
    lse_base += split * p.lse_split_stride;
    if (p.leftpad_k) {  // left-padded keys: skip the padding rows, the valid length shrinks by as much
        const int lp = p.leftpad_k[batch];
        sk = max(sk - lp, 0);
        k_base += (int64_t)lp * p.k_row_stride;
        v_base += (int64_t)lp * p.v_row_stride;
    }
#ifdef FA_CYCLES
    // (the entry stamp counts as a memory write in front of the cu_seqlens / leftpad lookups, which hipcc then issues as vector
    //  loads: "s" asm operands derived from them no longer assemble -- make them scalars again; developer builds only)
    auto uni64 = [](int64_t x) {
        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)x));
    };
    sq = __builtin_amdgcn_readfirstlane(sq); sk = __builtin_amdgcn_readfirstlane(sk);
    q_base = uni64(q_base); k_base = uni64(k_base); v_base = uni64(v_base); o_base = uni64(o_base); lse_base = uni64(lse_base);
#endif
    const T *qp = (const T *)p.q + q_base + (int64_t)head * p.q_head_stride;
    const T *kp = (const T *)p.k + k_base + (int64_t)kv_head * p.k_head_stride;
    const T *vp = (const T *)p.v + v_base + (int64_t)kv_head * p.v_head_stride;
    T *op = (T *)p.o + o_base + (int64_t)head * p.o_head_stride;
    // PERSIST: the chain of this workgroup.  Round t of the CU with index k inside its XCD (blockIdx.x >> 3; the launch is one
    // workgroup per CU, ids dealt round-robin over the XCDs) takes slot cpx t + k of the XCD's slot list (tile_of_wg), odd
    // rounds in reverse (cpx - 1 - k): under a causal mask the slots of a unit get lighter in order, and the zig-zag pairs a
    // heavy item with a light one.  Placement is a speed matter only: every item is taken by exactly one workgroup.
    int chain_round = 0;
    bool has_next = false;
    int m_block_n = 0, head_n = 0, batch_n = 0;
    int n_min_nx = 0;  // first key tile of the next item (> 0 under a left window; dense batches: the same shift for every item)
    const T *kp_n = kp, *vp_n = vp;
    // The chain is decoded once, lane t = round t (the host keeps chains at <= 64 rounds and every field inside its bits):
    // m_block << 20 | head << 10 | batch, or ~0 for a round without an item for this CU.  In the item loop a round is one
    // v_readlane -- the integer divisions of the decode (VALU reciprocals hipcc would hoist and keep alive across every
    // asm block, i.e. spill) happen here, for all rounds at once.
    uint32_t chain_tbl = ~0u;
    if constexpr (PERSIST) {
        const int cpx = (int)gridDim.x >> 3, k = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
        const int t = lane;
        const int slot = cpx * t + ((t & 1) ? cpx - 1 - k : k);
        const int wg = slot * 8 + xcd;
        if (cpx * t * 8 < p.grid && wg < p.grid) {
            const int tile = tile_of_wg(p, wg);
            if (tile < p.num_tiles) {
                const int per_kvh = p.h_ratio * p.num_m_blocks;
                const int bk = tile / per_kvh, r2 = tile % per_kvh;
                chain_tbl = (uint32_t)(p.num_m_blocks - 1 - r2 / p.h_ratio) << 20 | (uint32_t)((bk % p.h_k) * p.h_ratio + r2 % p.h_ratio) << 10 |
                            (uint32_t)(bk / p.h_k);
            }
        }
    }
    auto chain_next = [&]() {  // (m_block_n, head_n, batch_n) of the next item of the chain, or has_next = false
        has_next = false;
        const int rounds = (p.grid + (int)gridDim.x - 1) / (int)gridDim.x;  // (<= 64)
        for (int t = chain_round + 1; t < rounds; ++t) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)chain_tbl, t);
            if (e == ~0u) continue;
            batch_n = (int)(e & 1023u);
            head_n = (int)((e >> 10) & 1023u);
            m_block_n = (int)(e >> 20);
            chain_round = t;
            has_next = true;
            n_min_nx = 0;
            if (p.window_left >= 0) {  // item_geometry()'s key_lo of the next item (no split-KV, no varlen here)
                n_min_nx = max(0, m_block_n * 256 + (p.seqlen_k - p.seqlen_q) - p.window_left) / BLOCK_N;  // (dense: persist_ok)
            }
            break;
        }
    };
    // The few kernel arguments the main loop needs, detached from the kernarg SGPR tuples (hipcc loads the by-value
    // struct as s_load_dwordx8/x16 tuples and, once those spill, reloads a whole tuple through v_readlane -- VALU
    // instructions -- every tile just to reach one field).  The empty asm makes each one a fresh scalar value.
    const Scales sc = load_scales(p, batch, kv_head);
    const float alibi = load_alibi(p, sc, batch, head);
    int64_t k_rs64 = p.k_row_stride, v_rs64 = p.v_row_stride;
    // (scale * descale is a VALU product: bring it back to an SGPR by value, not by int conversion)
    float csc_arg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.scale_log2)));
    asm volatile("" : "+s"(k_rs64), "+s"(v_rs64), "+s"(csc_arg));
    // the two factors of the epilogue, as scalars as well: in VGPRs they are spilled to scratch across the main loop and
    // each reload is an awaited memory round trip in the epilogue (~1 us apiece); an SGPR spills to a VGPR lane
    float scale_e = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.scale)));
    float vdesc_e = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc.v_descale)));
    asm volatile("" : "+s"(scale_e), "+s"(vdesc_e));

    const int shift = sk - sq;
    int n_min, n_max, wrow, jend;
    // key-tile range, first row and half-step count of this wave for the item in (row_lo, ...); PERSIST: again per item
    auto item_geometry = [&]() {
        const int row_hi = min(sq, row_lo + BLOCK_M);
        int key_hi = sk, key_lo = 0;
        if (p.window_right >= 0) key_hi = min(sk, row_hi + shift + p.window_right);
        if (p.window_left >= 0) key_lo = max(0, row_lo + shift - p.window_left);
        int n_min_ = key_lo / BLOCK_N;
        int n_max_ = key_hi > 0 ? (key_hi + BLOCK_N - 1) / BLOCK_N : 0;
        split_range(p, split, n_min_, n_max_);
        n_min = __builtin_amdgcn_readfirstlane(n_min_);
        n_max = __builtin_amdgcn_readfirstlane(n_max_);
        wrow = __builtin_amdgcn_readfirstlane(row_lo + wave * 64);  // first row of this wave; q-block A = wrow.., B = wrow+32..
        // half-steps [0, jend) this wave computes (half-step j = keys [64 n_min + 32 j, +32)); later ones are
        // fully masked for all of its 64 rows (causal / right window).  Waves past the end of q: none.
        jend = 2 * (n_max - n_min);
        if (p.window_right >= 0) {
            const int last_key = min(sk - 1, wrow + 63 + shift + p.window_right);
            jend = min(jend, last_key >= n_min * BLOCK_N ? (last_key - n_min * BLOCK_N) / 32 + 1 : 0);
        }
        if (wrow >= sq || n_min >= n_max) jend = 0;
        jend = __builtin_amdgcn_readfirstlane(jend);
    };
    item_geometry();

    FA_STAMP_VAR(46);
    // ---- Q fragments (B operand of S^T = K.Q^T): fetched in the prologue below, behind the first K tile ----------
    u32x4 qa[KSTEPS], qb[KSTEPS];

    f32x16 oa[DBLOCKS], ob[DBLOCKS];
    {
        const u32x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int db = 0; db < DBLOCKS; ++db) {
            Mfma<T>::o_zero(oa[db], z4);
            Mfma<T>::o_zero(ob[db], z4);
        }
        // (hipcc may spill an accumulator to scratch right here: let the matrix pipe finish first)
        if constexpr (DBLOCKS == 4)
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(oa[2]), "+a"(oa[3]),
                         "+a"(ob[0]), "+a"(ob[1]), "+a"(ob[2]), "+a"(ob[3]));
        else
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(ob[0]), "+a"(ob[1]));
    }
    float m_a = -INFINITY, m_b = -INFINITY;  // running max (unscaled scores)
    float l_a = 0.f, l_b = 0.f;              // running sums, partial per lane half

    // ---- K/V staging by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write ---------------
    // One wave-instruction writes 1 KiB = 4 rows x 16 chunks of LDS, lane l -> byte 16 l.  The XOR swizzle of the
    // LDS image is applied on the SOURCE side: lane l of piece c fetches the chunk that belongs at that slot.
    // Rows past the end of the sequence are clamped to the last valid row and head-dim chunks past d to chunk 0
    // (duplicates are masked / meet zero Q chunks: see fa_fwd_kernel.h).  Loads are branch-free.
    int ld_row[LD_PER_THREAD], ld_col[LD_PER_THREAD];
    uint32_t koff[LD_PER_THREAD], voff[LD_PER_THREAD];  // byte offsets of this lane's chunks inside an in-range tile
    const int k_rs = (int)k_rs64, v_rs = (int)v_rs64;
    int kbase, vbase;  // lane parts of the LDS fragment addresses (below)
    // PERSIST: every per-lane table is rebuilt at the top of each work item from an opaque copy of the lane id -- kept alive
    // across the item loop (through every asm block and the epilogue) they cost spills; rebuilt they cost ~40 VALU per item
    auto lane_tables = [&](int lane_) {
#pragma unroll
    for (int i = 0; i < LD_PER_THREAD; ++i) {
        const int slot = wave * (LD_PER_THREAD * 64) + i * 64 + lane_;  // 16-byte slot index inside the tile image
        const int row = slot / CH_PER_ROW;
        // inverse of lds_off<D>: the chunk stored at slot position (slot % CH_PER_ROW) of this row
        int ch;
        if constexpr (D == 64) ch = (slot % CH_PER_ROW) ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
        else ch = (slot % CH_PER_ROW) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        ld_row[i] = row;
        ld_col[i] = (ch * 8 < p.d) ? ch * 8 : 0;
        koff[i] = (uint32_t)(row * k_rs + ld_col[i]) * 2u;
        voff[i] = (uint32_t)(row * v_rs + ld_col[i]) * 2u;
    }
    // ---- lane parts of the LDS read addresses; everything else is an immediate or one XOR ---------------
    const int r_ = lane_ & 31, hh_ = lane_ >> 5, i16 = lane_ & 15, g1 = (lane_ >> 4) & 1;
    kbase = lds_off<D>(r_, hh_);                                                          // ^ 32*ks, + half/buffer
    vbase = lds_off<D>(4 * hh_ + (i16 >> 2), 2 * g1 + ((i16 >> 1) & 1)) + 8 * (i16 & 1);  // ^ (64 db + 32 j2)
    };
    lane_tables(lane);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t lds_wave = lds0 + wave * (LD_PER_THREAD * 1024);
    // Tile starting at key k0 (may start before 0 or reach past sk): rows clamped into [0, sk).  The wave-uniform
    // 64-bit base points at the first clamped row; lane offsets are 32-bit and never negative.
    auto dma_tile = [&](const T *seq, int row_stride, int64_t row_stride64, const uint32_t (&fast_off)[LD_PER_THREAD],
                        int k0, uint32_t lds) {
        if constexpr (PERSIST) {
            // seqlen_k is a multiple of 64 here and a wave fetches 16 whole rows of a tile (rows 16 wave ..): instead of
            // clamping lanes, the wave's row group is moved into the sequence as a whole -- the rows this replaces (keys in
            // front of a head's first or behind its last) are never multiplied with anything that is kept, they only have
            // to be finite -- and the precomputed in-tile lane offsets serve every tile (no per-lane row / column tables
            // alive across the item loop)
            const int start = min(max(k0 + 16 * wave, 0), sk - 16) - 16 * wave;
            lds_dma<LD_PER_THREAD>(lds, seq + (int64_t)start * row_stride64, fast_off);
            return;
        }
        const int base_row = min(max(k0, 0), sk - 1);
        const T *base = seq + (int64_t)base_row * row_stride64;
        if (k0 >= 0 && k0 + BLOCK_N <= sk) {  // whole tile in range (wave-uniform): precomputed lane offsets
            lds_dma<LD_PER_THREAD>(lds, base, fast_off);
        } else {
            uint32_t off[LD_PER_THREAD];
#pragma unroll
            for (int i = 0; i < LD_PER_THREAD; ++i) {
                const int rel = min(max(k0 + ld_row[i], 0), sk - 1) - base_row;  // 0..63
                off[i] = (uint32_t)(rel * row_stride + ld_col[i]) * 2u;
            }
            lds_dma<LD_PER_THREAD>(lds, base, off);
        }
    };
    // K tile m = keys [64m - 32, 64m + 32)
    // PERSIST: tiles behind the item's last one are the NEXT item's first ones (its tile n_min_nx + i; 0 without a left window).  K tile n_max is a hybrid:
    // rows 0..31 = this item's last 32 keys (waves 0 and 1 fetch them), rows 32..63 = the next item's keys 0..31 (waves 2, 3).
    auto load_k = [&](int m, int buf) {
        if (PERSIST && has_next && (m > n_max || (m == n_max && wave >= 2)))
            dma_tile(kp_n, k_rs, k_rs64, koff, (m - n_max + n_min_nx) * BLOCK_N - 32, lds_wave + buf * TILE_BYTES);
        else
            dma_tile(kp, k_rs, k_rs64, koff, m * BLOCK_N - 32, lds_wave + buf * TILE_BYTES);
    };
    auto load_v = [&](int n, int buf) {
        if (PERSIST && has_next && n >= n_max)
            dma_tile(vp_n, v_rs, v_rs64, voff, (n - n_max + n_min_nx) * BLOCK_N, lds_wave + (3 + buf) * TILE_BYTES);
        else
            dma_tile(vp, v_rs, v_rs64, voff, n * BLOCK_N, lds_wave + (3 + buf) * TILE_BYTES);
    };


    // S(half kh of the K tile in buffer kbuf) for both q-blocks; every K fragment feeds two MFMAs
    auto qk_half = [&](int kbuf, int kh, f32x16 &sa, f32x16 &sb) {
        const char *base = smem + kbuf * TILE_BYTES + kh * (32 * ROWB);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const u32x4 kf = *(const u32x4 *)(base + (kbase ^ (32 * ks)));
            if (ks == 0) {
                Mfma<T>::s_first_pad(sa, kf, qa[ks]);
                Mfma<T>::s_first_pad(sb, kf, qb[ks]);
            } else {
                Mfma<T>::s_acc_pad(sa, kf, qa[ks]);
                Mfma<T>::s_acc_pad(sb, kf, qb[ks]);
            }
        }
    };
    // the same for the very first scores: all K fragments requested up front (nothing else hides the LDS latency there)
    auto qk_half_first = [&](int kbuf, int kh, f32x16 &sa, f32x16 &sb) {
        const char *base = smem + kbuf * TILE_BYTES + kh * (32 * ROWB);
        u32x4 kf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) kf[ks] = *(const u32x4 *)(base + (kbase ^ (32 * ks)));
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks == 0) {
                Mfma<T>::s_first_pad(sa, kf[ks], qa[ks]);
                Mfma<T>::s_first_pad(sb, kf[ks], qb[ks]);
            } else {
                Mfma<T>::s_acc_pad(sa, kf[ks], qa[ks]);
                Mfma<T>::s_acc_pad(sb, kf[ks], qb[ks]);
            }
        }
    };
    // O^T += V^T.P^T over half kh of the V tile in buffer vbuf; every V^T fragment feeds two MFMAs.
    // Element j of lane half hh of 16-key step st is key 16st + 8(j>>2) + 4hh + (j&3).
    // s0/s1: the score tiles written by the preceding asm MFMAs (fenced behind the first MFMA issued here).
    auto pv_half = [&](int vbuf, int kh, const u32x4 (&pa)[2], const u32x4 (&pb)[2], f32x16 &s0, f32x16 &s1) {
        const char *base = smem + (3 + vbuf) * TILE_BYTES + kh * (32 * ROWB);
#pragma unroll
        for (int db = 0; db < DBLOCKS; ++db) {
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
                if (db == 0 && st == 0) Mfma<T>::o_acc_fence(oa[db], vf, pa[st], s0, s1);
                else Mfma<T>::o_acc_pad(oa[db], vf, pa[st]);
                Mfma<T>::o_acc_pad(ob[db], vf, pb[st]);
            }
        }
    };
    // softcap + mask of a fresh score half (j = half-step index); mask only on boundary halves
    auto half_needs_mask = [&](int j) -> bool {
        const int k0 = n_min * BLOCK_N + 32 * j;
        bool need = (k0 + 32 > sk);
        if (p.window_right >= 0) need = need || (k0 + 31 > wrow + shift + p.window_right);
        if (p.window_left >= 0) need = need || (k0 < wrow + 63 + shift - p.window_left);
        return need;
    };
    auto prep_scores = [&](int j, f32x16 &sa, f32x16 &sb) {
        if constexpr (SOFTCAP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sa[i] = fast_tanh(sa[i] * sc.softcap_pre);
                sb[i] = fast_tanh(sb[i] * sc.softcap_pre);
            }
        }
        // (both branches rebuild their lane constants from the lane id: kept live across the main loop they are spilled
        //  to scratch, and the reload is a memory round trip in front of every masked half-step)
        if (p.alibi) {  // wave-uniform; bias on the (soft-capped) score, before masking: src/mask.h:156-186
            int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if constexpr (PERSIST) asm volatile("" : "+v"(ln));  // (not hoisted out of the item loop: see lane_tables)
            const int k0 = n_min * BLOCK_N + 32 * j + 4 * (ln >> 5);
            const int rel_a = wrow + (ln & 31) + shift - k0, rel_b = rel_a + 32;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = (i & 3) + 8 * (i >> 2);
                sa[i] -= alibi * fabsf((float)(rel_a - key));
                sb[i] -= alibi * fabsf((float)(rel_b - key));
            }
        }
        if (half_needs_mask(j)) {
            int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if constexpr (PERSIST) asm volatile("" : "+v"(ln));  // (not hoisted out of the item loop: see lane_tables)
            const int k0 = n_min * BLOCK_N + 32 * j + 4 * (ln >> 5);
            const int ra = wrow + (ln & 31) + shift, rb = ra + 32;  // diagonal key of this lane's rows
            int hi_a = sk, hi_b = sk, lo_a = 0, lo_b = 0;  // [lo, hi) visible
            if (p.window_right >= 0) {
                hi_a = min(sk, ra + p.window_right + 1);
                hi_b = min(sk, rb + p.window_right + 1);
            }
            if (p.window_left >= 0) {
                lo_a = max(0, ra - p.window_left);
                lo_b = max(0, rb - p.window_left);
            }
            hi_a -= k0; hi_b -= k0; lo_a -= k0; lo_b -= k0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = (i & 3) + 8 * (i >> 2);
                if (key >= hi_a || key < lo_a) sa[i] = -INFINITY;
                if (key >= hi_b || key < lo_b) sb[i] = -INFINITY;
            }
        }
    };
    // online softmax of one 32x32 score block; lane = query row.  Writes P^T fragments (the B operand of
    // the PV product), updates (m, l); alpha = factor the O accumulator still has to be multiplied with.
    auto softmax = [&](f32x16 &s, u32x4 (&pf)[2], float &m_run, float &l_run, float &alpha, bool &moved) {
        float mxa, mxb;
        rowmax16(s, m_run, mxa, mxb);
        const float m_new = half_swap_max(fmaxf(mxa, mxb));  // >= m_run, identical in both lane halves
        float m_eff;
        if constexpr (THR == 0.f) {
            moved = __any(m_new > m_run);
            m_eff = m_new;
        } else {
            moved = __any((m_new - m_run) * csc_arg > THR);  // -inf -> finite counts as moved
            m_eff = moved ? m_new : m_run;
        }
        const float mc = (m_eff == -INFINITY ? 0.f : m_eff) * csc_arg;
        alpha = __builtin_amdgcn_exp2f(m_run * csc_arg - mc);
        m_run = m_eff;
        float ps0 = 0.f, ps1 = 0.f;
        const float csc = csc_arg;  // wave-uniform, lives in an SGPR
#pragma unroll
        for (int i = 0; i < 16; i += 2)
            pf[i >> 3][(i & 7) >> 1] = Exp2Pair<T>::run(s[i], s[i + 1], csc, mc, ps0, ps1);
        l_run = l_run * alpha + (ps0 + ps1);
    };
    // Rare path (a running max grew by more than THR).  O lives in AGPRs and is written by asm MFMAs whose
    // result hazard hipcc does not pad: drain the matrix pipe, with the accumulators as operands of the
    // drain so no reader can be scheduled above it.
    auto drain_mfma = [&](f32x16 (&o)[DBLOCKS]) {
        if constexpr (DBLOCKS == 4)
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]));
        else
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o[0]), "+a"(o[1]));
    };
    // Same for ALL accumulators at once.  hipcc is free to move an accumulator to another AGPR tuple at any block
    // boundary (v_accvgpr_mov right behind the asm MFMA that produced it = an unpadded MFMA->VALU read); every place
    // where compiler-visible code may follow asm PV MFMAs (generic PV, fast-loop exits) therefore ends in this drain.
    auto drain_all = [&]() {
        if constexpr (DBLOCKS == 4)
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(oa[2]), "+a"(oa[3]),
                         "+a"(ob[0]), "+a"(ob[1]), "+a"(ob[2]), "+a"(ob[3]));
        else
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(ob[0]), "+a"(ob[1]));
    };
    auto rescale = [&](f32x16 (&o)[DBLOCKS], float alpha) {
        drain_mfma(o);
#pragma unroll
        for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
        drain_mfma(o);  // VALU / accvgpr writes -> next MFMA reading them as SrcC
    };

    // ---- prologue: K tiles n_min and n_min+1, V tile n_min; Q; first scores; softmax of A(0) ----------------------
    // Q: this wave's 64 rows by LDS-DMA (CH_PER_ROW coalesced 1-KiB pieces) into its own 64-row tile image behind the K/V
    // rings -- same swizzle as a K tile, so the fragment reads are the K fragment reads -- through a raw buffer descriptor:
    // rows past the end of q and head-dim chunks past d land as zeros.  (Round 1 gathered the fragments lane-per-row from
    // global memory, 32 B of each 128-B line per instruction: 2.4 us until the requests were even issued.)
    if (n_min < n_max) load_k(n_min, 0);
    auto q_dma = [&](const T *qhead, int wrow_) {  // rows [wrow_, wrow_ + 64) of the head at qhead -> this wave's Q image
        const int rows_here = min(sq - wrow_, 64);  // <= 0: nothing of this wave's rows exists
        const uint64_t qb_ = (uint64_t)(uintptr_t)(qhead + (int64_t)wrow_ * p.q_row_stride);
        u32x4 qdesc;
        qdesc[0] = (uint32_t)qb_;
        qdesc[1] = (uint32_t)(qb_ >> 32) & 0xffffu;  // stride 0: raw buffer
        qdesc[2] = rows_here > 0 ? (uint32_t)(((int64_t)(rows_here - 1) * p.q_row_stride + min(p.d, D)) * 2) : 0u;
        qdesc[3] = 0x00020000u;
#pragma unroll
        for (int i = 0; i < 4; ++i) qdesc[i] = __builtin_amdgcn_readfirstlane(qdesc[i]);
        const uint32_t q_img = lds0 + 6 * TILE_BYTES + wave * TILE_BYTES;
        const int q_rs = (int)p.q_row_stride;
        // (PERSIST: the lane offsets are rebuilt per item from an opaque copy of the lane id -- hoisted out of the item loop
        //  they are 16 registers that live across the whole sweep, i.e. spills with a vmcnt(0) wait in front of every reload)
        int lane_q = lane;
        if constexpr (PERSIST) {
            lane_q = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane_q));
        }
#pragma unroll
        for (int i = 0; i < CH_PER_ROW; ++i) {
            const int slot = i * 64 + lane_q;
            const int row = slot / CH_PER_ROW;
            int ch;  // inverse of lds_off<D>
            if constexpr (D == 64) ch = (slot % CH_PER_ROW) ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
            else ch = (slot % CH_PER_ROW) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            const uint32_t off = ch * 8 < p.d ? (uint32_t)(row * q_rs + ch * 8) * 2u : 0x7ffffff0u;
            lds_dma_buf1(q_img + i * 1024, qdesc, off);
        }
    };
    q_dma(qp, wrow);
    if (n_min < n_max) {
        load_v(n_min, 0);
        load_k(n_min + 1, 1);
        if (n_min + 2 <= n_max) load_k(n_min + 2, 2);
        if (n_min + 1 < n_max) load_v(n_min + 1, 1);
    }
    FA_T(1);
    FA_STAMP_VAR(48);  // Q and the first K/V tiles requested
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    FA_STAMP_VAR(49);
    tile_barrier<0>();                   // first tiles landed (asm LDS-DMA) and visible to every wave
    FA_STAMP_VAR(50);
    auto q_fragments = [&]() {
        const char *qimg = smem + 6 * TILE_BYTES + wave * TILE_BYTES;
        int kb_ = kbase;
        if constexpr (PERSIST) asm volatile("" : "+v"(kb_));  // (the 8 addresses are rebuilt per item, not kept alive across the sweep)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            qa[ks] = *(const u32x4 *)(qimg + (kb_ ^ (32 * ks)));
            qb[ks] = *(const u32x4 *)(qimg + (kb_ ^ (32 * ks)) + 32 * ROWB);
        }
        // Q is only ever an MFMA operand: pin it into the AGPR half of the register file (born there, stays there)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            asm volatile("; pin Q" : "+a"(qa[ks]));
            asm volatile("; pin Q" : "+a"(qb[ks]));
        }
    };
    if constexpr (!PERSIST) q_fragments();
    FA_STAMP_FLUSH(46); FA_STAMP_FLUSH(48); FA_STAMP_FLUSH(49); FA_STAMP_FLUSH(50); FA_STAMP_ENTRY_FLUSH(53);
    FA_STAMP_HWID(54);
    FA_T(2);
    // ring phase: K / V tile i of the current item lives in ring slot (i + ph) % 3 (PERSIST: the rings run on across items)
    int ph = 0;
#ifdef FA_CYCLES
    int fa_item = 0;
#endif
    // PERSIST: at an item switch the rings hold the next item's first tiles and the Q image its Q: no LDS to stage O through.
    // O leaves straight from the accumulators: lane (row r, half hh) holds 4 consecutive head dims of every 8 (dims 8 c + 4 hh
    // ..); one v_permlane32_swap per packed dword between the registers of chunks c and c + 1 gives lane (r, hh) all 8 dims of
    // chunk c + hh -- 16 bytes, one buffer store: a store instruction writes 32 bytes of each of 32 rows (whole 32-byte
    // sectors, merged to full lines in L2).  The raw descriptor covers this wave's rows: rows past the end of q and chunks
    // past d fall to the range check -- no predicates, no 64-bit lane addresses.
    // (The 16 stores are what the epilogue waits for: ~0.7 us of packing, ~1.1 us of the memory pipe pushing back, 32 row
    //  segments per instruction.  Spreading them over the item switch -- four at a time between the Q fragment reads, the
    //  decode and the Q prefetch -- was measured worse, 7.5 -> 8.9 us per switch: the prefetch then queues behind them.)
    auto epilogue_persist = [&]() {
        int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_e));  // (not hoisted out of the item loop: see lane_tables)
        const int r_e = lane_e & 31, hh_e = lane_e >> 5;
        const int row_a_e = wrow + r_e, row_b_e = wrow + 32 + r_e;
        const float lt_a = half_swap_sum(l_a), lt_b = half_swap_sum(l_b);
        const bool e_a = (lt_a == 0.f) || (lt_a != lt_a), e_b = (lt_b == 0.f) || (lt_b != lt_b);
        const float inv_a = (e_a ? 1.f : 1.f / lt_a) * vdesc_e, inv_b = (e_b ? 1.f : 1.f / lt_b) * vdesc_e;
        if (wrow >= sq) return;
        if (hh_e == 0) {
            if (row_a_e < sq) p.lse[lse_base + row_a_e] = e_a ? INFINITY : m_a * scale_e + __logf(lt_a);
            if (row_b_e < sq) p.lse[lse_base + row_b_e] = e_b ? INFINITY : m_b * scale_e + __logf(lt_b);
        }
        FA_PSTAMP(58);  // LSE written
        const int rows_here = min(sq - wrow, 64);
        const __amdgpu_buffer_rsrc_t odesc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(op + (int64_t)wrow * p.o_row_stride), 0,
            (int)(((int64_t)(rows_here - 1) * p.o_row_stride + min(p.d, D)) * 2), 0x00020000);
        const uint32_t o_rs2 = (uint32_t)p.o_row_stride * 2u;
        const uint32_t off_a = (uint32_t)r_e * o_rs2 + (uint32_t)hh_e * 16u, off_b = off_a + 32u * o_rs2;
        auto put = [&](const f32x16 &o, int db, int g4, float inv, uint32_t row_off) {  // chunks 4 db + g4 and + 1 (g4 even)
            const uint32_t x0 = Elem<T>::pack2(o[4 * g4] * inv, o[4 * g4 + 1] * inv), x1 = Elem<T>::pack2(o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv);
            const uint32_t y0 = Elem<T>::pack2(o[4 * g4 + 4] * inv, o[4 * g4 + 5] * inv), y1 = Elem<T>::pack2(o[4 * g4 + 6] * inv, o[4 * g4 + 7] * inv);
            // upper lanes of x <-> lower lanes of y: (r, 0) ends with its own half and (r, 1)'s of chunk c, (r, 1) with both of c + 1
            const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            const u32x4 v4 = {s0[0], s1[0], s0[1], s1[1]};
            const int ch = 4 * db + g4 + hh_e;
            const uint32_t off = ch * 8 < p.d ? row_off + (uint32_t)(4 * db + g4) * 16u : 0x7ffffff0u;
#ifdef FA_EPI_ABLATE_STORES   // developer-only timing ablation (results are WRONG): the epilogue without its stores
            asm volatile("" :: "v"(v4), "v"(off));
#else
            __builtin_amdgcn_raw_buffer_store_b128(v4, odesc, off, 0, 0);
#endif
        };
#pragma unroll
        for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4 += 2) {
                put(oa[db], db, g4, inv_a, off_a);
                put(ob[db], db, g4, inv_b, off_b);
            }
    };
    bool first_item = true;
    for (;;) {  // work items of this workgroup (exactly one unless PERSIST)
    if constexpr (PERSIST) {
        FA_PSTAMP(59);  // loop top
        {
            int lane_i = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane_i));
            lane_tables(lane_i);
        }
        FA_PSTAMP(60);  // lane tables rebuilt
        if (!first_item) {
            epilogue_persist();  // the previous item's O and LSE
            FA_PSTAMP(57);  // O stores issued
            if (!has_next) break;
            // ---- item switch: the rings hold the next item's K tiles 0..2 and V tiles 0, 1 (published by the barrier above), its
            //      Q fragments are in registers; O and the softmax state start over.  (The O stores above are still in flight:
            //      they are older than every LDS-DMA piece to come, so the counted vmcnt waits only ever over-wait for them.)
            ph = __builtin_amdgcn_readfirstlane((ph + (n_max - n_min)) % 3);
            m_block = m_block_n; head = head_n; batch = batch_n;
            kv_head = head / p.h_ratio;
            qp = (const T *)p.q + (int64_t)batch * p.q_batch_stride + (int64_t)head * p.q_head_stride;
            kp = kp_n;
            vp = vp_n;
            op = (T *)p.o + (int64_t)batch * p.o_batch_stride + (int64_t)head * p.o_head_stride;
            lse_base = ((int64_t)batch * p.h + head) * p.seqlen_q;
            row_lo = m_block * BLOCK_M;
            item_geometry();
            {
                const u32x4 z4 = {0, 0, 0, 0};
#pragma unroll
                for (int db = 0; db < DBLOCKS; ++db) {
                    Mfma<T>::o_zero(oa[db], z4);
                    Mfma<T>::o_zero(ob[db], z4);
                }
                asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oa[0]), "+a"(oa[1]), "+a"(oa[2]), "+a"(oa[3]),
                             "+a"(ob[0]), "+a"(ob[1]), "+a"(ob[2]), "+a"(ob[3]));
            }
            m_a = -INFINITY; m_b = -INFINITY;
            l_a = 0.f; l_b = 0.f;
#ifdef FA_CYCLES
            ++fa_item;
            FA_PSTAMP(55);  // item switch done: O stored (issued), state reset
#endif
        }
        first_item = false;
        // Q of the item to run now (first item: landed behind the prologue's barrier; later ones: prefetched during the previous
        // sweep, landed behind its closing barrier) moves into the fragment registers, which frees the Q image for ...
        q_fragments();
        // ... the next item of the chain: its K / V heads for the look-ahead stream, its Q rows into the Q image
        chain_next();
        if (has_next) {
            const int kv_head_n = head_n / p.h_ratio;
            kp_n = (const T *)p.k + (int64_t)batch_n * p.k_batch_stride + (int64_t)kv_head_n * p.k_head_stride;
            vp_n = (const T *)p.v + (int64_t)batch_n * p.v_batch_stride + (int64_t)kv_head_n * p.v_head_stride;
            q_dma((const T *)p.q + (int64_t)batch_n * p.q_batch_stride + (int64_t)head_n * p.q_head_stride,
                  m_block_n * BLOCK_M + wave * 64);
        }
        FA_PSTAMP(56);  // next item known, its Q requested; the first scores follow
    }
    int in_flight = 0;                   // LDS-DMA pieces this wave issued at the top of the current tile

    // Pipeline state at the boundary in front of half-step j (canonical naming):
    //   sbx = S_B(j) (masked, not yet exponentiated), pax = P_A(j), (m_a, l_a) through j, (m_b, l_b) through
    //   j-1, and O_A still owes the factor alpha_a when moved_a.
#ifdef FA_CYCLES
    int cyc_n = 0;
#endif
    f32x16 sa, sbx, sby;
    u32x4 pax[2], pay[2], pb[2];
    float alpha_a = 1.f, alpha_b = 1.f;
    bool moved_a = false, moved_b = false;
    if (jend > 0) {
        qk_half_first(ph, 1, sa, sbx);  // half-step 0 = second half of K tile n_min
        drain_scores(sa, sbx);
        prep_scores(0, sa, sbx);
        softmax(sa, pax, m_a, l_a, alpha_a, moved_a);
        moved_a = false;  // O_A is still zero: nothing to rescale
        // B's running max starts at the max of its first block (l_b stays 0): P_B(0) <= 1 whichever path exponentiates it,
        // so the fast path may take over from the very first tile
        float xa, xb;
        rowmax16(sbx, m_b, xa, xb);
        m_b = half_swap_max(fmaxf(xa, xb));
    }
    __syncthreads();  // every wave has read K tile n_min before tile n_min's end overwrites its buffer
    FA_T(3);

    const int J = 2 * (n_max - n_min);
    // ---- generic half-step: any boundary case (masks, last half-steps of the wave, rescales, redo after the fast
    //      path tripped its overflow guard). ---------------------------------------------------------------------
    bool redo_a = false;      // P_A(j) / l_a were produced with a stale max that turned out too small: redo from sa
    float l_a_saved = 0.f;
    auto generic_half = [&](int j) {
        const int i = j >> 1, kb = j & 1, slot = (i + ph) % 3, n = n_min + i;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot == 0 ? 2 : slot - 1;  // (slot+1)%3, (slot+2)%3
        if (kb == 0) {  // K tile n+3 over K tile n, V tile n+2 over V tile n-1 (both last read during tile n-1)
            in_flight = 0;
            if (n + 3 <= n_max || (PERSIST && has_next)) { load_k(n + 3, slot); in_flight += LD_PER_THREAD; }
            if (n + 2 < n_max || (PERSIST && has_next)) { load_v(n + 2, slot2); in_flight += LD_PER_THREAD; }
        }
        if (j < jend) {
            if (redo_a) {  // sa still holds S_A(j)
                l_a = l_a_saved;
                softmax(sa, pax, m_a, l_a, alpha_a, moved_a);
                redo_a = false;
            }
            if (moved_a) rescale(oa, alpha_a);  // deferred from softmax A(j)
            moved_a = false;
            if (j + 1 < jend) {
                qk_half(slot1, kb, sa, sby);
                softmax(sbx, pb, m_b, l_b, alpha_b, moved_b);
                if (moved_b) rescale(ob, alpha_b);
                drain_scores(sa, sby);
                prep_scores(j + 1, sa, sby);
                pv_half(slot, kb, pax, pb, sa, sby);
                drain_all();
                softmax(sa, pay, m_a, l_a, alpha_a, moved_a);
                sbx = sby;
                pax[0] = pay[0];
                pax[1] = pay[1];
            } else {
                softmax(sbx, pb, m_b, l_b, alpha_b, moved_b);
                if (moved_b) rescale(ob, alpha_b);
                pv_half(slot, kb, pax, pb, sa, sby);
                drain_all();
            }
        }
        // tiles n+1's K/V (issued one tile ago) have landed; every wave is done with this tile's buffers
        if (kb == 1) tile_barrier_n(in_flight, LD_PER_THREAD);
    };
    // ---- fast half-step: interior of the sweep (no masks, a next half-step exists).  The running max is NOT
    //      recomputed here: P = exp2(S*c - m_stale*c) and the per-lane partial row sums, which are needed anyway,
    //      double as the guard -- a partial sum above 2^THR means some score outgrew the stale max by up to THR
    //      (or more) and the block is redone by the generic path with a fresh max.  So O is touched by nothing but
    //      the AGPR-pinned MFMAs, and the VALU work per 32x32 block is fma+exp+add per score plus one cvt_pk per pair.
    //      Hand-placed schedule: the wave is alone on its SIMD and issues in order, so every pair of MFMAs is
    //      followed by a fixed slice of that work and the slices are pinned with sched_barrier.
    //      (B's P is consumed in the same half-step, so B keeps a cheap max look-ahead in phase 2 instead of the sum
    //      guard.)  Returns true when the NEXT half-step must take the generic path; redo_a says A must be redone. --
    constexpr float LIM = (THR > 0.f) ? (float)(1u << (int)THR) : 1.0f;
    // kf0/kf1: the first two K fragments of this half-step, fetched by the previous half-step (KB = 1) or by the
    // driver right behind the tile barrier (KB = 0); a KB = 0 half-step leaves the next one's in them.
    // kdma / vdma (+ per-lane byte offsets): wave-uniform source bases of K tile n+3 / V tile n+2; their LDS-DMA pieces
    // are issued one per slice of the KB = 0 half-step, behind the slice's MFMAs.  Look-ahead tiles that reach past the
    // sequence (or are not needed at all) are fetched with rows clamped to the last valid one, like the generic path.
    auto fast_half = [&](auto slot_c, auto kb_c, f32x16 &sb_cur, f32x16 &sb_nxt, u32x4 (&pa_cur)[2],
                         u32x4 (&pa_nxt)[2], u32x4 &kf0, u32x4 &kf1, const T *kdma, const T *vdma,
                         const uint32_t (&kdma_off)[LD_PER_THREAD], const uint32_t (&vdma_off)[LD_PER_THREAD]) -> bool {
        constexpr int SLOT = decltype(slot_c)::value, KB = decltype(kb_c)::value;
        constexpr uint32_t KDST = SLOT * TILE_BYTES, VDST = (3 + (SLOT + 2) % 3) * TILE_BYTES;
        const char *kb_base = smem + ((SLOT + 1) % 3) * TILE_BYTES + KB * (32 * ROWB);
        const char *vb_base = smem + (3 + SLOT) * TILE_BYTES + KB * (32 * ROWB);
        auto k_frag = [&](int ks) {
            if (FA_ABLATE & 16) return qa[ks];  // timing only: no LDS fragment reads
            return *(const u32x4 *)(kb_base + (kbase ^ (32 * ks)));
        };
        auto v_frag = [&](int t) {  // step t = (db, st)
            const int db = t >> 1, st = t & 1;
            u32x4 vf;
            if (FA_ABLATE & 16) return qb[t];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
                    vb_base + (vbase ^ (64 * db + 32 * j2)) + (16 * st + 8 * j2) * ROWB));
                const u32x2 x2 = __builtin_bit_cast(u32x2, x);
                vf[2 * j2] = x2[0];
                vf[2 * j2 + 1] = x2[1];
            }
            return vf;
        };
        const float csc = csc_arg;  // wave-uniform, lives in an SGPR

        FA_C();
        // ---------- phase 1: S(j+1) = K.Q^T on the matrix pipe || exp/sum/pack of B(j) on the VALU ----------
        // LDS fragments are fetched two slices (>= 128 cycles) ahead of the MFMA that consumes them; the first
        // two V^T fragments of phase 2 are fetched during the last two slices of phase 1.
        u32x4 vfa, vfb;  // V^T fragments handed from phase 1 to phase 2
        {
            constexpr int PER = 16 / KSTEPS;  // score elements of B(j) finished per k-step
            const float mcb = (m_b == -INFINITY ? 0.f : m_b) * csc;
            float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                u32x4 kf2 = kf1;
                if (ks + 2 < KSTEPS) kf2 = k_frag(ks + 2);
                else if (ks + 2 == KSTEPS) vfa = v_frag(0);
                else vfb = v_frag(1);
                // MFMA A ; fma fma exp exp ; MFMA B ; add add cvt   (+ further pairs when PER > 2)
                const int e = ks * PER;
                if (FA_ABLATE & 1) {
                    if (ks == 0) { Mfma<T>::s_first(sa, kf0, qa[0]); Mfma<T>::s_first(sb_nxt, kf0, qb[0]); }
                    else { Mfma<T>::s_acc(sa, kf0, qa[ks]); Mfma<T>::s_acc(sb_nxt, kf0, qb[ks]); }
                } else {
                    if (ks == 0)
                        pb[e >> 3][(e & 7) >> 1] = Fused<T>::template score_slice<true>(
                            sa, qa[0], sb_nxt, qb[0], kf0, sb_cur[e], sb_cur[e + 1], csc, mcb, ps0, ps1);
                    else
                        pb[e >> 3][(e & 7) >> 1] = Fused<T>::template score_slice<false>(
                            sa, qa[ks], sb_nxt, qb[ks], kf0, sb_cur[e], sb_cur[e + 1], csc, mcb, ps0, ps1);
#pragma unroll
                    for (int e2 = e + 2; e2 < e + PER; e2 += 2)
                        pb[e2 >> 3][(e2 & 7) >> 1] = Exp2Pair<T>::run(sb_cur[e2], sb_cur[e2 + 1], csc, mcb, ps0, ps1);
                }
                if (KB == 0 && ks < LD_PER_THREAD && !(FA_ABLATE & 4))
                    lds_dma1(lds_wave + KDST + ks * 1024, kdma, kdma_off[ks < LD_PER_THREAD ? ks : 0]);
                kf0 = kf1;
                kf1 = kf2;
                __builtin_amdgcn_sched_barrier(0);
            }
            l_b += ps0 + ps1;
        }

        FA_C();
        // ---------- phase 2: O += V^T.P^T on the matrix pipe || exp/sum/pack of A(j+1) on the VALU ----------
        {
            constexpr int NSTEP = 2 * DBLOCKS;  // (db, st) steps, two MFMAs each
            constexpr int PER = 16 / NSTEP;     // score elements of A(j+1) finished per step
            const float mca = (m_a == -INFINITY ? 0.f : m_a) * csc;
            float ps0 = 0.f, ps1 = 0.f, nxa = 0.f, nxb = 0.f;
            u32x4 vf = vfa, vf_next = vfb;
#pragma unroll
            for (int t = 0; t < NSTEP; ++t) {
                u32x4 vf_next2 = vf_next;
                if (t + 2 < NSTEP) vf_next2 = v_frag(t + 2);
                const int db = t >> 1, st = t & 1;
                // the first K fragments of the NEXT half-step ride in the last two slices (same tile: no barrier between)
                if (KB == 0 && t == NSTEP - 2) kf0 = *(const u32x4 *)(kb_base + 32 * ROWB + (kbase ^ 0));
                if (KB == 0 && t == NSTEP - 1) kf1 = *(const u32x4 *)(kb_base + 32 * ROWB + (kbase ^ 32));
                const int e = t * PER;
                if (FA_ABLATE & 2) {
                    if (t == 0) Mfma<T>::o_acc_fence(oa[db], vf, pa_cur[st], sa, sb_nxt);
                    else Mfma<T>::o_acc(oa[db], vf, pa_cur[st]);
                    Mfma<T>::o_acc(ob[db], vf, pb[st]);
                } else {
                    if (t == 0)
                        pa_nxt[e >> 3][(e & 7) >> 1] = Fused<T>::template out_slice<true>(
                            oa[db], pa_cur[st], ob[db], pb[st], vf, sa[e], sa[e + 1], csc, mca, ps0, ps1);
                    else
                        pa_nxt[e >> 3][(e & 7) >> 1] = Fused<T>::template out_slice<false>(
                            oa[db], pa_cur[st], ob[db], pb[st], vf, sa[e], sa[e + 1], csc, mca, ps0, ps1);
#pragma unroll
                    for (int e2 = e + 2; e2 < e + PER; e2 += 2)
                        pa_nxt[e2 >> 3][(e2 & 7) >> 1] = Exp2Pair<T>::run(sa[e2], sa[e2 + 1], csc, mca, ps0, ps1);
                    if (t == NSTEP / 2) rowmax16(sb_nxt, m_b, nxa, nxb);  // look-ahead max of B(j+1)
                }
                if (KB == 0 && t < LD_PER_THREAD && !(FA_ABLATE & 4))
                    lds_dma1(lds_wave + VDST + t * 1024, vdma, vdma_off[t < LD_PER_THREAD ? t : 0]);
                vf = vf_next;
                vf_next = vf_next2;
                __builtin_amdgcn_sched_barrier(0);
            }
            const float ps = ps0 + ps1;
            l_a_saved = l_a;
            l_a += ps;
            redo_a = __any(!(ps <= LIM));  // also catches inf / NaN
            float nb;
            asm("v_max_f32 %0, %1, %2" : "=v"(nb) : "v"(nxa), "v"(nxb));
            const float m_new_b = half_swap_max(nb);
            const bool b_moves = __any((m_new_b - m_b) * csc > THR);
            return redo_a || b_moves;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // last half-step index whose scores need no mask (for this wave), and the fast limit: every half-step
    // jj of a fast tile pair [j, j+4) needs jj + 1 <= fast_last
    int fast_last = -1;
    int fast_first = 0;  // first half-step index whose scores need no LEFT-window mask (0 without a left window)
    if (!SOFTCAP && jend > 0 && !p.alibi) {
        int nomask = (sk - n_min * BLOCK_N) / 32 - 1;
        if (p.window_right >= 0) {
            const int t = wrow + shift + p.window_right - 31 - n_min * BLOCK_N;
            nomask = min(nomask, t >= 0 ? t / 32 : -1);
        }
        fast_last = min(nomask, jend - 1);
        if (p.window_left >= 0) {  // half-step jj needs no left mask iff its first key >= (last row of the wave) + shift - window_left
            const int t = wrow + 63 + shift - p.window_left - n_min * BLOCK_N;
            fast_first = t > 0 ? (t + 31) / 32 : 0;
        }
    }
    fast_last = __builtin_amdgcn_readfirstlane(fast_last);
    fast_first = __builtin_amdgcn_readfirstlane(fast_first);
    // masks the generated block can apply itself: sequence end, causal / right window (no ALiBi or softcap)
    const bool mask_ok = !SOFTCAP && jend > 0 && !p.alibi;
    auto from_ok = [&](int jt) { return jt + 1 >= fast_first; };  // the scores a block starting at jt computes are S(jt + 1) ...

    // Driver.  ONE call site of generic_half (its body is large; inlining it twice wrecks register allocation).
    auto to_canonical_after_odd = [&]() {  // after a KB = 0 fast half-step the next scores / P_A live in sby / pay
        sbx = sby;
        pax[0] = pay[0];
        pax[1] = pay[1];
    };
    using I2 = std::integral_constant<int, 2>;
    int j = 0;
    // A tile whose first half-step is jt may take the fast path when both its half-steps (and the one behind them) need
    // no mask.
    auto tile_ok = [&](int jt) { return jt + 2 <= fast_last && from_ok(jt); };
    bool first = true;  // the prologue leaves the pipeline state a fast tile expects: tile 0 may go straight to the fast loop
    while (j < J) {
        // generic until the next tile boundary (at least one half-step, except in front of tile 0: guarantees progress)
        if (!(first && (tile_ok(0) || (mask_ok && from_ok(0) && jend > 0)))) {
            do {
                generic_half(j);
                ++j;
            } while ((j & 1) != 0 && j < J);
        }
        first = false;
        if ((!tile_ok(j) && !(mask_ok && from_ok(j) && (j & 1) == 0 && j < jend)) || moved_a || redo_a) continue;
        {   // B(j) must be safe to exponentiate with its stale max (inside the loop the look-ahead guarantees it)
            float xa, xb;
            rowmax16(sbx, m_b, xa, xb);
            if (__any(!((half_swap_max(fmaxf(xa, xb)) - m_b) * csc_arg <= THR))) continue;  // (NaN from -inf - -inf: not safe)
        }
        // The bulk of the sweep: the generated asm loop (fa_fwd_loop_gen.h), for every run of >= 2 mask-free tiles whose
        // look-ahead LDS-DMA tiles (K tile n+3, V tile n+2) lie fully inside the sequence.  What it leaves (the last
        // three or so mask-free tiles, whose look-ahead rows have to be clamped) goes through the C++ form below.
        bool tripped = false;
        if constexpr (!SOFTCAP && !(FA_ABLATE & 32)) {
            // (the descriptors address bytes with 32 bits.  Longer sequences -- 1 GiB or more of K or V per (batch, kv head) --
            //  run every half-step through generic_half: correct, about half the speed; the C++ form of the fast loop below is
            //  only compiled for soft-cap kernels)
            const bool addr32 = (int64_t)sk * k_rs64 < (1ll << 30) && (int64_t)sk * v_rs64 < (1ll << 30);
            auto run_block = [&](auto masked_c, int count, bool restore_la) {
                constexpr bool MASKED = decltype(masked_c)::value;
                const int n_cur = n_min + (j >> 1);
                auto make_desc = [&](const T *base, int64_t rs64) {
                    const uint64_t b = (uint64_t)(uintptr_t)base;
                    u32x4 dsc;
                    dsc[0] = (uint32_t)b;
                    dsc[1] = (uint32_t)(b >> 32) & 0xffffu;                           // stride 0: raw buffer
                    dsc[2] = (uint32_t)(((int64_t)(sk - 1) * rs64 + min(p.d, D)) * 2);  // bytes to the end of the last valid row
                    dsc[3] = 0x00020000u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dsc[i] = __builtin_amdgcn_readfirstlane(dsc[i]);
                    return dsc;
                };
                u32x4 kdesc, vdesc;
                uint32_t ktile_nx = 0, vtile_nx = 0, kswc = ~0u, vswc = ~0u, ktile, vtile;
                const uint32_t kstep = (uint32_t)(BLOCK_N * k_rs * 2), vstep = (uint32_t)(BLOCK_N * v_rs * 2);
                if constexpr (PERSIST) {
                    // One raw descriptor per tensor (the host checked: every byte offset fits 32 bits, seqlen_k is a multiple
                    // of 64): an item is a byte offset, and the look-ahead stream switches to the next item's offset at the
                    // tile step the generated block counts down to (kswc / vswc).  The K base lies 32 rows in front of the
                    // tensor -- K tile m of a head starts 64 m - 32 rows behind the head's row 0 -- and only pieces whose lane
                    // rows are >= 32 are ever issued for a tile 0 (the hybrid tile's second half: waves 2 and 3).
                    auto tensor_desc = [&](const void *base, int64_t bs, int64_t hs, int64_t rs64, int64_t lead_rows) {
                        const uint64_t b0 = (uint64_t)(uintptr_t)base - (uint64_t)(lead_rows * rs64 * 2);
                        const int64_t extent = (int64_t)(p.b - 1) * bs + (int64_t)(p.h_k - 1) * hs + (int64_t)(sk - 1 + lead_rows) * rs64 + min(p.d, D);
                        u32x4 dsc;
                        dsc[0] = (uint32_t)b0;
                        dsc[1] = (uint32_t)(b0 >> 32) & 0xffffu;
                        dsc[2] = (uint32_t)(extent * 2);
                        dsc[3] = 0x00020000u;
#pragma unroll
                        for (int i = 0; i < 4; ++i) dsc[i] = __builtin_amdgcn_readfirstlane(dsc[i]);
                        return dsc;
                    };
                    kdesc = tensor_desc(p.k, p.k_batch_stride, p.k_head_stride, k_rs64, 32);
                    vdesc = tensor_desc(p.v, p.v_batch_stride, p.v_head_stride, v_rs64, 0);
                    const uint32_t koff_i = (uint32_t)((const char *)kp - (const char *)p.k), voff_i = (uint32_t)((const char *)vp - (const char *)p.v);
                    const int mk = n_cur + 3, mv = n_cur + 2;       // the tiles the block's first step fetches
                    ktile = koff_i + (uint32_t)mk * kstep;
                    vtile = voff_i + (uint32_t)mv * vstep;
                    if (has_next) {
                        const int k_first = n_max + 1 - (wave >= 2 ? 1 : 0);  // first K tile index that is the next item's
                        kswc = (uint32_t)max(0, k_first - mk);
                        vswc = (uint32_t)max(0, n_max - mv);
                        ktile_nx = (uint32_t)((const char *)kp_n - (const char *)p.k) + (uint32_t)(max(mk, k_first) - n_max + n_min_nx) * kstep;
                        vtile_nx = (uint32_t)((const char *)vp_n - (const char *)p.v) + (uint32_t)(max(mv, n_max) - n_max + n_min_nx) * vstep;
                    }
                } else {
                    kdesc = make_desc(kp, k_rs64);
                    vdesc = make_desc(vp, v_rs64);
                    ktile = (uint32_t)(((n_cur + 3) * BLOCK_N - 32) * k_rs * 2);
                    vtile = (uint32_t)((n_cur + 2) * BLOCK_N * v_rs * 2);
                }
                uint32_t koffb[LD_PER_THREAD], voffb[LD_PER_THREAD];
#pragma unroll
                for (int i = 0; i < LD_PER_THREAD; ++i) {
                    koffb[i] = koff[i] - 1024u * i;
                    voffb[i] = voff[i] - 1024u * i;
                }
                const float csc = csc_arg;
                int done = 0, ra = 0, rb = 0;
                uint64_t redo = 0;
                if constexpr (MASKED) {  // last visible key of this lane's rows, relative to the key base of S(j+1) in its lane half
                    int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    if constexpr (PERSIST) asm volatile("" : "+v"(ln));  // (not hoisted out of the item loop: see lane_tables)
                    const int kb1 = n_min * BLOCK_N + 32 * (j + 1) + 4 * (ln >> 5);
                    const int row = wrow + (ln & 31) + shift;
                    ra = (p.window_right >= 0 ? min(sk - 1, row + p.window_right) : sk - 1) - kb1;
                    rb = (p.window_right >= 0 ? min(sk - 1, row + 32 + p.window_right) : sk - 1) - kb1;
                }
#ifdef FA_CYCLES
                unsigned long long *cb_ = (unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024) + (threadIdx.x >> 6) * 64;
                if (!MASKED && !PERSIST) { cb_[40] = __builtin_amdgcn_s_memtime(); cb_[42] = __builtin_amdgcn_s_memrealtime(); cb_[44] = count; }
                if (PERSIST) FA_PSTAMP(MASKED ? 44 : 42);
#endif
                // (PERSIST: THR / c from an opaque copy of c -- as a loop invariant of the item loop hipcc keeps the quotient in
                //  a spilled register, and its reload in front of every block waits for the Q prefetch, vmcnt(0))
                float csc_q = csc;
                if constexpr (PERSIST) asm volatile("" : "+s"(csc_q));
                if constexpr (PERSIST)
                    FastLoop128P<T, MASKED>::run(oa, ob, qa, qb, sa, sbx, sby, pax, pay, l_a, l_b, l_a_saved,
                                                 (m_a == -INFINITY ? 0.f : m_a) * csc, (m_b == -INFINITY ? 0.f : m_b) * csc, m_b,
                                                 (uint32_t)kbase, (uint32_t)vbase, koffb, voffb, csc, THR / csc_q, LIM, kdesc,
                                                 vdesc, ktile, vtile, kstep, vstep, lds0, lds_wave, ((j >> 1) + ph) % 3, count, done,
                                                 redo, ra, rb, ktile_nx, vtile_nx, kswc, vswc);
                else {
                using Loop = std::conditional_t<D == 128, FastLoop128<T, DEFF, MASKED>, FastLoop64<T, MASKED>>;
                Loop::run(oa, ob, qa, qb, sa, sbx, sby, pax, pay, l_a, l_b, l_a_saved,
                                                  (m_a == -INFINITY ? 0.f : m_a) * csc, (m_b == -INFINITY ? 0.f : m_b) * csc, m_b,
                                                  (uint32_t)kbase, (uint32_t)vbase, koffb, voffb, csc, THR / csc, LIM, kdesc,
                                                  vdesc, ktile, vtile, kstep, vstep, lds0, lds_wave, ((j >> 1) + ph) % 3, count, done,
                                                  redo, ra, rb);
                }
#ifdef FA_CYCLES
                if (!MASKED && !PERSIST) { cb_[41] = __builtin_amdgcn_s_memtime(); cb_[43] = __builtin_amdgcn_s_memrealtime(); cb_[45] = done; }
                if (PERSIST) FA_PSTAMP(MASKED ? 45 : 43);
#endif
                j += done;
                in_flight = 2 * LD_PER_THREAD;
                redo_a = redo != 0;
                if (count != 0) {  // a guard tripped: the next half-step is the generic path's
                    tripped = true;
                    if (done & 1) to_canonical_after_odd();
                }
                // The block's last half-step formed a phantom P_A (scores of a half-step this wave does not need: zeros behind a
                // mask, but REAL keys behind an unmasked last tile or behind the end of a split-KV range): only l_a saw it.
                // Also when that very half-step tripped guard A -- the generic path has nothing left to redo (j >= jend) and
                // would keep the phantom's row sums (found under split-KV + sliding window: DESIGN.md 4.1b).
                if (restore_la && j >= jend) {
                    l_a = l_a_saved;
                    redo_a = false;
                }
            };
            // (1) The bulk of the sweep: every run of >= 2 mask-free tiles -- including the wave's LAST tile when its own
            // half-steps need no mask (`phantom`: the block then also forms S(jend) / P_A(jend) from whatever K tile follows,
            // real keys behind a causal diagonal or the zeros that rows past the end of the sequence read as; nothing of that
            // is used, and the one thing it touches, l_a, is restored from l_a_saved).
            if (addr32 && tile_ok(j)) {
                const bool phantom = fast_last == jend - 1 && (jend & 1) == 0 && jend - j >= 2 && !(FA_ABLATE & 64);
                const int count = phantom ? (jend - j) >> 1 : (fast_last - j) >> 1;
                if (count >= 2) run_block(std::false_type{}, count, phantom);
            }
            // (2) What is left of this wave's tiles -- the diagonal tiles under a causal / right-window mask, the tail tile of a
            // sequence that is not a multiple of 64 -- in the MASKED form of the block (the fresh scores get the mask, two
            // VALU instructions per score, before anything reads them).  Whole tiles: a trailing half-step the wave does not
            // need is fully masked and contributes exact zeros; the last half-step's P_A is a phantom as above.
            if (addr32 && mask_ok && from_ok(j) && !tripped && (j & 1) == 0 && j < jend && !(FA_ABLATE & 128))
                run_block(std::true_type{}, (jend + 1 - j) >> 1, true);
        }
        // fast: up to three tiles (one turn of the LDS rings) per iteration; the first turn starts at the ring slot of
        // tile j/2 (`skip` slots are already behind us); every fast tile issues 2 LD_PER_THREAD LDS-DMA pieces, one per
        // slice of its first half-step
        int skip = (j >> 1) % 3;
        if constexpr (SOFTCAP || (FA_ABLATE & 32))
        while (!tripped && tile_ok(j)) {
            const int n = n_min + (j >> 1) - skip;  // the tile in ring slot 0 of this turn
            int done = 0;        // half-steps completed in this iteration
            bool odd_exit = false, x = false, stop = false;
            u32x4 kf0, kf1;      // first two K fragments of the next half-step (fetched behind the tile barrier)
            auto k_prefetch = [&](int kbuf) {
                kf0 = *(const u32x4 *)(smem + kbuf * TILE_BYTES + (kbase ^ 0));
                kf1 = *(const u32x4 *)(smem + kbuf * TILE_BYTES + (kbase ^ 32));
            };
            // source of a look-ahead tile starting at key k0: base of its first (clamped) row + this lane's offsets
            uint32_t ko[LD_PER_THREAD], vo[LD_PER_THREAD];
            auto tile_src = [&](const T *seq, int rs, int64_t rs64, const uint32_t (&in_range)[LD_PER_THREAD], int k0,
                                uint32_t (&off)[LD_PER_THREAD]) -> const T * {
                const int base_row = min(max(k0, 0), sk - 1);
                if (k0 >= 0 && k0 + BLOCK_N <= sk) {  // wave-uniform
#pragma unroll
                    for (int i = 0; i < LD_PER_THREAD; ++i) off[i] = in_range[i];
                } else {
#pragma unroll
                    for (int i = 0; i < LD_PER_THREAD; ++i)
                        off[i] = (uint32_t)((min(max(k0 + ld_row[i], 0), sk - 1) - base_row) * rs + ld_col[i]) * 2u;
                }
                return seq + (int64_t)base_row * rs64;
            };
            auto k_src = [&](int m) { return tile_src(kp, k_rs, k_rs64, koff, m * BLOCK_N - 32, ko); };
            auto v_src = [&](int t) { return tile_src(vp, v_rs, v_rs64, voff, t * BLOCK_N, vo); };
            if (skip <= 0) {
                k_prefetch(1);       // tile slot 0 reads K buffer 1
                x = fast_half(I0{}, I0{}, sbx, sby, pax, pay, kf0, kf1, k_src(n + 3), v_src(n + 2), ko, vo);
                done += 1; odd_exit = true;
                if (!x) {
                    x = fast_half(I0{}, I1{}, sby, sbx, pay, pax, kf0, kf1, nullptr, nullptr, ko, vo);
                    if (!(FA_ABLATE & 8)) tile_barrier<2 * LD_PER_THREAD>();
                    done += 1; odd_exit = false;
                }
            }
            if (!x && skip <= 1) {
                if (done != 0 && !tile_ok(j + done)) stop = true;
                else {
                    k_prefetch(2);
                    x = fast_half(I1{}, I0{}, sbx, sby, pax, pay, kf0, kf1, k_src(n + 4), v_src(n + 3), ko, vo);
                    done += 1; odd_exit = true;
                    if (!x) {
                        x = fast_half(I1{}, I1{}, sby, sbx, pay, pax, kf0, kf1, nullptr, nullptr, ko, vo);
                        if (!(FA_ABLATE & 8)) tile_barrier<2 * LD_PER_THREAD>();
                        done += 1; odd_exit = false;
                    }
                }
            }
            if (!x && !stop) {
                if (done != 0 && !tile_ok(j + done)) stop = true;
                else {
                    k_prefetch(0);
                    x = fast_half(I2{}, I0{}, sbx, sby, pax, pay, kf0, kf1, k_src(n + 5), v_src(n + 4), ko, vo);
                    done += 1; odd_exit = true;
                    if (!x) {
                        x = fast_half(I2{}, I1{}, sby, sbx, pay, pax, kf0, kf1, nullptr, nullptr, ko, vo);
                        if (!(FA_ABLATE & 8)) tile_barrier<2 * LD_PER_THREAD>();
                        done += 1; odd_exit = false;
                    }
                }
            }
            j += done;
            skip = 0;
            in_flight = 2 * LD_PER_THREAD;
            if (x) {
                if (odd_exit) to_canonical_after_odd();
                break;
            }
            if (stop) break;
        }
        drain_all();  // leaving the fast loop: compiler-visible code may touch the accumulators from here on
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    FA_T(4);
    tile_barrier<0>();  // every wave's LDS-DMA (incl. look-ahead tiles past the end) has landed: the ring can be reused
    FA_T(5);
    // ---- L2 prefetch for the workgroup that takes this CU next.  Workgroups are dispatched in id order and (apart from
    // causal imbalance) take equally long, so the successor is id + p.num_cus (read from the device by the host); it lives on the same XCD (same id mod
    // 8), i.e. behind the same L2.  Its Q rows are a cold, badly coalesced gather (~4 us: tools/wg_phases.py); touching one
    // dword of each of their cache lines here, with the whole epilogue (~3 us) in front of the end of this workgroup, turns
    // that into L2 hits.  The data lands in a 1 KiB dump area of LDS; dense batches only (no cu_seqlens lookups here).
    FA_STAMP(51);  // key sweep done
    if (!PERSIST && !p.cu_seqlens_q && !p.seqused_q && p.num_splits <= 1 && p.q_row_stride < (1 << 20)) {
        const int wg2 = blockIdx.x + p.num_cus;
        const int tile2 = tile_of_wg(p, wg2);
        if (wg2 < p.grid && tile2 < p.num_tiles) {
            const int per_kvh = p.h_ratio * p.num_m_blocks;
            const int bk2 = tile2 / per_kvh, r2 = tile2 % per_kvh;
            const int m_block2 = p.num_m_blocks - 1 - r2 / p.h_ratio;
            const int head2 = (bk2 % p.h_k) * p.h_ratio + r2 % p.h_ratio;
            const T *q2 = (const T *)p.q + (int64_t)(bk2 / p.h_k) * p.q_batch_stride + (int64_t)head2 * p.q_head_stride +
                          (int64_t)(m_block2 * BLOCK_M) * p.q_row_stride;  // wave-uniform
            int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if constexpr (PERSIST) asm volatile("" : "+v"(ln));  // (not hoisted out of the item loop: see lane_tables)
            const int row2 = min(wave * 64 + ln, p.seqlen_q - 1 - m_block2 * BLOCK_M);  // row inside the block, clamped
            const uint32_t off2 = (uint32_t)(row2 * (int)p.q_row_stride) * 2u;
            const uint32_t dump = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem + 6 * BLOCK_N * D * 2 + wave * 256;
#pragma unroll
            for (int line = 0; line < (D * 2) / 128; ++line) lds_dma_touch(dump, q2, off2 + line * 128);
        }
    }
    drain_all();        // asm MFMA results -> VALU readers
    if constexpr (!PERSIST) {
        // ---- epilogue (inline at the end of the sweep: as a lambda defined in front of the item loop hipcc turned scalar
        //      values of the prologue into vector ones and the asm operands that need them in SGPRs stopped assembling)
        // lane constants are rebuilt from the lane id here: the ones computed in front of the main loop were spilled to
        // scratch by then, and every reload is a separately awaited memory round trip (tools/wg_phases.py: 3.3 us epilogue)
        int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int r_e = lane_e & 31, hh_e = lane_e >> 5;
        const int row_a_e = wrow + r_e, row_b_e = wrow + 32 + r_e;
        const float lt_a = half_swap_sum(l_a), lt_b = half_swap_sum(l_b);
        const bool e_a = (lt_a == 0.f) || (lt_a != lt_a), e_b = (lt_b == 0.f) || (lt_b != lt_b);
        const float inv_a = (e_a ? 1.f : 1.f / lt_a) * vdesc_e, inv_b = (e_b ? 1.f : 1.f / lt_b) * vdesc_e;
        const bool wave_active = wrow < sq;
        if (wave_active) {
            if (hh_e == 0) {
                if (row_a_e < sq) p.lse[lse_base + row_a_e] = e_a ? INFINITY : m_a * scale_e + __logf(lt_a);
                if (row_b_e < sq) p.lse[lse_base + row_b_e] = e_b ? INFINITY : m_b * scale_e + __logf(lt_b);
            }
            if (!PERSIST && p.num_splits > 1) {
                // split-KV partial: fp32 in the caller's workspace (role of out_accum, csrc/flash_attn/flash_api.cpp:297-318), straight
                // from the accumulators -- 4 consecutive head dims = one 16-byte store per lane; the merge launch rounds once
                float *opf = (float *)p.o + o_base + (int64_t)head * p.o_head_stride;
#pragma unroll
                for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int col = db * 32 + 8 * g4 + 4 * hh_e;
                        if (col < p.d) {
                            if (row_a_e < sq)
                                *(float4 *)(opf + (int64_t)row_a_e * p.o_row_stride + col) =
                                    make_float4(oa[db][4 * g4] * inv_a, oa[db][4 * g4 + 1] * inv_a, oa[db][4 * g4 + 2] * inv_a, oa[db][4 * g4 + 3] * inv_a);
                            if (row_b_e < sq)
                                *(float4 *)(opf + (int64_t)row_b_e * p.o_row_stride + col) =
                                    make_float4(ob[db][4 * g4] * inv_b, ob[db][4 * g4 + 1] * inv_b, ob[db][4 * g4 + 2] * inv_b, ob[db][4 * g4 + 3] * inv_b);
                        }
                    }
            } else {
            // O staging: 64 rows per wave, padded rows inside the K/V rings (reused: the sweep is over)
            char *obuf = smem + wave * (64 * O_ROW_BYTES);
#pragma unroll
            for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    u32x2 wa, wb;
                    wa[0] = Elem<T>::pack2(oa[db][4 * g4] * inv_a, oa[db][4 * g4 + 1] * inv_a);
                    wa[1] = Elem<T>::pack2(oa[db][4 * g4 + 2] * inv_a, oa[db][4 * g4 + 3] * inv_a);
                    wb[0] = Elem<T>::pack2(ob[db][4 * g4] * inv_b, ob[db][4 * g4 + 1] * inv_b);
                    wb[1] = Elem<T>::pack2(ob[db][4 * g4 + 2] * inv_b, ob[db][4 * g4 + 3] * inv_b);
                    const int col = (db * 32 + 8 * g4 + 4 * hh_e) * 2;
                    *(u32x2 *)(obuf + r_e * O_ROW_BYTES + col) = wa;
                    *(u32x2 *)(obuf + (32 + r_e) * O_ROW_BYTES + col) = wb;
                }
            }
        }
        // (no workgroup barrier: a wave reads back only its own 64 staged rows, and LDS operations of one wave are in order)
        FA_STAMP(52);  // O normalised and in LDS
        if (wave_active && p.num_splits <= 1) {
            const char *obuf = smem + wave * (64 * O_ROW_BYTES);
            // (LDS reads outside the predicate: all of them are issued before the first store; inside it each read is
            //  waited for in its own exec-masked block -- 16 serial LDS round trips at D = 128)
            constexpr int NCH = (64 * CH_PER_ROW) / 64;
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
                if (wrow + row < sq && ch * 8 < p.d) *(u32x4 *)(op + (int64_t)(wrow + row) * p.o_row_stride + ch * 8) = val[i];
            }
        }
        break;
    }
    }  // work items
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the prefetch's LDS-DMA has landed before this workgroup's LDS is released
#ifdef FA_CYCLES
    if constexpr (!PERSIST) {
    ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[(threadIdx.x >> 6) * 64 + 47] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (blockIdx.x >= FA_CYCLES_WG0 && blockIdx.x < FA_CYCLES_WG0 + 256 * FA_CYCLES_STRIDE && (blockIdx.x - FA_CYCLES_WG0) % FA_CYCLES_STRIDE == 0)
        fa_cycle_buf[(blockIdx.x - FA_CYCLES_WG0) / FA_CYCLES_STRIDE * 256 + threadIdx.x] = ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[threadIdx.x];
    }
#endif
#ifdef FA_TIMING
    FA_T(6);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // O stores of this wave retired
    FA_T(7);
    if (blockIdx.x < 4096 && (threadIdx.x & 63) < 8)
        fa_timing_buf[blockIdx.x * 32 + (threadIdx.x >> 6) * 8 + (threadIdx.x & 63)] =
            ((unsigned long long *)(smem + 6 * BLOCK_N * D * 2 + 1024))[(threadIdx.x >> 6) * 8 + (threadIdx.x & 63)];
#endif
}

template <int D>
constexpr int smem_bytes_w64() {
    constexpr int kv = 6 * BLOCK_N * D * 2;  // the K/V rings; the O staging of the epilogue (4 x 64 x (2 D + 16)) fits inside
    static_assert(kv >= 4 * 64 * (D * 2 + 16), "O staging must fit the K/V rings");
    // + the Q staging images (4 waves x 64 rows); once the fragments are in registers the front of that region serves as the
    // dump area of the successor's L2 prefetch (4 x 256 B) and, in developer builds, holds the time stamps
    return kv + 4 * BLOCK_N * D * 2;
}

}  // namespace fa
