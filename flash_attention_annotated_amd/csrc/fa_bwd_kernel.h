// fa_bwd_kernel.h — gfx950 (CDNA4) FlashAttention backward, hand-written HIP.
//
// Roles re-derived for MI355X (not translated) from the reference's backward:
//   preprocess  csrc/flash_attn/src/flash_bwd_preprocess_kernel.h:60-127 (dot(dO, O))          -> bwd_dot_kernel
//   main loop   csrc/flash_attn/src/flash_bwd_kernel.h:80-830 (compute_dq_dk_dv_1colblock)       -> bwd_dkdv_kernel
//               + dQ accumulated through atomics / dq_accum and convert_dQ (:flash_bwd_preprocess_kernel.h:232-)
//                                                                                               -> bwd_dq_kernel
//
// With  S = scale * Q K^T (+ softcap, + ALiBi, masked),  P = exp(S - LSE),  D_i = sum_d dO_id O_id:
//   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - D) [o (1 - tanh^2) under softcap],   dQ = scale dS K,   dK = scale dS^T Q
//
// Design.  Two MFMA kernels so that every output element has ONE producer (no atomics, bit-reproducible):
//   * dK/dV: workgroup = 128 keys of one (batch, kv head); wave w owns keys 32w..32w+31 and keeps their K and V
//     fragments in registers for the whole kernel (B operands).  Q and dO tiles (64 rows) of every query head of
//     the GQA group stream through double-buffered, XOR-swizzled LDS.  S = Q K^T and dP = dO V^T land with the KEY
//     on the lane and the query row in the accumulator registers, so P and dS, rounded to 16 bit, ARE the B operands
//     of  dV^T += dO^T P  and  dK^T += Q^T dS  (same register trick as the forward); the transposed A operands
//     (dO^T, Q^T) come from the row-major LDS tiles through ds_read_b64_tr_b16.  dK/dV of a GQA group are summed
//     in fp32 registers (the reference sums 16-bit per-head copies afterwards, flash_api.cpp:964-968).
//   * dQ: the forward's shape -- workgroup = 128 query rows, wave = 32 rows, Q and dO fragments in registers,
//     K/V tiles (64 keys) through LDS; S^T = K Q^T and dP^T = V dO^T have the query on the lane, so LSE and D are
//     per-lane scalars and dS^T is directly the B operand of  dQ^T += K^T dS^T  (K^T through transposing reads of
//     the same K tile).
#pragma once

#include <type_traits>

#include "fa_fwd_kernel.h"
#include "fa_fwd_kernel_w64.h"  // Mfma<T>: inline-asm MFMAs with explicit register classes
#ifdef FA_BWD_LOOP_HEADER          // developer-only: an ablated build of the generated loop (tools/gen_bwd_loop.py --ablate)
#include FA_BWD_LOOP_HEADER
#else
#include "fa_bwd_loop_gen.h"    // BwdLoop128<T>: the generated dK / dV tile loop (tools/gen_bwd_loop.py)
#endif
#include "fa_bwd_dq_loop_gen.h" // BwdDqLoop128<T>: the generated dQ tile loop (tools/gen_bwd_dq_loop.py)

namespace fa {

// Score-tile MFMAs whose B operand stays in arch VGPRs (used where the AGPR half is full of accumulators);
// Mfma<T>::s_first / s_acc take B from AGPRs.  hipcc pads no hazards around these: callers drain before VALU reads.
// (s_nop 1 in front: the B operand may have been copied out of an AGPR by a VALU instruction right before, and hipcc pads
//  nothing for asm MFMAs)
template <typename T> struct BMfma;
template <> struct BMfma<__bf16> {
    static __device__ __forceinline__ void s_first_v(f32x16 &d, u32x4 a, u32x4 b) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void s_acc_v(f32x16 &d, u32x4 a, u32x4 b) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
};
template <> struct BMfma<_Float16> {
    static __device__ __forceinline__ void s_first_v(f32x16 &d, u32x4 a, u32x4 b) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void s_acc_v(f32x16 &d, u32x4 a, u32x4 b) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
};
// asm MFMA results -> VALU readers: 11+ wait states, with the tiles as operands so nothing is scheduled across
template <int NB>
__device__ __forceinline__ void drain_tiles(f32x16 (&a)[NB], f32x16 (&b)[NB]) {
    if constexpr (NB == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a[0]), "+v"(b[0]));
}
template <int N>
__device__ __forceinline__ void drain_acc(f32x16 (&a)[N]) {
    static_assert(N == 2 || N == 4 || N == 5 || N == 6 || N == 8, "accumulator tiles per array");
    if constexpr (N == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a[0]), "+a"(a[1]));
    else if constexpr (N == 4) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]));
    else if constexpr (N == 5) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]));
    else if constexpr (N == 6) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]),
                      "+a"(a[6]), "+a"(a[7]));
}

template <typename L> struct LoopTag { using type = L; };  // (C++17: no std::type_identity)

#ifndef FA_BWD_ABLATE
#define FA_BWD_ABLATE 0  // developer-only: 1 = no generated asm block in bwd_dkdv_kernel
#endif

struct BParams {
    const void *q, *k, *v, *o, *dout;
    const float *lse;
    void *dq, *dk, *dv;
    float *dsum;
    const int32_t *cu_seqlens_q, *cu_seqlens_k;
    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    int64_t do_batch_stride, do_row_stride, do_head_stride;
    int64_t dq_batch_stride, dq_row_stride, dq_head_stride;
    int64_t dk_batch_stride, dk_row_stride, dk_head_stride;
    int64_t dv_batch_stride, dv_row_stride, dv_head_stride;
    int64_t dsum_row_len;
    int32_t b, seqlen_q, seqlen_k, h, h_k, d, total_q;
    int32_t d_v;           // head dim of v / o / dout / dv (= d unless the FA3 headdim_v differs; wide tile only: fa_bwd_validate)
    int32_t h_ratio;
    int32_t num_blocks;    // blocks per (batch, head): key blocks (dK/dV) or query blocks (dQ)
    int32_t num_tiles;     // work list length
    int32_t whole_slots;   // workgroup slots per XCD dealt as whole (batch, head) units; the slots behind them block by block
    int32_t grid;          // workgroups launched (a multiple of 8, see decode_block)
    int32_t window_left, window_right;
    float scale_log2;      // log2(e) * (softmax_scale, or the softcap value under softcap)
    float softcap_pre;     // softmax_scale / softcap (0 when softcap is off)
    float out_scale;       // softmax_scale: the factor of dQ and dK
    const float *alibi;
    int32_t alibi_bs;
    // dropout (as KParams): keep iff the element's random byte <= drop_thr (255 = off)
    int32_t drop_thr;
    float rp_dropout;
    const uint64_t *rng_state;
};

// Work list (batch, head, block) cut into units = all blocks of one (batch, head) (they stream the same operands);
// units are dealt round-robin to the 8 XCDs exactly like tile_of_wg() does for the forward: whole units as far as they
// deal evenly (whole_slots), the remaining heads block by block.
__device__ __forceinline__ bool decode_block(const BParams &p, int &block, int &head, int &batch, int heads) {
    const int wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3;
    const int tile = slot < p.whole_slots ? ((slot / p.num_blocks) * 8 + xcd) * p.num_blocks + slot % p.num_blocks : slot * 8 + xcd;
    if (tile >= p.num_tiles) return false;
    const int bh = tile / p.num_blocks;
    block = p.num_blocks - 1 - tile % p.num_blocks;
    batch = bh / heads;
    head = bh % heads;
    return true;
}

struct BSeq {
    int sq, sk;
    int64_t q_base, k_base, v_base, o_base, do_base, dq_base, dk_base, dv_base;
    int64_t stat_base;  // + head * stat_head_stride + row : index into softmax_lse
    int64_t dsum_base;  // same for softmax_d
    int64_t lse_hs, dsum_hs;
};
__device__ __forceinline__ BSeq bwd_seq(const BParams &p, int batch) {
    BSeq s;
    if (p.cu_seqlens_q) {
        const int q0 = p.cu_seqlens_q[batch], k0 = p.cu_seqlens_k[batch];
        s.sq = p.cu_seqlens_q[batch + 1] - q0;
        s.sk = p.cu_seqlens_k[batch + 1] - k0;
        s.q_base = (int64_t)q0 * p.q_row_stride;
        s.o_base = (int64_t)q0 * p.o_row_stride;
        s.do_base = (int64_t)q0 * p.do_row_stride;
        s.dq_base = (int64_t)q0 * p.dq_row_stride;
        s.k_base = (int64_t)k0 * p.k_row_stride;
        s.v_base = (int64_t)k0 * p.v_row_stride;
        s.dk_base = (int64_t)k0 * p.dk_row_stride;
        s.dv_base = (int64_t)k0 * p.dv_row_stride;
        s.stat_base = q0;
        s.dsum_base = q0;
        s.lse_hs = p.total_q;
        s.dsum_hs = p.dsum_row_len;
    } else {
        s.sq = p.seqlen_q;
        s.sk = p.seqlen_k;
        s.q_base = (int64_t)batch * p.q_batch_stride;
        s.o_base = (int64_t)batch * p.o_batch_stride;
        s.do_base = (int64_t)batch * p.do_batch_stride;
        s.dq_base = (int64_t)batch * p.dq_batch_stride;
        s.k_base = (int64_t)batch * p.k_batch_stride;
        s.v_base = (int64_t)batch * p.v_batch_stride;
        s.dk_base = (int64_t)batch * p.dk_batch_stride;
        s.dv_base = (int64_t)batch * p.dv_batch_stride;
        s.stat_base = (int64_t)batch * p.h * p.seqlen_q;
        s.dsum_base = (int64_t)batch * p.h * p.dsum_row_len;
        s.lse_hs = p.seqlen_q;
        s.dsum_hs = p.dsum_row_len;
    }
    return s;
}

// two packed 16-bit values -> fp32 (plain bit operations for bf16)
template <typename T>
__device__ __forceinline__ void unpack2(uint32_t w, float &lo, float &hi) {
    if constexpr (sizeof(T) == 2 && __is_same(T, __bf16)) {
        lo = __uint_as_float(w << 16);
        hi = __uint_as_float(w & 0xffff0000u);
    } else {
        lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
        hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
    }
}

// ---- D = rowsum(dO * O), fp32.  LPR lanes share one (row, head): 16-byte loads (a wave reads whole rows, coalesced),
//      shuffle reduce.  HBM-bound. ------------------------------------------------------------------------------------
template <typename T, int LPR>
__global__ __launch_bounds__(256) void bwd_dot_kernel(const BParams p) {
    const int64_t rows_total = p.cu_seqlens_q ? (int64_t)p.total_q : (int64_t)p.b * p.seqlen_q;
    const int64_t items = rows_total * p.h;
    const int sub = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (256 / LPR);
    const int64_t n_iter = (items + stride - 1) / stride;  // the same for every lane: the shuffles need all of them
    for (int64_t n = 0; n < n_iter; ++n) {
        const int64_t it = n * stride + (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
        const bool live = it < items;
        const int64_t item = live ? it : 0;
        const int head = (int)(item % p.h);
        const int64_t row = item / p.h;
        int64_t o_off, do_off, d_off;
        if (p.cu_seqlens_q) {
            o_off = row * p.o_row_stride;
            do_off = row * p.do_row_stride;
            d_off = (int64_t)head * p.dsum_row_len + row;
        } else {
            const int64_t bb = row / p.seqlen_q, rr = row % p.seqlen_q;
            o_off = bb * p.o_batch_stride + rr * p.o_row_stride;
            do_off = bb * p.do_batch_stride + rr * p.do_row_stride;
            d_off = (bb * p.h + head) * p.dsum_row_len + rr;
        }
        const T *op = (const T *)p.o + o_off + (int64_t)head * p.o_head_stride;
        const T *gp = (const T *)p.dout + do_off + (int64_t)head * p.do_head_stride;
        float acc = 0.f;
        for (int c = sub * 8; c < p.d_v; c += LPR * 8) {
            const u32x4 a = *(const u32x4 *)(op + c);
            const u32x4 g = *(const u32x4 *)(gp + c);
            const uint32_t aw[4] = {a[0], a[1], a[2], a[3]}, gw[4] = {g[0], g[1], g[2], g[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0, x1, y0, y1;
                unpack2<T>(aw[j], x0, x1);
                unpack2<T>(gw[j], y0, y1);
                acc += x0 * y0 + x1 * y1;
            }
        }
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (live && sub == 0) p.dsum[d_off] = acc;
    }
}

// Elementwise core shared by both kernels: from raw score x and dP to (P, dS) for one (query i, key j).
// `visible` is only looked at when MASK (boundary tiles); rows past the end of q carry LSE = +inf, i.e. P = 0.
// (DROPOUT is a compile-time switch: as a run-time branch it cost the plain backward 2.2 %, same-box A/B)
template <bool SOFTCAP, bool MASK, bool DROPOUT>
__device__ __forceinline__ void bwd_point(const BParams &p, float x, float dp, float lse2, float dsum, float alibi2,
                                          int rel /* i + sk - sq - j */, bool visible, uint32_t rv /* dropout byte of (i, j) */,
                                          float &pv, float &ds) {
    float t = 0.f, sl;
    if constexpr (SOFTCAP) {
        t = fast_tanh(x * p.softcap_pre);
        sl = t * p.scale_log2 - lse2;
    } else {
        sl = x * p.scale_log2 - lse2;
    }
    if (p.alibi) sl -= alibi2 * fabsf((float)rel);
    pv = __builtin_amdgcn_exp2f(sl);
    if constexpr (MASK) pv = visible ? pv : 0.f;
    bool keep = true;
    if constexpr (DROPOUT) {  // O = (keep . P / (1 - p)) V: dP picks up the same factor, D = rowsum(dO . O) still holds
        keep = rv <= (uint32_t)p.drop_thr;
        dp = keep ? dp * p.rp_dropout : 0.f;
    }
    ds = pv * (dp - dsum);
    if constexpr (SOFTCAP) ds *= (1.f - t * t);
    if constexpr (DROPOUT) pv = keep ? pv * p.rp_dropout : 0.f;  // the P that multiplies dO in dV
}

// ------------------------------------------------------------------------------------------------------------------
// dK / dV.  NB = 32-key blocks per wave: every Q / dO / Q^T / dO^T fragment read from LDS feeds NB MFMAs (the
// one-block form moves 1 KiB of LDS per MFMA, which is the LDS bandwidth limit of the CU).
// ------------------------------------------------------------------------------------------------------------------
// PART (head-dim tile 256): 0 = dK and dV in one sweep; 1 = dV only (S, dV: no V operand, no dP); 2 = dK only (S, dP, dK).
// At D = 256 both accumulator sets are 256 registers -- the whole AGPR file -- beside 128 registers of resident K / V
// fragments: the one-sweep form spills its way through every tile (119 TFLOP/s of issued work at s8192, rocprofv3: 18.5 ms
// of a 20.4 ms backward).  Two launches, each with ONE pinned accumulator set, recompute S once more (5 matrix products
// instead of 4) and run without scratch traffic in the tile loop.
template <typename T, int D, int NB, bool SOFTCAP, bool DROPOUT = false, int DEFF = D, int PART = 0>
__global__ __launch_bounds__(256, 1) void bwd_dkdv_kernel(const BParams p) {
    constexpr int NT = 256;
    constexpr bool DO_DK = PART != 1, DO_DV = PART != 2;
    constexpr int WKEYS = 32 * NB;             // keys per wave
    constexpr int BLOCK_K = 4 * WKEYS;         // keys per workgroup
    constexpr int BM = 64;                     // query rows per streamed tile
    // head-dim tile 256 with DEFF = 192 / 160 (head dims 129..192): the k-steps and accumulator blocks of the zero padding are
    // left out of every product; LDS rows, staging and the epilogue keep the tile's 512-byte rows.  (The 128 tile's DEFF = 96
    // only selects the generated loop: its C++ tile path keeps the full tile.)
    constexpr int DE = D == 256 ? DEFF : D;
    constexpr int KSTEPS = DE / 16;
    constexpr int DBLOCKS = DE / 32;
    constexpr int CH_PER_ROW = D / 8;
    constexpr int TILE_BYTES = BM * D * 2;
    constexpr int CHUNKS = BM * CH_PER_ROW;
    constexpr int LD_PER_THREAD = CHUNKS / NT;
    static_assert(CHUNKS % NT == 0, "tile must divide over the workgroup");
    constexpr int O_ROW_BYTES = D * 2 + 16;
    constexpr float LOG2E = 1.4426950408889634f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [Q0 | Q1 | dO0 | dO1 | lse[2][64] | dsum[2][64]]; the epilogue reuses the front as 4 x [32][O_ROW_BYTES]
    float *lse_s = (float *)(smem + 4 * TILE_BYTES);
    float *dsum_s = lse_s + 2 * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    // lane parts of the LDS read addresses; everything else is an immediate or one XOR (lds_off's swizzle only
    // looks at the low 4 row bits, and chunk bits above the swizzle width pass through the XOR): see fa_fwd_kernel_w64.h
    constexpr int ROWB = D * 2;
    const int kbase = lds_off<D>(r, hh);                                                        // ^ 32*ks, + 32*ROWB*blk
    const int vbase = lds_off<D>(4 * hh + (i16 >> 2), 2 * g1 + ((i16 >> 1) & 1)) + 8 * (i16 & 1);  // ^ (64 db + 32 j2)

    int n_block, kv_head, batch;
    if (!decode_block(p, n_block, kv_head, batch, p.h_k)) return;
    // decode_block() deals the blocks of a unit from the last to the first -- the heaviest QUERY block of a causal problem
    // first, right for the dQ sweep.  Key blocks are the other way round: block 0 is seen by every row, the last one by the
    // fewest.  Heaviest first here too, so that the launch ends on light workgroups.
    n_block = p.num_blocks - 1 - n_block;
    const BSeq sq_ = bwd_seq(p, batch);
    const int sq = sq_.sq, sk = sq_.sk;
    const int n0 = n_block * BLOCK_K;
    if (n0 >= sk) return;
    const int shift = sk - sq;

    // query rows that can see any key of this block
    const int last_key = min(sk, n0 + BLOCK_K) - 1;
    int row_lo = 0, row_hi = sq;
    if (p.window_right >= 0) row_lo = max(0, n0 - shift - p.window_right);
    if (p.window_left >= 0) row_hi = min(sq, last_key - shift + p.window_left + 1);
    const int m_min = row_lo / BM;
    const int m_max = row_hi > row_lo ? (row_hi + BM - 1) / BM : m_min;
    const int num_m = m_max - m_min;
    const int total_it = num_m * p.h_ratio;

    const int key_w0 = n0 + wave * WKEYS;

    // ---- K, V fragments of this wave's keys: B operands of S = Q K^T and dP = dO V^T ---------------------------
    const T *kp = (const T *)p.k + sq_.k_base + (int64_t)kv_head * p.k_head_stride;
    const T *vp = (const T *)p.v + sq_.v_base + (int64_t)kv_head * p.v_head_stride;
    // (the dK/dV accumulators live in AGPRs -- asm MFMAs with "+a" --, K/V stay in arch VGPRs: hipcc copies AGPR-pinned
    //  B operands back and forth around every use here)
    u32x4 kf[NB][KSTEPS], vf[NB][KSTEPS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int key = key_w0 + 32 * nb + r;
            const int d0 = ks * 16 + hh * 8;
            u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
            if (key < sk && d0 < p.d) a = *(const u32x4 *)(kp + (int64_t)key * p.k_row_stride + d0);
            if constexpr (DO_DK) {
                if (key < sk && d0 < p.d_v) b = *(const u32x4 *)(vp + (int64_t)key * p.v_row_stride + d0);
            }
            kf[nb][ks] = a;
            vf[nb][ks] = b;  // (PART 1: never read)
        }

    // Accumulators pinned in AGPRs by asm MFMAs -- unless both sets would take ALL 256 AGPRs (D = 256): hipcc then has
    // no AGPR temporaries left and rotates the whole file around the asm statements; there only dV is pinned and dK
    // stays compiler-managed (measured: 5.8 -> 5.0 ms on b2 s4096 h8 d256).
    constexpr bool PIN_ACC = (PART == 0 ? 2 : 1) * NB * DBLOCKS * 16 < 256;  // dK (and dV)
    // (D = 256 with dropout: the extra live state makes hipcc move the pinned tiles around inside MFMA hazard windows;
    //  that rare combination runs on compiler-managed accumulators)
    constexpr bool PIN_DV = PART != 0 || !(D == 256 && DROPOUT);
    f32x16 dk_acc[DO_DK ? NB * DBLOCKS : 1], dv_acc[DO_DV ? NB * DBLOCKS : 1];  // [nb * DBLOCKS + db]
    {
        const u32x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NB * DBLOCKS; ++i) {
            if constexpr (DO_DK) {
                if constexpr (PIN_ACC) Mfma<T>::o_zero(dk_acc[i], z4);
                else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) dk_acc[i][e] = 0.f;
                }
            }
            if constexpr (DO_DV) {
                if constexpr (PIN_DV) Mfma<T>::o_zero(dv_acc[i], z4);
                else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) dv_acc[i][e] = 0.f;
                }
            }
        }
        // (hipcc may spill an accumulator to scratch right here -- it does at D = 256: let the matrix pipe finish first)
        if constexpr (DO_DK && PIN_ACC) drain_acc(dk_acc);
        if constexpr (DO_DV && PIN_DV) drain_acc(dv_acc);
    }

    // ---- Q / dO tile staging: rows clamped into the sequence (clamped rows are masked).  LDS-DMA
    //      (global_load_lds_dwordx4, no staging registers, swizzle applied on the source side: see
    //      fa_fwd_kernel_w64.h), 2 / 4 / 8 pieces per wave at head-dim tiles 64 / 128 / 256. ------------------------------
    constexpr bool DMA = LD_PER_THREAD <= 8;   // (round 3: the 256 tile's 8 pieces per wave go by LDS-DMA too)
    constexpr int NSTAGE = DMA ? 1 : LD_PER_THREAD;
    u32x4 qreg[NSTAGE], greg[NSTAGE];
    float stat_reg = 0.f;
    auto tile_head = [&](int it) { return kv_head * p.h_ratio + it / num_m; };
    auto tile_row0 = [&](int it) { return (m_min + it % num_m) * BM; };
    int dma_row[LD_PER_THREAD], dma_col[LD_PER_THREAD], dma_colv[LD_PER_THREAD];
    uint32_t q_off[LD_PER_THREAD], g_off[LD_PER_THREAD];  // byte offsets of this lane's chunks inside an in-range tile
    const int q_rs = (int)p.q_row_stride, g_rs = (int)p.do_row_stride;  // host guarantees < 2^24
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int slot = wave * (LD_PER_THREAD * 64) + i * 64 + lane;  // 16-byte slot inside the tile image
            const int row = slot / CH_PER_ROW;
            int ch;  // inverse of lds_off<D>: the chunk stored at this slot
            if constexpr (D == 64) ch = (slot % CH_PER_ROW) ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
            else ch = (slot % CH_PER_ROW) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            dma_row[i] = row;
            dma_col[i] = (ch * 8 < p.d) ? ch * 8 : 0;
            // dO has d_v columns (= d unless the FA3 headdim_v differs: wide tile only, no second table elsewhere)
            dma_colv[i] = D == 256 ? ((ch * 8 < p.d_v) ? ch * 8 : 0) : dma_col[i];
            q_off[i] = (uint32_t)(row * q_rs + dma_col[i]) * 2u;
            g_off[i] = (uint32_t)(row * g_rs + dma_colv[i]) * 2u;
        }
    }
    const uint32_t lds_wave = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * (LD_PER_THREAD * 1024);
    auto load_tile = [&](int it, int buf) {
        const int head = tile_head(it), row0 = tile_row0(it);
        const T *qp = (const T *)p.q + sq_.q_base + (int64_t)head * p.q_head_stride;
        const T *gp = (const T *)p.dout + sq_.do_base + (int64_t)head * p.do_head_stride;
        if constexpr (DMA) {
            const T *qt = qp + (int64_t)row0 * p.q_row_stride, *gt = gp + (int64_t)row0 * p.do_row_stride;  // wave-uniform
            if (row0 + BM <= sq) {
                lds_dma<LD_PER_THREAD>(lds_wave + buf * TILE_BYTES, qt, q_off);
                lds_dma<LD_PER_THREAD>(lds_wave + (2 + buf) * TILE_BYTES, gt, g_off);
            } else {
                uint32_t qo[LD_PER_THREAD], go[LD_PER_THREAD];
#pragma unroll
                for (int i = 0; i < LD_PER_THREAD; ++i) {
                    const int rel = min(row0 + dma_row[i], sq - 1) - row0;
                    qo[i] = (uint32_t)(rel * q_rs + dma_col[i]) * 2u;
                    go[i] = (uint32_t)(rel * g_rs + dma_colv[i]) * 2u;
                }
                lds_dma<LD_PER_THREAD>(lds_wave + buf * TILE_BYTES, qt, qo);
                lds_dma<LD_PER_THREAD>(lds_wave + (2 + buf) * TILE_BYTES, gt, go);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                const int c = tid + i * NT;
                const int row = min(row0 + c / CH_PER_ROW, sq - 1);
                const int ch = c % CH_PER_ROW;
                const int col = (ch * 8 < p.d) ? ch * 8 : 0, colg = (ch * 8 < p.d_v) ? ch * 8 : 0;
                qreg[i] = *(const u32x4 *)(qp + (int64_t)row * p.q_row_stride + col);
                greg[i] = *(const u32x4 *)(gp + (int64_t)row * p.do_row_stride + colg);
            }
        }
        if (tid < 2 * BM) {  // threads 0..63: LSE (log2 units), 64..127: D
            const int row = min(row0 + (tid & (BM - 1)), sq - 1);
            if (tid < BM) {
                const float l = p.lse[sq_.stat_base + (int64_t)head * sq_.lse_hs + row];
                stat_reg = l * LOG2E;  // +inf (row without keys) stays +inf: P = exp2(-inf) = 0
            } else {
                stat_reg = p.dsum[sq_.dsum_base + (int64_t)head * sq_.dsum_hs + row];
            }
        }
    };
    auto store_tile = [&](int buf) {
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                const int c = tid + i * NT;
                const int off = lds_off<D>(c / CH_PER_ROW, c % CH_PER_ROW);
                *(u32x4 *)(smem + buf * TILE_BYTES + off) = qreg[i];
                *(u32x4 *)(smem + (2 + buf) * TILE_BYTES + off) = greg[i];
            }
        }
        if (tid < BM) lse_s[buf * BM + tid] = stat_reg;
        else if (tid < 2 * BM) dsum_s[buf * BM + tid - BM] = stat_reg;
    };

    if (total_it > 0) {
        load_tile(0, 0);
        store_tile(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see fa_fwd_kernel.h
    tile_barrier<0>();                   // asm LDS-DMA pieces landed + workgroup barrier

    for (int it = 0; it < total_it; ++it) {
        const int cur = it & 1;
        // The bulk of the sweep at head dim 128: runs of tiles of one head with nothing to mask for this wave's keys go through
        // the generated asm block (fa_bwd_loop_gen.h; same LDS layout and barrier protocol, so the four waves choose
        // independently).  The tile behind a run must belong to the same head (the block prefetches it through the head's
        // buffer descriptors); the run's tiles lie fully inside the sequence.
        if constexpr ((D == 128 || D == 64) && NB == 1 && !SOFTCAP && !DROPOUT && DMA && PART == 0 && !(FA_BWD_ABLATE & 1)) {
            const int row0 = tile_row0(it);
            const int left = num_m - it % num_m;  // tiles of this head from `it` on
            int hi_row = sq - BM;
            if (p.window_left >= 0) hi_row = min(hi_row, key_w0 - (BM - 1) - shift + p.window_left);
            int plain = row0 <= hi_row ? (hi_row - row0) / BM + 1 : 0;
            if (key_w0 + WKEYS > sk) plain = 0;
            if (p.window_right >= 0 && key_w0 + WKEYS - 1 > row0 + shift + p.window_right) plain = 0;
            const int count = min(plain, left - 1);
            const bool addr32 = (int64_t)sq * p.q_row_stride < (1ll << 30) && (int64_t)sq * p.do_row_stride < (1ll << 30);
            if (count >= 2 && !p.alibi && addr32) {
                const int head = tile_head(it);
                auto make_desc = [&](const void *base, uint32_t bytes) {
                    const uint64_t b = (uint64_t)(uintptr_t)base;
                    u32x4 dsc;
                    dsc[0] = (uint32_t)b;
                    dsc[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0: raw buffer
                    dsc[2] = bytes;                          // past it: zeros
                    dsc[3] = 0x00020000u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dsc[i] = __builtin_amdgcn_readfirstlane(dsc[i]);
                    return dsc;
                };
                const T *qp = (const T *)p.q + sq_.q_base + (int64_t)head * p.q_head_stride;
                const T *gp = (const T *)p.dout + sq_.do_base + (int64_t)head * p.do_head_stride;
                const u32x4 qdesc = make_desc(qp, (uint32_t)(((int64_t)(sq - 1) * p.q_row_stride + min(p.d, D)) * 2));
                const u32x4 gdesc = make_desc(gp, (uint32_t)(((int64_t)(sq - 1) * p.do_row_stride + min(p.d, D)) * 2));
                const bool odd = (wave & 1) != 0;  // waves 0, 2 stage the LSE row of the next tile, waves 1, 3 its D row
                const float *sp = odd ? p.dsum + sq_.dsum_base + (int64_t)head * sq_.dsum_hs
                                      : p.lse + sq_.stat_base + (int64_t)head * sq_.lse_hs;
                const u32x4 sdesc = make_desc(sp, (uint32_t)sq * 4u);
                uint32_t qoffb[LD_PER_THREAD], goffb[LD_PER_THREAD];
#pragma unroll
                for (int i = 0; i < LD_PER_THREAD; ++i) {
                    qoffb[i] = q_off[i] - 1024u * i;  // (the instruction offset that steps the LDS target also enters the source)
                    goffb[i] = g_off[i] - 1024u * i;
                }
                const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
                const uint32_t stat0 = lds0 + 4 * TILE_BYTES;
                auto run_loop = [&](auto tag) {   // (DEFF = 96: head dims <= 96 on the 128-wide tiles, the form that skips the zero padding)
                    using Loop = typename decltype(tag)::type;
                    Loop::run(dk_acc, dv_acc, kf[0], vf[0], (uint32_t)kbase, (uint32_t)vbase, qoffb, goffb,
                                   stat0 + 16 * hh, (uint32_t)lane * 4u, stat0 + (odd ? 2 * BM * 4 : 0) + lane * 4, p.scale_log2,
                                   odd ? 1.f : LOG2E, qdesc, gdesc, sdesc, (uint32_t)((row0 + BM) * q_rs * 2),
                                   (uint32_t)((row0 + BM) * g_rs * 2), (uint32_t)((row0 + BM) * 4), (uint32_t)(BM * q_rs * 2),
                                   (uint32_t)(BM * g_rs * 2), lds0, lds_wave, cur, count);
                };
                if constexpr (D == 128) {
                    if constexpr (DEFF == 96) run_loop(LoopTag<BwdLoop96<T>>{});
                    else run_loop(LoopTag<BwdLoop128<T>>{});
                } else {
                    run_loop(LoopTag<BwdLoop64<T>>{});
                }
                it += count - 1;  // tile it + count is in LDS, its barrier passed
                continue;
            }
        }
        const bool has_next = it + 1 < total_it;
        if (has_next) load_tile(it + 1, cur ^ 1);  // (buffer cur^1 was last read before the previous barrier)

        const int head = tile_head(it), row0 = tile_row0(it);
        const float alibi2 = p.alibi ? p.alibi[(int64_t)batch * p.alibi_bs + head] * LOG2E : 0.f;
        const uint32_t seed_mix = DROPOUT ? fa_seed_mix(p.rng_state, batch * p.h + head) : 0u;
        const char *qbuf = smem + cur * TILE_BYTES;
        const char *gbuf = smem + (2 + cur) * TILE_BYTES;

        // wave-level skip: no (row, key) pair of this tile x this wave's keys is visible
        bool skip = key_w0 >= sk;
        if (p.window_right >= 0) skip = skip || (key_w0 > row0 + BM - 1 + shift + p.window_right);
        if (p.window_left >= 0) skip = skip || (key_w0 + WKEYS - 1 < row0 + shift - p.window_left);
        // masks are only evaluated where a boundary crosses this (64 rows x WKEYS keys) block; rows past the end of
        // q are clamped copies: mask those too
        bool need_mask = (key_w0 + WKEYS > sk) || (row0 + BM > sq);
        if (p.window_right >= 0) need_mask = need_mask || (key_w0 + WKEYS - 1 > row0 + shift + p.window_right);
        if (p.window_left >= 0) need_mask = need_mask || (key_w0 < row0 + BM - 1 + shift - p.window_left);

        if (!skip) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                // ---- S = Q K^T and dP = dO V^T for 32 query rows x this wave's keys -----------------------
                f32x16 s[NB], dp[NB];
                // LDS fragments are fetched PF steps ahead of the MFMAs that consume them (the wave is alone on its
                // SIMD: nothing else hides the ~200-cycle LDS latency); sched_barrier pins the order
                constexpr int PF = 2;
                auto row_frag = [&](const char *buf, int ks) {
                    return *(const u32x4 *)(buf + ((kbase ^ (32 * ks)) + rb * (32 * ROWB)));
                };
                u32x4 qa_r[PF + 1], ga_r[PF + 1];
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    qa_r[i] = row_frag(qbuf, i);
                    if constexpr (DO_DK) ga_r[i] = row_frag(gbuf, i);
                }
                if constexpr (!DO_DK) {  // dV only: dP is never formed (bwd_point's dS is dead code there)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) dp[nb][e] = 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    if (ks + PF < KSTEPS) {
                        qa_r[(ks + PF) % (PF + 1)] = row_frag(qbuf, ks + PF);
                        if constexpr (DO_DK) ga_r[(ks + PF) % (PF + 1)] = row_frag(gbuf, ks + PF);
                    }
                    const u32x4 qa = qa_r[ks % (PF + 1)];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (ks == 0) BMfma<T>::s_first_v(s[nb], qa, kf[nb][ks]);
                        else BMfma<T>::s_acc_v(s[nb], qa, kf[nb][ks]);
                        if constexpr (DO_DK) {
                            const u32x4 ga = ga_r[ks % (PF + 1)];
                            if (ks == 0) BMfma<T>::s_first_v(dp[nb], ga, vf[nb][ks]);
                            else BMfma<T>::s_acc_v(dp[nb], ga, vf[nb][ks]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                drain_tiles<NB>(s, dp);
                // ---- P and dS: key on the lane, query row = register ----------------------------------
                auto pointwise = [&](auto mask_c) {
                    constexpr bool MASK = decltype(mask_c)::value;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int rbase = 32 * rb + 8 * g4 + 4 * hh;
                        const float4 l4 = *(const float4 *)(lse_s + cur * BM + rbase);
                        const float4 d4 = *(const float4 *)(dsum_s + cur * BM + rbase);
                        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dsv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const int my_key = key_w0 + 32 * nb + r;
                            uint32_t rblk = 0;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int i = 4 * g4 + e;
                                const int qi = row0 + rbase + e;
                                const int rel = qi + shift - my_key;
                                bool vis = true;
                                if constexpr (MASK) {
                                    vis = (my_key < sk) && (qi < sq);
                                    if (p.window_right >= 0) vis = vis && (rel + p.window_right >= 0);
                                    if (p.window_left >= 0) vis = vis && (rel <= p.window_left);
                                }
                                uint32_t rv = 0;
                                if constexpr (DROPOUT) {  // rows qi (e even) and qi + 1 share one hash: byte 2 (e & 1) + (key & 1)
                                    if ((e & 1) == 0) rblk = fa_rand_block(seed_mix, (uint32_t)qi >> 1, (uint32_t)my_key >> 1) >> (8 * (my_key & 1));
                                    rv = (rblk >> (16 * (e & 1))) & 255u;
                                }
                                float pv, ds;
                                bwd_point<SOFTCAP, MASK, DROPOUT>(p, s[nb][i], dp[nb][i], lv[e], dsv[e], alibi2, rel, vis, rv, pv, ds);
                                s[nb][i] = pv;
                                dp[nb][i] = ds;
                            }
                        }
                    }
                };
                __builtin_amdgcn_sched_barrier(0);
                if (need_mask) pointwise(std::true_type{});
                else pointwise(std::false_type{});
                u32x4 pf[NB][2], dsf[NB][2];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            pf[nb][st][j] = Elem<T>::pack2(s[nb][8 * st + 2 * j], s[nb][8 * st + 2 * j + 1]);
                            dsf[nb][st][j] = Elem<T>::pack2(dp[nb][8 * st + 2 * j], dp[nb][8 * st + 2 * j + 1]);
                        }
                __builtin_amdgcn_sched_barrier(0);
                // ---- dV^T += dO^T P,  dK^T += Q^T dS  (A operands through transposing LDS reads) ----------
                // element j of lane half hh of 16-row step st is query row 16st + 8(j>>2) + 4hh + (j&3)
                auto tr_frag = [&](const char *buf, int t) {  // step t = (db, st)
                    const int db = t >> 1, st = t & 1;
                    u32x4 f;
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int off = (vbase ^ (64 * db + 32 * j2)) + (32 * rb + 16 * st + 8 * j2) * ROWB;
                        const u32x2 a = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4 *)(buf + off)));
                        f[2 * j2] = a[0];
                        f[2 * j2 + 1] = a[1];
                    }
                    return f;
                };
                constexpr int NT2 = 2 * DBLOCKS;
                u32x4 gt_r[PF + 1], qt_r[PF + 1];
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    if constexpr (DO_DV) gt_r[i] = tr_frag(gbuf, i);
                    if constexpr (DO_DK) qt_r[i] = tr_frag(qbuf, i);
                }
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    if (t + PF < NT2) {
                        if constexpr (DO_DV) gt_r[(t + PF) % (PF + 1)] = tr_frag(gbuf, t + PF);
                        if constexpr (DO_DK) qt_r[(t + PF) % (PF + 1)] = tr_frag(qbuf, t + PF);
                    }
                    const int db = t >> 1, st = t & 1;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {  // (s_nop 1 in front: VALU-packed P / dS -> MFMA operand)
                        if constexpr (DO_DV) {
                            const u32x4 gt = gt_r[t % (PF + 1)];
                            if constexpr (PIN_DV) Mfma<T>::o_acc_pad(dv_acc[nb * DBLOCKS + db], gt, pf[nb][st]);
                            else dv_acc[nb * DBLOCKS + db] = Elem<T>::mma(gt, pf[nb][st], dv_acc[nb * DBLOCKS + db]);
                        }
                        if constexpr (DO_DK) {
                            const u32x4 qt = qt_r[t % (PF + 1)];
                            if constexpr (PIN_ACC) Mfma<T>::o_acc_pad(dk_acc[nb * DBLOCKS + db], qt, dsf[nb][st]);
                            else dk_acc[nb * DBLOCKS + db] = Elem<T>::mma(qt, dsf[nb][st], dk_acc[nb * DBLOCKS + db]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        if (has_next) store_tile(cur ^ 1);
        tile_barrier<0>();  // next tile's LDS-DMA landed (vmcnt(0)), LDS writes visible, workgroup barrier
    }

    // ---- epilogue: dK^T / dV^T registers (lane = key, registers = head dim) -> LDS -> coalesced rows ------------
    if constexpr (DO_DK && PIN_ACC) drain_acc(dk_acc);  // asm MFMA results -> VALU readers
    if constexpr (DO_DV && PIN_DV) drain_acc(dv_acc);
    T *dkp = (T *)p.dk + sq_.dk_base + (int64_t)kv_head * p.dk_head_stride;
    T *dvp = (T *)p.dv + sq_.dv_base + (int64_t)kv_head * p.dv_head_stride;
    char *obuf = smem + wave * (32 * O_ROW_BYTES);
#pragma unroll
    for (int which = DO_DK ? 0 : 1; which < (DO_DV ? 2 : 1); ++which) {
        const float f = which == 0 ? p.out_scale : 1.f;
        T *dst = which == 0 ? dkp : dvp;
        const int64_t rs = which == 0 ? p.dk_row_stride : p.dv_row_stride;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x16 &acc = (which == 0 ? dk_acc : dv_acc)[(which == 0 ? DO_DK : DO_DV) ? nb * DBLOCKS + db : 0];
                    u32x2 w;
                    w[0] = Elem<T>::pack2(acc[4 * g4] * f, acc[4 * g4 + 1] * f);
                    w[1] = Elem<T>::pack2(acc[4 * g4 + 2] * f, acc[4 * g4 + 3] * f);
                    *(u32x2 *)(obuf + r * O_ROW_BYTES + (db * 32 + 8 * g4 + 4 * hh) * 2) = w;
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < (32 * CH_PER_ROW) / 64; ++i) {
                const int c = lane + i * 64;
                const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
                const int key = key_w0 + 32 * nb + row;
                if (key < sk && ch * 8 < (which == 0 ? p.d : p.d_v)) {
                    const u32x4 val = *(const u32x4 *)(obuf + row * O_ROW_BYTES + ch * 16);
                    *(u32x4 *)(dst + (int64_t)key * rs + ch * 8) = val;
                }
            }
            __syncthreads();
        }
    }
}

template <int D>
constexpr int smem_bytes_dkdv() {
    constexpr int tiles = 4 * 64 * D * 2 + 4 * 64 * 4;
    constexpr int o = 4 * 32 * (D * 2 + 16);
    return tiles > o ? tiles : o;
}

template <int D> constexpr int dq_nbuf() { return D <= 128 ? 3 : 2; }  // K / V LDS slots of bwd_dq_kernel

// ------------------------------------------------------------------------------------------------------------------
// dQ.  NB = 32-row query blocks per wave (K / V / K^T fragments from LDS feed NB MFMAs each).
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int D, int NB, bool SOFTCAP, bool DROPOUT = false, int DEFF = D>
__global__ __launch_bounds__(256, 1) void bwd_dq_kernel(const BParams p) {
    constexpr int NT = 256;
    constexpr int WROWS = 32 * NB;
    constexpr int BLOCK_M = 4 * WROWS;
    constexpr int DE = D == 256 ? DEFF : D;    // (see bwd_dkdv_kernel)
    constexpr int KSTEPS = DE / 16;
    constexpr int DBLOCKS = DE / 32;
    constexpr int CH_PER_ROW = D / 8;
    constexpr int TILE_BYTES = BLOCK_N * D * 2;
    constexpr int CHUNKS = BLOCK_N * CH_PER_ROW;
    constexpr int LD_PER_THREAD = CHUNKS / NT;
    static_assert(CHUNKS % NT == 0, "tile must divide over the workgroup");
    constexpr int O_ROW_BYTES = D * 2 + 16;
    constexpr float LOG2E = 1.4426950408889634f;

    // [K0 .. K(NBUF-1) | V0 .. V(NBUF-1)]: two slots, three at head dims <= 128 (the generated block reads a tile's K one barrier
    // longer than the C++ path: see tools/gen_bwd_dq_loop.py)
    constexpr int NBUF = dq_nbuf<D>();
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    // lane parts of the LDS read addresses; everything else is an immediate or one XOR (lds_off's swizzle only
    // looks at the low 4 row bits, and chunk bits above the swizzle width pass through the XOR): see fa_fwd_kernel_w64.h
    constexpr int ROWB = D * 2;
    const int kbase = lds_off<D>(r, hh);                                                        // ^ 32*ks, + 32*ROWB*blk
    const int vbase = lds_off<D>(4 * hh + (i16 >> 2), 2 * g1 + ((i16 >> 1) & 1)) + 8 * (i16 & 1);  // ^ (64 db + 32 j2)

    int m_block, head, batch;
    if (!decode_block(p, m_block, head, batch, p.h)) return;
    const int kv_head = head / p.h_ratio;
    const BSeq sq_ = bwd_seq(p, batch);
    const int sq = sq_.sq, sk = sq_.sk;
    const int row_lo = m_block * BLOCK_M;
    if (row_lo >= sq) return;
    const int shift = sk - sq;
    const int row_hi = min(sq, row_lo + BLOCK_M);
    int key_hi = sk, key_lo = 0;
    if (p.window_right >= 0) key_hi = min(sk, row_hi + shift + p.window_right);
    if (p.window_left >= 0) key_lo = max(0, row_lo + shift - p.window_left);
    const int n_min = key_lo / BLOCK_N;
    const int n_max = key_hi > 0 ? (key_hi + BLOCK_N - 1) / BLOCK_N : 0;

    const int wrow = row_lo + wave * WROWS;
    const bool wave_active = wrow < sq;

    const T *qp = (const T *)p.q + sq_.q_base + (int64_t)head * p.q_head_stride;
    const T *gp = (const T *)p.dout + sq_.do_base + (int64_t)head * p.do_head_stride;
    const T *kp = (const T *)p.k + sq_.k_base + (int64_t)kv_head * p.k_head_stride;
    const T *vp = (const T *)p.v + sq_.v_base + (int64_t)kv_head * p.v_head_stride;

    // ---- Q, dO fragments (B operands), LSE and D of this lane's rows ---------------------------------------
    u32x4 qf[NB][KSTEPS], gf[NB][KSTEPS];
    float lse2[NB], dsum[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int my_row = wrow + 32 * nb + r;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d0 = ks * 16 + hh * 8;
            u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
            if (my_row < sq && d0 < p.d) a = *(const u32x4 *)(qp + (int64_t)my_row * p.q_row_stride + d0);
            if (my_row < sq && d0 < p.d_v) b = *(const u32x4 *)(gp + (int64_t)my_row * p.do_row_stride + d0);
            qf[nb][ks] = a;
            gf[nb][ks] = b;
            asm volatile("; pin Q" : "+a"(qf[nb][ks]));   // B operands of every score MFMA: AGPR residents
            asm volatile("; pin dO" : "+a"(gf[nb][ks]));
        }
        lse2[nb] = INFINITY;  // rows past the end of q: P = exp2(-inf) = 0
        dsum[nb] = 0.f;
        if (my_row < sq) {
            lse2[nb] = p.lse[sq_.stat_base + (int64_t)head * sq_.lse_hs + my_row] * LOG2E;
            dsum[nb] = p.dsum[sq_.dsum_base + (int64_t)head * sq_.dsum_hs + my_row];
        }
    }
    const float alibi2 = p.alibi ? p.alibi[(int64_t)batch * p.alibi_bs + head] * LOG2E : 0.f;
    const uint32_t seed_mix = DROPOUT ? fa_seed_mix(p.rng_state, batch * p.h + head) : 0u;

    f32x16 dq_acc[NB * DBLOCKS];  // [nb * DBLOCKS + db], AGPRs
    {
        const u32x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NB * DBLOCKS; ++i) Mfma<T>::o_zero(dq_acc[i], z4);
    }

    // ---- K/V staging: clamped rows, by LDS-DMA (2 / 4 / 8 pieces per wave at head-dim tiles 64 / 128 / 256) ----------
    constexpr bool DMA = LD_PER_THREAD <= 8;   // (round 3: the 256 tile's 8 pieces per wave go by LDS-DMA too)
    constexpr int NSTAGE = DMA ? 1 : LD_PER_THREAD;
    u32x4 kreg[NSTAGE], vreg[NSTAGE];
    int dma_row[LD_PER_THREAD], dma_col[LD_PER_THREAD], dma_colv[LD_PER_THREAD];
    uint32_t k_off[LD_PER_THREAD], v_off[LD_PER_THREAD];
    const int k_rs = (int)p.k_row_stride, v_rs = (int)p.v_row_stride;  // host guarantees < 2^24
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int slot = wave * (LD_PER_THREAD * 64) + i * 64 + lane;
            const int row = slot / CH_PER_ROW;
            int ch;
            if constexpr (D == 64) ch = (slot % CH_PER_ROW) ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
            else ch = (slot % CH_PER_ROW) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            dma_row[i] = row;
            dma_col[i] = (ch * 8 < p.d) ? ch * 8 : 0;
            dma_colv[i] = D == 256 ? ((ch * 8 < p.d_v) ? ch * 8 : 0) : dma_col[i];
            k_off[i] = (uint32_t)(row * k_rs + dma_col[i]) * 2u;
            v_off[i] = (uint32_t)(row * v_rs + dma_colv[i]) * 2u;
        }
    }
    const uint32_t lds_wave = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * (LD_PER_THREAD * 1024);
    auto load_tile = [&](int n, int buf) {
        const int k0 = n * BLOCK_N;
        if constexpr (DMA) {
            const T *kt = kp + (int64_t)k0 * p.k_row_stride, *vt = vp + (int64_t)k0 * p.v_row_stride;  // wave-uniform
            if (k0 + BLOCK_N <= sk) {
                lds_dma<LD_PER_THREAD>(lds_wave + buf * TILE_BYTES, kt, k_off);
                lds_dma<LD_PER_THREAD>(lds_wave + (NBUF + buf) * TILE_BYTES, vt, v_off);
            } else {
                uint32_t ko[LD_PER_THREAD], vo[LD_PER_THREAD];
#pragma unroll
                for (int i = 0; i < LD_PER_THREAD; ++i) {
                    const int rel = min(k0 + dma_row[i], sk - 1) - k0;
                    ko[i] = (uint32_t)(rel * k_rs + dma_col[i]) * 2u;
                    vo[i] = (uint32_t)(rel * v_rs + dma_colv[i]) * 2u;
                }
                lds_dma<LD_PER_THREAD>(lds_wave + buf * TILE_BYTES, kt, ko);
                lds_dma<LD_PER_THREAD>(lds_wave + (NBUF + buf) * TILE_BYTES, vt, vo);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                const int c = tid + i * NT;
                const int row = min(k0 + c / CH_PER_ROW, sk - 1);
                const int ch = c % CH_PER_ROW;
                const int col = (ch * 8 < p.d) ? ch * 8 : 0, colv = (ch * 8 < p.d_v) ? ch * 8 : 0;
                kreg[i] = *(const u32x4 *)(kp + (int64_t)row * p.k_row_stride + col);
                vreg[i] = *(const u32x4 *)(vp + (int64_t)row * p.v_row_stride + colv);
            }
        }
    };
    auto store_tile = [&](int buf) {
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                const int c = tid + i * NT;
                const int off = lds_off<D>(c / CH_PER_ROW, c % CH_PER_ROW);
                *(u32x4 *)(smem + buf * TILE_BYTES + off) = kreg[i];
                *(u32x4 *)(smem + (NBUF + buf) * TILE_BYTES + off) = vreg[i];
            }
        }
    };

    if (n_min < n_max) {
        load_tile(n_min, 0);
        store_tile(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    tile_barrier<0>();

    for (int n = n_min; n < n_max; ++n) {
        const int cur = (n - n_min) % NBUF;
        const int nxt = (cur + 1) % NBUF;
        // The bulk of the sweep at head dim 128: runs of tiles with nothing to mask for this wave's rows go through the
        // generated asm block (fa_bwd_dq_loop_gen.h; same barrier / LDS-DMA protocol, so the four waves choose independently).
        if constexpr ((D == 128 || D == 64) && NB == 2 && !SOFTCAP && !DROPOUT && DMA && !(FA_BWD_ABLATE & 2)) {
            int n_hi = sk / BLOCK_N - 1;  // last tile fully inside the keys
            if (p.window_right >= 0) {
                const int lim = wrow + shift + p.window_right - (BLOCK_N - 1);  // first key of the tile <= lim
                n_hi = min(n_hi, lim >= 0 ? lim / BLOCK_N : -1);
            }
            int n_lo = 0;
            if (p.window_left >= 0) {
                const int lo = wrow + WROWS - 1 + shift - p.window_left;        // first key of the tile >= lo
                n_lo = lo > 0 ? (lo + BLOCK_N - 1) / BLOCK_N : 0;
            }
            const int count = (n >= n_lo && n <= n_hi) ? min(n_hi, n_max - 1) - n + 1 : 0;
            const bool addr32 = (int64_t)sk * p.k_row_stride < (1ll << 30) && (int64_t)sk * p.v_row_stride < (1ll << 30);
            if (count >= 2 && wave_active && !p.alibi && addr32) {
                auto make_desc = [&](const T *base, int64_t rs64) {
                    const uint64_t b = (uint64_t)(uintptr_t)base;
                    u32x4 dsc;
                    dsc[0] = (uint32_t)b;
                    dsc[1] = (uint32_t)(b >> 32) & 0xffffu;                            // stride 0: raw buffer
                    dsc[2] = (uint32_t)(((int64_t)(sk - 1) * rs64 + min(p.d, D)) * 2);  // past the last valid row: zeros
                    dsc[3] = 0x00020000u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dsc[i] = __builtin_amdgcn_readfirstlane(dsc[i]);
                    return dsc;
                };
                const u32x4 kdesc = make_desc(kp, p.k_row_stride), vdesc = make_desc(vp, p.v_row_stride);
                uint32_t koffb[LD_PER_THREAD], voffb[LD_PER_THREAD];
#pragma unroll
                for (int i = 0; i < LD_PER_THREAD; ++i) {
                    koffb[i] = k_off[i] - 1024u * i;  // (the instruction offset that steps the LDS target also enters the source)
                    voffb[i] = v_off[i] - 1024u * i;
                }
                const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
                // (no lambda around the call: all 256 AGPRs are operands of the block, and hipcc shuffles them through scratch when
                //  the arrays reach the asm statement through a closure.  DEFF = 96: head dims <= 96 on the 128-wide tiles)
#define FA_DQ_LOOP_ARGS dq_acc, qf[0], qf[1], gf[0], gf[1], lse2[0], lse2[1], dsum[0], dsum[1], (uint32_t)kbase, \
                                     (uint32_t)vbase, koffb, voffb, p.scale_log2, kdesc, vdesc, \
                                     (uint32_t)((n + 1) * BLOCK_N * k_rs * 2), (uint32_t)((n + 1) * BLOCK_N * v_rs * 2), \
                                     (uint32_t)(BLOCK_N * k_rs * 2), (uint32_t)(BLOCK_N * v_rs * 2), lds0, lds_wave, cur, count
                if constexpr (D == 64) BwdDqLoop64<T>::run(FA_DQ_LOOP_ARGS);
                else if constexpr (DEFF == 96) BwdDqLoop96<T>::run(FA_DQ_LOOP_ARGS);
                else BwdDqLoop128<T>::run(FA_DQ_LOOP_ARGS);
#undef FA_DQ_LOOP_ARGS
                n += count - 1;  // tile n + count is in LDS, its barrier passed
                continue;
            }
        }
        const bool has_next = n + 1 < n_max;
        if (has_next) load_tile(n + 1, nxt);

        const int k0 = n * BLOCK_N;
        bool skip = !wave_active;
        if (p.window_right >= 0) skip = skip || (k0 > wrow + WROWS - 1 + shift + p.window_right);
        if (p.window_left >= 0) skip = skip || (k0 + BLOCK_N - 1 < wrow + shift - p.window_left);
        // masks only where a boundary crosses this (WROWS rows x 64 keys) block (rows past the end of q: LSE = +inf)
        bool need_mask = (k0 + BLOCK_N > sk);
        if (p.window_right >= 0) need_mask = need_mask || (k0 + BLOCK_N - 1 > wrow + shift + p.window_right);
        if (p.window_left >= 0) need_mask = need_mask || (k0 < wrow + WROWS - 1 + shift - p.window_left);

        if (!skip) {
            const char *kbuf = smem + cur * TILE_BYTES;
            const char *vbuf = smem + (NBUF + cur) * TILE_BYTES;
            // two 32-key halves, one after the other (keeps the live score accumulators at 32 NB registers)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                // ---- S^T = K Q^T, dP^T = V dO^T: 32 keys, query on the lane ----------------------------------
                f32x16 s[NB], dp[NB];
                constexpr int PF = 2;  // LDS fragments are fetched PF steps ahead of their MFMAs (see bwd_dkdv_kernel)
                auto row_frag = [&](const char *buf, int ks) {
                    return *(const u32x4 *)(buf + ((kbase ^ (32 * ks)) + kb * (32 * ROWB)));
                };
                u32x4 ka_r[PF + 1], va_r[PF + 1];
#pragma unroll
                for (int i = 0; i < PF; ++i) { ka_r[i] = row_frag(kbuf, i); va_r[i] = row_frag(vbuf, i); }
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    if (ks + PF < KSTEPS) {
                        ka_r[(ks + PF) % (PF + 1)] = row_frag(kbuf, ks + PF);
                        va_r[(ks + PF) % (PF + 1)] = row_frag(vbuf, ks + PF);
                    }
                    const u32x4 ka = ka_r[ks % (PF + 1)], va = va_r[ks % (PF + 1)];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (ks == 0) { Mfma<T>::s_first(s[nb], ka, qf[nb][ks]); Mfma<T>::s_first(dp[nb], va, gf[nb][ks]); }
                        else { Mfma<T>::s_acc(s[nb], ka, qf[nb][ks]); Mfma<T>::s_acc(dp[nb], va, gf[nb][ks]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                drain_tiles<NB>(s, dp);
                auto pointwise = [&](auto mask_c) {
                    constexpr bool MASK = decltype(mask_c)::value;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int my_row = wrow + 32 * nb + r;
                        uint32_t rblk = 0;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int key = k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                            const int rel = my_row + shift - key;
                            bool vis = true;
                            if constexpr (MASK) {
                                vis = key < sk;
                                if (p.window_right >= 0) vis = vis && (rel + p.window_right >= 0);
                                if (p.window_left >= 0) vis = vis && (rel <= p.window_left);
                            }
                            uint32_t rv = 0;
                            if constexpr (DROPOUT) {  // keys key (i even) and key + 1 share one hash
                                if ((i & 1) == 0) rblk = fa_rand_block(seed_mix, (uint32_t)my_row >> 1, (uint32_t)key >> 1) >> (16 * (my_row & 1));
                                rv = (rblk >> (8 * (i & 1))) & 255u;
                            }
                            float pv, ds;
                            bwd_point<SOFTCAP, MASK, DROPOUT>(p, s[nb][i], dp[nb][i], lse2[nb], dsum[nb], alibi2, rel, vis, rv, pv, ds);
                            dp[nb][i] = ds;
                        }
                    }
                };
                if (need_mask) pointwise(std::true_type{});
                else pointwise(std::false_type{});
                u32x4 dsf[NB][2];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            dsf[nb][st][j] = Elem<T>::pack2(dp[nb][8 * st + 2 * j], dp[nb][8 * st + 2 * j + 1]);
                __builtin_amdgcn_sched_barrier(0);
                // ---- dQ^T += K^T dS^T ------------------------------------------------------------------------
                auto tr_frag = [&](int t) {  // step t = (db, st)
                    const int db = t >> 1, st = t & 1;
                    u32x4 f;
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int off = (vbase ^ (64 * db + 32 * j2)) + (32 * kb + 16 * st + 8 * j2) * ROWB;
                        const u32x2 a = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4 *)(kbuf + off)));
                        f[2 * j2] = a[0];
                        f[2 * j2 + 1] = a[1];
                    }
                    return f;
                };
                constexpr int NT2 = 2 * DBLOCKS;
                u32x4 kt_r[PF + 1];
#pragma unroll
                for (int i = 0; i < PF; ++i) kt_r[i] = tr_frag(i);
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    if (t + PF < NT2) kt_r[(t + PF) % (PF + 1)] = tr_frag(t + PF);
                    const u32x4 kt = kt_r[t % (PF + 1)];
                    const int db = t >> 1, st = t & 1;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) Mfma<T>::o_acc_pad(dq_acc[nb * DBLOCKS + db], kt, dsf[nb][st]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        if (has_next) store_tile(nxt);
        tile_barrier<0>();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    // (a wave leaving the generated block reads its last tile's K^T fragments AFTER the last tile barrier: nobody may
    //  reuse the LDS before every wave is here)
    if constexpr (NBUF == 3) __syncthreads();
    drain_acc(dq_acc);  // asm MFMA results -> VALU readers
    T *dqp = (T *)p.dq + sq_.dq_base + (int64_t)head * p.dq_head_stride;
    char *obuf = smem + wave * (32 * O_ROW_BYTES);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x16 &acc = dq_acc[nb * DBLOCKS + db];
                u32x2 w;
                w[0] = Elem<T>::pack2(acc[4 * g4] * p.out_scale, acc[4 * g4 + 1] * p.out_scale);
                w[1] = Elem<T>::pack2(acc[4 * g4 + 2] * p.out_scale, acc[4 * g4 + 3] * p.out_scale);
                *(u32x2 *)(obuf + r * O_ROW_BYTES + (db * 32 + 8 * g4 + 4 * hh) * 2) = w;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < (32 * CH_PER_ROW) / 64; ++i) {
            const int c = lane + i * 64;
            const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
            const int qrow = wrow + 32 * nb + row;
            if (qrow < sq && ch * 8 < p.d) {
                const u32x4 val = *(const u32x4 *)(obuf + row * O_ROW_BYTES + ch * 16);
                *(u32x4 *)(dqp + (int64_t)qrow * p.dq_row_stride + ch * 8) = val;
            }
        }
        __syncthreads();
    }
}

template <int D>
constexpr int smem_bytes_dq() {
    constexpr int kv = 2 * dq_nbuf<D>() * BLOCK_N * D * 2;
    constexpr int o = 4 * 32 * (D * 2 + 16);
    return kv > o ? kv : o;
}

}  // namespace fa
