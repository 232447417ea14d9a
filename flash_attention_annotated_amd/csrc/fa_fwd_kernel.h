// fa_fwd_kernel.h — gfx950 (CDNA4) FlashAttention forward mainloop, hand-written HIP.
//
// Roles re-derived for MI355X (not translated) from the reference's forward:
//   mainloop   hopper/mainloop_fwd_sm90_tma_gmma_ws.hpp:589-891,953-1352,
//              csrc/flash_attn/src/flash_fwd_kernel.h:51-494      -> fwd_kernel() tile loop
//   softmax    hopper/softmax.h:92-168, csrc/flash_attn/src/softmax.h:128-187 -> online_softmax_tile()
//   mask       hopper/mask.h:44-162, csrc/flash_attn/src/mask.h:111-212        -> apply_mask()
//   block      hopper/block.h:14-59, csrc/flash_attn/src/block_info.h:12-45     -> SeqInfo / n-range
//   epilogue   hopper/epilogue_fwd.hpp:213-402                                  -> store_output()
//   scheduler  hopper/tile_scheduler.hpp:36-136,218-363                         -> decode_tile() (XCD-aware)
//
// Design (wave64, MFMA 32x32x16, 160 KiB LDS):
//   * workgroup = NWAVES waves, each wave owns 32 query rows (BLOCK_M = 32*NWAVES), K/V tile = 64 keys.
//   * scores are computed TRANSPOSED, S^T = K.Q^T, so one lane owns one query row: the 32x32 f32
//     accumulator has the query on the lane and the keys in its 16 registers.  Row max / row sum are
//     in-register reductions plus one v_permlane32_swap between the two lane halves.
//   * O is accumulated transposed as well, O^T = V^T.P^T: the S^T accumulator registers, rounded to
//     bf16/fp16, ARE the B operand of that product (no LDS round trip, no shuffles), and the online
//     softmax rescale is one per-lane scalar.
//   * V^T fragments come from a row-major V tile in LDS through ds_read_b64_tr_b16 (hardware transpose).
//   * K and V tiles are double-buffered in LDS; global loads for tile n+1 are issued before the
//     compute of tile n and written to LDS after it (one barrier per tile).
//   * LDS images are XOR-swizzled so that both the ds_read_b128 row reads (K) and the transposed
//     reads (V) are bank-conflict free.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Device-side copy of the fields of fa_fwd_params the kernel needs (kernarg, by value).
struct KParams {
    const void *q, *k, *v;
    void *o;
    float *lse;
    const int32_t *cu_seqlens_q, *cu_seqlens_k, *seqused_q, *seqused_k;
    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    int32_t b, seqlen_q, seqlen_k, h, h_k, d, total_q;
    int32_t h_ratio;       // h / h_k
    int32_t num_m_blocks;  // ceil(seqlen_q / BLOCK_M)
    int32_t num_tiles;     // num_m_blocks * h * b  (work list length)
    int32_t unit_tiles;    // tiles per scheduling unit (see tile_of_wg)
    int32_t whole_slots;   // workgroup slots per XCD dealt in units of unit_tiles; the slots behind them in units of h_ratio
    int32_t grid;          // workgroups launched (a multiple of 8)
    int32_t num_cus;       // compute units of the device (one 256-thread workgroup of the pipelined kernels per CU)
    int32_t window_left, window_right;  // <0 unbounded; causal => right = 0
    float scale;           // softmax_scale (softcap: the softcap value)
    float scale_log2;      // scale * log2(e)
    float softcap_pre;     // softmax_scale / softcap (0 when softcap is off)
    // fp8 inputs: per-(batch, kv head) fp32 dequantisation factors (NULL = 1).  q*k is folded into the softmax scale,
    // v into the final O normalisation (hopper/flash_fwd_kernel_sm90.h:408-415, mainloop_fwd_sm90...hpp:1241-1250).
    const float *q_descale, *k_descale, *v_descale;
    int32_t qd_bs, qd_hs, kd_bs, kd_hs, vd_bs, vd_hs;  // element strides (batch, head)
    const float *alibi;    // ALiBi slopes (h) or (b, h), NULL = off
    int32_t alibi_bs;      // batch stride of alibi (0 for the (h) form)
    const int32_t *kv_batch_idx;  // KV-cache decode: cache entry of each batch row (NULL = identity), dense only
    const int32_t *block_table;   // paged KV: page of key row j of batch i = block_table[i * bt_bs + j / page_size]
    int32_t bt_bs, page_size;     // (only the fwd_kernel shape reads paged caches; any page size, see load_tile)
    // split-KV: workgroup id = split * grid + (id inside one split's grid); split s handles the s-th part of the key
    // blocks of its tile and writes a normalised partial O / LSE at o + s * o_split_stride, lse + s * lse_split_stride
    int32_t num_splits;
    int64_t o_split_stride, lse_split_stride;
    const int32_t *leftpad_k;  // rows of padding in front of each sequence's keys (NULL = none)
    // dropout: keep iff fa_rand8(...) <= drop_thr (255 = off); rp_dropout = 1 / (1 - p); s_dmask: optional uint8 randvals
    const uint64_t *rng_state;
    uint8_t *s_dmask;
    int32_t drop_thr;
    float rp_dropout;
    // ABI v12 (fwd_kernel only; the host keeps the pipelined kernels away from both): head dim of V / O (FA3 headdim_v,
    // hopper/flash_api.cpp:764,782-792; = d everywhere else) and attention_chunk (hopper/mask.h:116-119; 0 = off)
    int32_t dv;
    int32_t chunk;
};

// floor(x / c) * c for c > 0 and any sign of x: first key of the attention chunk of diagonal position x (flash::round_down on
// the reference's FastDivmod, hopper/mask.h:117; the oracle subtracts a Python remainder, hopper/test_util.py:220)
__device__ __forceinline__ int chunk_floor(int x, int c) { return (x >= 0 ? x / c : -((-x + c - 1) / c)) * c; }

// Counter-based random bytes for the dropout decision: ONE 32-bit avalanche (lowbias32) per 2 x 2 block of the score
// matrix -- rows 2a, 2a+1 x keys 2b, 2b+1 of (batch*h + head) -- gives the four 8-bit values of the block (byte index
// 2 (row & 1) + (key & 1)).  Every kernel owns pairs of neighbouring elements (two keys of a row in the forward and dQ,
// two rows of a key in dK/dV), so the hash is evaluated once per two elements.  Forward and backward regenerate the
// same values.
__device__ __forceinline__ uint32_t fa_rand_block(uint32_t seed_mix, uint32_t row2, uint32_t key2) {
    uint32_t x = seed_mix ^ (row2 * 0x9E3779B1u) ^ (key2 * 0x85EBCA77u + 0x165667B1u);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t fa_rand8(uint32_t seed_mix, uint32_t row, uint32_t key) {
    return (fa_rand_block(seed_mix, row >> 1, key >> 1) >> (8 * (((row & 1) << 1) | (key & 1)))) & 255u;
}
__device__ __forceinline__ uint32_t fa_seed_mix(const uint64_t *rng_state, int bh) {
    const uint64_t seed = rng_state[0], off = rng_state[1];
    uint32_t x = (uint32_t)seed ^ (uint32_t)(seed >> 32) * 0x27D4EB2Fu ^ (uint32_t)off * 0xC2B2AE3Du ^ (uint32_t)(off >> 32);
    x ^= (uint32_t)bh * 0x2545F491u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}

// key-block range of split `split` out of p.num_splits (empty ranges are fine: O = 0, LSE = +inf, weight 0 in the merge)
__device__ __forceinline__ void split_range(const KParams &p, int split, int &n_min, int &n_max) {
    if (p.num_splits <= 1) return;
    const int per = (n_max - n_min + p.num_splits - 1) / p.num_splits;
    const int lo = n_min + split * per;
    n_max = min(n_max, lo + per);
    n_min = min(lo, n_max);
}

// per-workgroup effective scales for (batch, kv_head)
struct Scales {
    float scale, scale_log2, softcap_pre, v_descale;
};
// ALiBi slope of (batch, head) in units of the raw score the kernels carry (bias = -slope * |i + sk - sq - j| is
// defined on the scaled score: divide by the factor the softmax multiplies scores with).  0 = off.
__device__ __forceinline__ float load_alibi(const KParams &p, const Scales &sc, int batch, int head) {
    if (!p.alibi) return 0.f;
    return p.alibi[(int64_t)batch * p.alibi_bs + head] / sc.scale;
}
__device__ __forceinline__ Scales load_scales(const KParams &p, int batch, int kv_head) {
    float qk = 1.f, vd = 1.f;
    if (p.q_descale) qk *= p.q_descale[batch * p.qd_bs + kv_head * p.qd_hs];
    if (p.k_descale) qk *= p.k_descale[batch * p.kd_bs + kv_head * p.kd_hs];
    if (p.v_descale) vd = p.v_descale[batch * p.vd_bs + kv_head * p.vd_hs];
    Scales s;
    if (p.softcap_pre != 0.f) {  // softcap: tanh((q.k * qk) * scale / cap) * cap -> only the pre-factor sees qk
        s.scale = p.scale; s.scale_log2 = p.scale_log2; s.softcap_pre = p.softcap_pre * qk;
    } else {
        s.scale = p.scale * qk; s.scale_log2 = p.scale_log2 * qk; s.softcap_pre = 0.f;
    }
    s.v_descale = vd;
    return s;
}

template <typename T> struct Elem;
template <> struct Elem<__bf16> {
    static __device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    // accumulate into an AGPR-resident tile: the "+a" constraint pins the accumulator to the AGPR half of the
    // register file (hipcc otherwise shuttles it through arch VGPRs around every VALU use)
    static __device__ __forceinline__ void mma_agpr(f32x16 &c, u32x4 a, u32x4 b) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    // score step with the Q fragment (B operand) living in an AGPR: c (VGPRs) += a . q.  s_nop 1: c / a may have been
    // written by VALU / LDS-return just before; the caller drains the pipe before VALU reads c (hipcc pads nothing here)
    static __device__ __forceinline__ void mma_qa(f32x16 &c, u32x4 a, u32x4 &q) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %1, %0" : "+v"(c), "+a"(q) : "v"(a));
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round-to-nearest-even
        return __builtin_bit_cast(uint32_t, v);
    }
};
template <> struct Elem<_Float16> {
    static __device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_agpr(f32x16 &c, u32x4 a, u32x4 b) {
        asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    // score step with the Q fragment (B operand) living in an AGPR: c (VGPRs) += a . q.  s_nop 1: c / a may have been
    // written by VALU / LDS-return just before; the caller drains the pipe before VALU reads c (hipcc pads nothing here)
    static __device__ __forceinline__ void mma_qa(f32x16 &c, u32x4 a, u32x4 &q) {
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(c), "+a"(q) : "v"(a));
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        f16x2 v = {(_Float16)lo, (_Float16)hi};  // round-to-nearest-even
        return __builtin_bit_cast(uint32_t, v);
    }
};

constexpr int BLOCK_N = 64;  // keys per K/V tile

// ---- tile scheduler (role of hopper/tile_scheduler.hpp:36-136, 218-363) --------------------------------------------
// Work list, in order: (batch, kv head, m_block descending = heaviest causal blocks first, q head of the GQA group).
// It is cut into UNITS of `unit_tiles` consecutive tiles -- normally all tiles of one (batch, kv head), i.e. everything
// that streams the same K/V -- and the units are dealt round-robin to the 8 XCDs: workgroup ids are dealt round-robin
// over the XCDs by the dispatcher (wg % 8 labels the workgroups that share an L2), so XCD x runs units x, x+8, ...
// one after the other.  Sharing workgroups hit one L2 (HBM traffic ~ algorithmic bytes), and XCDs get the same number of
// units of every batch entry, which balances ragged batches (a contiguous split gave whole sequences to single XCDs).
// Placement is a speed matter only.  Returns false for padding workgroups.
// Workgroup -> tile.  Workgroup ids go round-robin over the 8 XCDs (id & 7); an XCD's slots (id >> 3) first walk its share of
// the (batch, kv head) units -- whole heads, so that an XCD's L2 holds one head's K/V at a time -- as far as those deal evenly
// (whole_slots), then the remaining heads are dealt by m_block (units of the GQA group's h_ratio tiles), which keeps every
// XCD equally loaded for any head count.
__device__ __forceinline__ int tile_of_wg(const KParams &p, int wg) {
    const int xcd = wg & 7, slot = wg >> 3;
    if (slot < p.whole_slots) return ((slot / p.unit_tiles) * 8 + xcd) * p.unit_tiles + slot % p.unit_tiles;
    const int s2 = slot - p.whole_slots;
    return p.whole_slots * 8 + ((s2 / p.h_ratio) * 8 + xcd) * p.h_ratio + s2 % p.h_ratio;
}
__device__ __forceinline__ bool decode_tile(const KParams &p, int &m_block, int &head, int &batch, int &split) {
    split = p.num_splits > 1 ? blockIdx.x / p.grid : 0;  // p.grid is a multiple of 8: the XCD of a tile does not depend on the split
    const int wg = p.num_splits > 1 ? blockIdx.x % p.grid : blockIdx.x;
    const int tile = tile_of_wg(p, wg);
    if (tile >= p.num_tiles) return false;
    const int per_kvh = p.h_ratio * p.num_m_blocks;
    const int bk = tile / per_kvh, r = tile % per_kvh;
    batch = bk / p.h_k;
    m_block = p.num_m_blocks - 1 - r / p.h_ratio;
    head = (bk % p.h_k) * p.h_ratio + r % p.h_ratio;
    return true;
}

// Byte offset of 16-byte chunk `ch` of row `row` inside a [rows][D] 16-bit LDS tile.
// The XOR keeps (a) ds_read_b128 of 16 lanes reading the same chunk of 16 rows distinct mod 16 and
// (b) ds_read_b64_tr_b16 of 4 consecutive rows x 64 bytes on distinct banks.
template <int D>
__device__ __forceinline__ int lds_off(int row, int ch) {
    if constexpr (D == 64) {
        const int g = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
        return row * 128 + 16 * (ch ^ g);
    } else {
        const int g = ((row & 3) << 2) | ((row >> 2) & 3);
        return row * (D * 2) + 16 * (ch ^ g);
    }
}

__device__ __forceinline__ float half_swap_max(float x) {
    // max over the two lane halves (lane l <-> l^32); one v_permlane32_swap
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); role of cutlass::fast_tanh (hopper/utils.h:635-641)
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // exp(2x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

__device__ __forceinline__ float max3(float a, float b, float c) {
    // one v_max3_f32; the plain fmaxf chain gets a canonicalising v_max per MFMA output from hipcc
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// DROPOUT is a compile-time switch: as a run-time branch its bookkeeping sat in every instantiation and cost the
// decode shape (D = 128, 4 waves, 256-register budget) a third of its speed in extra spills.
// DEFF: head dims actually contracted / produced (<= D, multiple of 32): the k-steps and O blocks of the zero padding are
// skipped while the LDS images keep the rows of the D tile (D = 256 with DEFF = 192: head dims 129..192, hopper/tile_size.h).
// EXTRA: the instantiations that know attention_chunk (p.chunk) and a V / O head dim of its own (p.dv) -- FA3-only arguments,
// ABI v12.  Compiled out of the others: as run-time branches they cost the decode shape a spill inside its tile loop.
template <typename T, int D, int NWAVES, bool SOFTCAP, bool DROPOUT = false, int DEFF = D, bool EXTRA = false>
__global__ __launch_bounds__(NWAVES * 64, (D <= 128 ? 2 : 1)) void fwd_kernel(const KParams p) {
    constexpr int NT = NWAVES * 64;
    constexpr int BLOCK_M = NWAVES * 32;
    constexpr int KSTEPS = DEFF / 16;          // k-steps of the QK^T product
    constexpr int DBLOCKS = DEFF / 32;         // 32-wide blocks of the head dim (O^T row blocks)
    constexpr int CH_PER_ROW = D / 8;          // 16-byte chunks per row
    constexpr int TILE_BYTES = BLOCK_N * D * 2;
    constexpr int CHUNKS = BLOCK_N * CH_PER_ROW;
    constexpr int LD_PER_THREAD = CHUNKS / NT;
    static_assert(CHUNKS % NT == 0, "tile must divide over the workgroup");
    constexpr int O_ROW_BYTES = D * 2 + 16;    // padded epilogue row (16-byte aligned)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K0 | K1 | V0 | V1]; the epilogue reuses the whole region as NWAVES x [32][O_ROW_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int hh = lane >> 5;

    int m_block, head, batch, split;
    if (!decode_tile(p, m_block, head, batch, split)) return;  // whole workgroup (padding)
    const int kv_head = head / p.h_ratio;

    // ---- sequence bookkeeping (BlockInfo / SeqlenInfo role) --------------------------------------
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
    if (row_lo >= sq) return;  // whole workgroup: nothing to do (varlen / padded grid)
    const Scales sc = load_scales(p, batch, kv_head);
    const float alibi = load_alibi(p, sc, batch, head);
    // dropout bookkeeping
    uint32_t seed_mix = 0u;
    if constexpr (DROPOUT) seed_mix = fa_seed_mix(p.rng_state, batch * p.h + head);
    int64_t dmask_base = 0;  // s_dmask: dense (b, h, sq, sk), varlen (h, total_q, max_seqlen_k)
    const int64_t dmask_rs = p.seqlen_k;
    if (DROPOUT && p.s_dmask) {
        if (p.cu_seqlens_q) dmask_base = ((int64_t)head * p.total_q + p.cu_seqlens_q[batch]) * p.seqlen_k;
        else dmask_base = ((int64_t)batch * p.h + head) * p.seqlen_q * (int64_t)p.seqlen_k;
    }

    o_base += split * p.o_split_stride;      // split-KV: partial results of split s (0 when off)
    lse_base += split * p.lse_split_stride;
    if (p.leftpad_k) {  // left-padded keys: skip the padding rows, the valid length shrinks by as much
        const int lp = p.leftpad_k[batch];
        sk = max(sk - lp, 0);
        k_base += (int64_t)lp * p.k_row_stride;
        v_base += (int64_t)lp * p.v_row_stride;
    }
    if (p.block_table) k_base = v_base = 0;  // paged: the page supplies the batch offset
    const int32_t *pages = p.block_table ? p.block_table + (int64_t)batch * p.bt_bs : nullptr;
    const T *qp = (const T *)p.q + q_base + (int64_t)head * p.q_head_stride;
    const T *kp = (const T *)p.k + k_base + (int64_t)kv_head * p.k_head_stride;
    const T *vp = (const T *)p.v + v_base + (int64_t)kv_head * p.v_head_stride;
    T *op = (T *)p.o + o_base + (int64_t)head * p.o_head_stride;

    // ---- key range of this row block (BlockMN::get_n_block_min_max role) ------------------------
    const int shift = sk - sq;  // bottom-right aligned masks
    const int row_hi = min(sq, row_lo + BLOCK_M);
    int key_hi = sk, key_lo = 0;
    if (p.window_right >= 0) key_hi = min(sk, row_hi + shift + p.window_right);
    if (p.window_left >= 0) key_lo = max(0, row_lo + shift - p.window_left);
    if (EXTRA && p.chunk > 0) {  // attention_chunk: the block's rows see nothing outside [chunk start of its first row, chunk end of its last)
        key_lo = max(key_lo, chunk_floor(row_lo + shift, p.chunk));
        key_hi = min(key_hi, chunk_floor(row_hi - 1 + shift, p.chunk) + p.chunk);
    }
    int n_min = key_lo / BLOCK_N;
    int n_max = key_hi > 0 ? (key_hi + BLOCK_N - 1) / BLOCK_N : 0;
    split_range(p, split, n_min, n_max);

    const int wrow = row_lo + wave * 32;          // first row of this wave
    const int my_row = wrow + r;                  // the query row this lane owns
    const bool wave_active = wrow < sq;

    // ---- Q fragments: B operand of S^T = K.Q^T; lane (r,hh) holds Q[row r][16ks + 8hh .. +8] ------
    // (branch-free loads, zeroed by selects afterwards: predicated loads are waited for one by one -- see fa_fwd_kernel_w64.h)
    u32x4 qf[KSTEPS];
    {
        const T *qr = qp + (int64_t)min(my_row, sq - 1) * p.q_row_stride;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d0 = ks * 16 + hh * 8;
            qf[ks] = *(const u32x4 *)(qr + (d0 < p.d ? d0 : 0));
        }
        const u32x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = (ks * 16 + hh * 8 < p.d && my_row < sq) ? qf[ks] : z4;
    }
    // D = 256: 16 Q fragments = 64 registers that only MFMAs read.  Pinned into the AGPR half of the register file they
    // stop competing with the softmax for arch VGPRs (the compiler-placed version spilled ~10 registers to scratch and
    // reloaded them inside every tile: one wave per SIMD here, nothing hides those round trips).
    constexpr bool Q_IN_AGPR = (D == 256);
    if constexpr (Q_IN_AGPR) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("; pin Q" : "+a"(qf[ks]));
    }

    // ---- accumulators --------------------------------------------------------------------------
    f32x16 o_acc[DBLOCKS];
#pragma unroll
    for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o_acc[db][i] = 0.f;
    float m_run = -INFINITY;  // running row max (unscaled scores), same in both lane halves
    float l_run = 0.f;        // running row sum, PARTIAL per lane half (combined in the epilogue)

    // ---- K/V staging -------------------------------------------------------------------------
    // Branch-free loads: a 64-bit wave-uniform tile base plus a 32-bit per-lane offset.  Rows past the
    // end of the sequence are CLAMPED to the last valid row and head-dim chunks past d to chunk 0
    // instead of being zero-filled: the duplicated keys are masked to -inf (P = 0) and a duplicated
    // K chunk meets a zero Q chunk, so finite duplicates contribute exactly 0.  No predication means
    // no exec-masked branches and no conservative vmcnt waits in front of the MFMAs.
    u32x4 kreg[LD_PER_THREAD], vreg[LD_PER_THREAD];
    // thread t stages chunk (t % CH_PER_ROW) of rows t / CH_PER_ROW + i * ROWS_PER_PASS: two lane constants, the rest are
    // immediates (as per-i arrays they were 2 LD_PER_THREAD live registers that hipcc spilled and reloaded every tile)
    static_assert(NT % CH_PER_ROW == 0, "a pass of the workgroup covers whole rows");
    constexpr int ROWS_PER_PASS = NT / CH_PER_ROW;
    const int ld_row0 = tid / CH_PER_ROW;
    const int ld_col0 = ((tid % CH_PER_ROW) * 8 < p.d) ? (tid % CH_PER_ROW) * 8 : 0;
    // V has p.dv columns (= p.d unless the FA3 headdim_v differs): chunks past them duplicate chunk 0 and only ever reach
    // O columns >= dv, which the epilogue does not store
    const int ld_col0v = !EXTRA ? ld_col0 : ((tid % CH_PER_ROW) * 8 < p.dv) ? (tid % CH_PER_ROW) * 8 : 0;
    const int k_rs = (int)p.k_row_stride, v_rs = (int)p.v_row_stride;  // host guarantees 64 * stride < 2^31
    auto load_tile = [&](int n) {
        const int k0 = n * BLOCK_N;
        const T *kt = kp + (int64_t)k0 * p.k_row_stride;  // scalar
        const T *vt = vp + (int64_t)k0 * p.v_row_stride;
        const int last = sk - 1 - k0;                     // >= 0 for every tile in [n_min, n_max)
        if (pages) {
            if (p.page_size % BLOCK_N == 0) {  // a 64-key tile lies inside one page
                const int page = pages[k0 / p.page_size], in_page = k0 % p.page_size;
                kt = kp + (int64_t)page * p.k_batch_stride + (int64_t)in_page * p.k_row_stride;
                vt = vp + (int64_t)page * p.v_batch_stride + (int64_t)in_page * p.v_row_stride;
            } else {  // any other page size (FA3: "page_block_size can be arbitrary"): the page is looked up per row
#pragma unroll
                for (int i = 0; i < LD_PER_THREAD; ++i) {
                    const int row = k0 + min(ld_row0 + i * ROWS_PER_PASS, last);
                    const int pi = row / p.page_size;
                    const int64_t page = pages[pi];
                    const int in_page = row - pi * p.page_size;
                    kreg[i] = *(const u32x4 *)(kp + page * p.k_batch_stride + (int64_t)in_page * p.k_row_stride + ld_col0);
                    vreg[i] = *(const u32x4 *)(vp + page * p.v_batch_stride + (int64_t)in_page * p.v_row_stride + ld_col0v);
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int row = min(ld_row0 + i * ROWS_PER_PASS, last);
            kreg[i] = *(const u32x4 *)(kt + (uint32_t)(row * k_rs + ld_col0));
            vreg[i] = *(const u32x4 *)(vt + (uint32_t)(row * v_rs + ld_col0v));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int c = tid + i * NT;
            const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
            const int off = lds_off<D>(row, ch);
            *(u32x4 *)(smem + buf * TILE_BYTES + off) = kreg[i];
            *(u32x4 *)(smem + (2 + buf) * TILE_BYTES + off) = vreg[i];
        }
    };

    // lane-constant pieces of the LDS read addresses; everything else is an immediate or one XOR (a separate VGPR per
    // swizzled address costs ~100 registers at D = 256): see fa_fwd_kernel_w64.h
    const int i16 = lane & 15;           // lane inside its 16-lane group
    const int g1 = (lane >> 4) & 1;      // which 16-column half of a 32-wide d block
    const int kbase = lds_off<D>(r, hh);
    const int vbase = lds_off<D>(4 * hh + (i16 >> 2), 2 * g1 + ((i16 >> 1) & 1)) + 8 * (i16 & 1);

    if (n_min < n_max) {
        load_tile(n_min);
        store_tile(0);
    }
    // Retire every prologue load (Q included) here: otherwise hipcc's waitcnt pass keeps Q "pending" on the
    // loop back-edge and drains vmcnt(0) in front of the first QK^T MFMA of EVERY tile, right after the
    // next tile's loads were issued.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    for (int n = n_min; n < n_max; ++n) {
        const int cur = (n - n_min) & 1;
        const bool has_next = (n + 1 < n_max);
        if (has_next) load_tile(n + 1);

        const int k0 = n * BLOCK_N;
        // wave-uniform tile classification (3-phase split of hopper/block.h:106-135, decided per tile)
        bool skip = !wave_active;
        bool need_mask = (k0 + BLOCK_N > sk);
        if (p.window_right >= 0) {
            skip = skip || (k0 > wrow + 31 + shift + p.window_right);
            need_mask = need_mask || (k0 + BLOCK_N - 1 > wrow + shift + p.window_right);
        }
        if (p.window_left >= 0) {
            skip = skip || (k0 + BLOCK_N - 1 < wrow + shift - p.window_left);
            need_mask = need_mask || (k0 < wrow + 31 + shift - p.window_left);
        }
        if (EXTRA && p.chunk > 0) {  // the wave's rows see [w_lo, w_hi) at most; all of them see [w_in_lo, w_in_hi) (may be empty)
            const int w_lo = chunk_floor(wrow + shift, p.chunk), w_in_lo = chunk_floor(wrow + 31 + shift, p.chunk);
            skip = skip || (k0 + BLOCK_N - 1 < w_lo) || (k0 >= w_in_lo + p.chunk);
            need_mask = need_mask || (k0 < w_in_lo) || (k0 + BLOCK_N > w_lo + p.chunk);
        }

        if (!skip) {
            const char *kbuf = smem + cur * TILE_BYTES;
            const char *vbuf = smem + (2 + cur) * TILE_BYTES;

            // ---- S^T = K.Q^T : two 32-key blocks ------------------------------------------------
            f32x16 s[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[0][i] = 0.f; s[1][i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int off = kbase ^ (32 * ks);  // = lds_off<D>(r, 2 ks + hh): the swizzle XORs chunk bits 0-3 only
                const u32x4 kf0 = *(const u32x4 *)(kbuf + off);
                const u32x4 kf1 = *(const u32x4 *)(kbuf + off + 32 * D * 2);
                if constexpr (Q_IN_AGPR) {
                    Elem<T>::mma_qa(s[0], kf0, qf[ks]);
                    Elem<T>::mma_qa(s[1], kf1, qf[ks]);
                } else {
                    s[0] = Elem<T>::mma(kf0, qf[ks], s[0]);
                    s[1] = Elem<T>::mma(kf1, qf[ks], s[1]);
                }
            }
            if constexpr (Q_IN_AGPR) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s[0]), "+v"(s[1]));  // asm MFMA results -> VALU

            if constexpr (SOFTCAP) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[kb][i] = fast_tanh(s[kb][i] * sc.softcap_pre);
            }

            if (p.alibi) {  // wave-uniform; bias on the (soft-capped) score, before masking: src/mask.h:156-186
                const int rel0 = my_row + shift - k0 - 4 * hh;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rel = rel0 - (kb * 32 + (i & 3) + 8 * (i >> 2));
                        s[kb][i] -= alibi * fabsf((float)rel);
                    }
            }

            // ---- mask (boundary tiles only) -----------------------------------------------------
            if (need_mask) {
                int lim_hi = sk;  // exclusive
                int lim_lo = 0;   // inclusive
                if (p.window_right >= 0) lim_hi = min(sk, my_row + shift + p.window_right + 1);
                if (p.window_left >= 0) lim_lo = max(0, my_row + shift - p.window_left);
                if (EXTRA && p.chunk > 0) {  // hopper/mask.h:116-119: the window intersected with the row's chunk
                    const int c_lo = chunk_floor(my_row + shift, p.chunk);
                    lim_lo = max(lim_lo, c_lo);
                    lim_hi = min(lim_hi, c_lo + p.chunk);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        if (key >= lim_hi || key < lim_lo) s[kb][i] = -INFINITY;
                    }
            }

            // ---- online softmax (per lane = per query row) ----------------------------------------
            float mx = max3(s[0][0], s[1][0], m_run);
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = max3(mx, s[0][i], s[1][i]);
            const float m_new = half_swap_max(mx);  // >= m_run (m_run is identical in both halves)
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far
            const float mc = m_use * sc.scale_log2;
            if (__any(m_new > m_run)) {  // wave-uniform; bit-identical to always rescaling
                const float alpha = __builtin_amdgcn_exp2f(m_run * sc.scale_log2 - mc);
                l_run *= alpha;
#pragma unroll
                for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o_acc[db][i] *= alpha;
            }
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pv = __builtin_amdgcn_exp2f(s[kb][i] * sc.scale_log2 - mc);
                    s[kb][i] = pv;
                    psum += pv;
                }
            l_run += psum;  // (the normaliser sums the probabilities BEFORE dropout)

            if constexpr (DROPOUT) {  // dropout of the probabilities that feed the PV product
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {  // elements i, i+1 = keys key, key+1 (key even): one hash
                        const int key = k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        const uint32_t x = fa_rand_block(seed_mix, (uint32_t)my_row >> 1, (uint32_t)key >> 1) >> (16 * (my_row & 1));
                        const uint32_t rv0 = x & 255u, rv1 = (x >> 8) & 255u;
                        if (rv0 > (uint32_t)p.drop_thr) s[kb][i] = 0.f;
                        if (rv1 > (uint32_t)p.drop_thr) s[kb][i + 1] = 0.f;
                        if (p.s_dmask && my_row < sq) {
                            uint8_t *dm = p.s_dmask + dmask_base + (int64_t)my_row * dmask_rs + key;
                            if (key < sk) dm[0] = (uint8_t)rv0;
                            if (key + 1 < sk) dm[1] = (uint8_t)rv1;
                        }
                    }
            }

            // ---- P^T fragments: accumulator registers ARE the B operand of O^T += V^T.P^T ------------
            u32x4 pf[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int kb = st >> 1, b8 = (st & 1) * 8;
#pragma unroll
                for (int j = 0; j < 4; ++j) pf[st][j] = Elem<T>::pack2(s[kb][b8 + 2 * j], s[kb][b8 + 2 * j + 1]);
            }

            // ---- O^T += V^T.P^T ----------------------------------------------------------------
            // element j of lane half hh of k-step st is key 16st + 8(j>>2) + 4hh + (j&3)
#pragma unroll
            for (int db = 0; db < DBLOCKS; ++db) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    u32x4 vf;
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        // = lds_off<D>(16 st + 8 j2 + 4 hh + (i16 >> 2), 4 db + 2 g1 + ((i16 >> 1) & 1)) + 8 (i16 & 1)
                        const int off = (vbase ^ (64 * db + 32 * j2)) + (16 * st + 8 * j2) * (D * 2);
                        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4 *)(vbuf + off));
                        const u32x2 t2 = __builtin_bit_cast(u32x2, t);
                        vf[2 * j2] = t2[0];
                        vf[2 * j2 + 1] = t2[1];
                    }
                    o_acc[db] = Elem<T>::mma(vf, pf[st], o_acc[db]);
                }
            }
        }

        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise, LSE, O^T regs -> LDS -> coalesced rows -------------------------------
    // (the loop's last barrier has retired every K/V read, so the region can be reused)
    const float l_tot = half_swap_sum(l_run);
    const bool empty = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = (empty ? 1.f : 1.f / l_tot) * sc.v_descale * p.rp_dropout;
    if (wave_active) {
        if (hh == 0 && my_row < sq) {
            // csrc/flash_attn/src/softmax.h:178-180: +inf for rows with no valid key
            p.lse[lse_base + my_row] = empty ? INFINITY : m_run * sc.scale + __logf(l_tot);
        }
        if (p.num_splits > 1) {
            // split-KV partial: fp32 in the caller's workspace (role of out_accum, csrc/flash_attn/flash_api.cpp:297-318), straight
            // from the accumulators -- 4 consecutive head dims = one 16-byte store per lane; the merge launch rounds once
            float *opf = (float *)p.o + o_base + (int64_t)head * p.o_head_stride + (int64_t)my_row * p.o_row_stride;
            if (my_row < sq) {
#pragma unroll
                for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int col = db * 32 + 8 * g4 + 4 * hh;
                        if (col < (EXTRA ? p.dv : p.d))
                            *(float4 *)(opf + col) = make_float4(o_acc[db][4 * g4] * inv, o_acc[db][4 * g4 + 1] * inv,
                                                                 o_acc[db][4 * g4 + 2] * inv, o_acc[db][4 * g4 + 3] * inv);
                    }
            }
        } else {
        char *obuf = smem + wave * (32 * O_ROW_BYTES);
#pragma unroll
        for (int db = 0; db < DBLOCKS; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w[0] = Elem<T>::pack2(o_acc[db][4 * g4] * inv, o_acc[db][4 * g4 + 1] * inv);
                w[1] = Elem<T>::pack2(o_acc[db][4 * g4 + 2] * inv, o_acc[db][4 * g4 + 3] * inv);
                *(u32x2 *)(obuf + r * O_ROW_BYTES + (db * 32 + 8 * g4 + 4 * hh) * 2) = w;
            }
        }
    }
    if (p.num_splits > 1) return;  // (uniform over the launch: no wave is left waiting at the barrier below)
    __syncthreads();
    if (wave_active) {
        const char *obuf = smem + wave * (32 * O_ROW_BYTES);
        // (LDS reads outside the predicate: all of them are issued before the first store; inside it each read is
        //  waited for in its own exec-masked block -- 16 serial LDS round trips at D = 128)
        constexpr int NCH = (32 * CH_PER_ROW) / 64;
        u32x4 val[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * 64;
            val[i] = *(const u32x4 *)(obuf + (c / CH_PER_ROW) * O_ROW_BYTES + (c % CH_PER_ROW) * 16);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + i * 64;
            const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
            if (wrow + row < sq && ch * 8 < (EXTRA ? p.dv : p.d)) *(u32x4 *)(op + (int64_t)(wrow + row) * p.o_row_stride + ch * 8) = val[i];
        }
    }
}

template <int D, int NWAVES>
constexpr int smem_bytes() {
    constexpr int kv = 4 * BLOCK_N * D * 2;
    constexpr int o = NWAVES * 32 * (D * 2 + 16);
    return kv > o ? kv : o;
}

}  // namespace fa
