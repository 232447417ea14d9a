// fa_fwd_api.hip — C-ABI entry points declared in include/fa_fwd.h.
//
// Host-side role of mha_fwd / mha_varlen_fwd (csrc/flash_attn/flash_api.cpp:350-512, 514-755),
// set_params_fprop (:26-159) and run_mha_fwd (:243-255): validate, fill the kernel params,
// pick the instantiation, launch on the caller's stream.  No allocation, no synchronisation.
#include "fa_fwd.h"
#include "fa_fwd_kernel.h"
#include "fa_fwd_kernel_w64.h"
#include "fa_fwd_kernel_fp8.h"
#include "fa_fwd_kernel_d256.h"

#include <algorithm>
#include <atomic>
#include <cmath>

namespace {

std::atomic<int> g_default_variant{0};
std::atomic<int> g_persist_mode{0};  // test hook: 0 = the library's choice, -1 = never the persistent kernel, 1 = whenever it can run

// ---- fp8 e4m3 -> bf16 expansion (exact), strided source -> contiguous (rows, heads, d) destination ------------------
// One thread = 8 elements (8-byte load, 16-byte store).  HBM-bound elementwise pass.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ void expand_fp8_kernel(const uint8_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t rows,
                                  int rows_per_batch, int heads, int d, int64_t batch_stride, int64_t row_stride,
                                  int64_t head_stride) {
    const int chunks = d >> 3;
    const int64_t total = rows * heads * chunks;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int64_t rh = i / chunks;
        const int hd = (int)(rh % heads);
        const int64_t row = rh / heads;
        const int64_t b = rows_per_batch > 0 ? row / rows_per_batch : 0;
        const int64_t r = rows_per_batch > 0 ? row % rows_per_batch : row;
        const uint2 v = *reinterpret_cast<const uint2 *>(src + b * batch_stride + r * row_stride + hd * head_stride + c * 8);
        uint32_t out[4];
        const uint32_t w[2] = {v.x, v.y};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[k], false);
            const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[k], true);
            const bf16x2_t a = {(__bf16)lo[0], (__bf16)lo[1]};
            const bf16x2_t bq = {(__bf16)hi[0], (__bf16)hi[1]};
            out[2 * k] = __builtin_bit_cast(uint32_t, a);
            out[2 * k + 1] = __builtin_bit_cast(uint32_t, bq);
        }
        *reinterpret_cast<uint4 *>(dst + i * 4) = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// ---- rotary embedding on 16-byte chunks (8 elements), fp32 math, round to the storage type -------------------------
template <typename T>
__device__ __forceinline__ void unpack8(const uint4 &w, float (&x)[8]) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (sizeof(T) == 2 && __is_same(T, __bf16)) {
            x[2 * j] = __uint_as_float(u[j] << 16);
            x[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u);
        } else {
            x[2 * j] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u[j] & 0xffffu));
            x[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u[j] >> 16));
        }
    }
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&x)[8]) {
    uint32_t u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = fa::Elem<T>::pack2(x[2 * j], x[2 * j + 1]);
    return make_uint4(u[0], u[1], u[2], u[3]);
}
// One work item = one pair of chunks.  Interleaved: chunk c holds pairs (2j, 2j+1): rotated in place with cos/sin
// [4c, 4c+4).  Otherwise chunk c (< rotary_dim/16) pairs with chunk c + rotary_dim/16, cos/sin [8c, 8c+8).
// Chunks past rotary_dim are copied.  `item` enumerates ceil(d/16) slots per (row, head): slot j covers chunk j (first
// half of the rotary part), its partner, and -- past the rotary part -- the plain chunks 2 per slot.
template <typename T>
__device__ __forceinline__ void rotary_slot(const T *src, T *dst, int d, int rd, bool interleaved, int slot,
                                            const T *cos_row, const T *sin_row) {
    const int chunks = d >> 3, rchunks = rd >> 3;
    int c0, c1;  // the two chunks of this slot (c1 = -1: none)
    if (slot < rchunks / 2 + (rchunks & 1)) {
        if (interleaved) { c0 = 2 * slot; c1 = 2 * slot + 1 < rchunks ? 2 * slot + 1 : -1; }
        else { c0 = slot; c1 = slot + rchunks / 2; }
    } else {  // plain chunks behind the rotary part, two per slot
        const int k = slot - (rchunks / 2 + (rchunks & 1));
        c0 = rchunks + 2 * k;
        c1 = c0 + 1 < chunks ? c0 + 1 : -1;
        if (c0 >= chunks) return;
        *reinterpret_cast<uint4 *>(dst + c0 * 8) = *reinterpret_cast<const uint4 *>(src + c0 * 8);
        if (c1 >= 0) *reinterpret_cast<uint4 *>(dst + c1 * 8) = *reinterpret_cast<const uint4 *>(src + c1 * 8);
        return;
    }
    if (interleaved) {
        const int cs[2] = {c0, c1};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = cs[q];
            if (c < 0) continue;
            float x[8], cv[8], sv[8], y[8];
            unpack8<T>(*reinterpret_cast<const uint4 *>(src + c * 8), x);
            // 4 cos/sin values for this chunk: load the aligned 8 and pick the half
            unpack8<T>(*reinterpret_cast<const uint4 *>(cos_row + (c >> 1) * 8), cv);
            unpack8<T>(*reinterpret_cast<const uint4 *>(sin_row + (c >> 1) * 8), sv);
            const int h = (c & 1) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float co = cv[h + j], si = sv[h + j];
                y[2 * j] = x[2 * j] * co - x[2 * j + 1] * si;
                y[2 * j + 1] = x[2 * j + 1] * co + x[2 * j] * si;
            }
            *reinterpret_cast<uint4 *>(dst + c * 8) = pack8<T>(y);
        }
    } else {
        float x1[8], x2[8], cv[8], sv[8], y1[8], y2[8];
        unpack8<T>(*reinterpret_cast<const uint4 *>(src + c0 * 8), x1);
        unpack8<T>(*reinterpret_cast<const uint4 *>(src + c1 * 8), x2);
        unpack8<T>(*reinterpret_cast<const uint4 *>(cos_row + c0 * 8), cv);
        unpack8<T>(*reinterpret_cast<const uint4 *>(sin_row + c0 * 8), sv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            y1[j] = x1[j] * cv[j] - x2[j] * sv[j];
            y2[j] = x2[j] * cv[j] + x1[j] * sv[j];
        }
        *reinterpret_cast<uint4 *>(dst + c0 * 8) = pack8<T>(y1);
        *reinterpret_cast<uint4 *>(dst + c1 * 8) = pack8<T>(y2);
    }
}

template <typename T>
__global__ void rotary_kernel(const fa_rotary_params p) {
    const int slots = (p.d / 8 + 1) / 2;
    const int64_t total = (int64_t)p.b * p.s * p.h * slots;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % slots);
        int64_t t = i / slots;
        const int hd = (int)(t % p.h);
        t /= p.h;
        const int row = (int)(t % p.s);
        const int b = (int)(t / p.s);
        const int pos = p.seqlen_offsets[b] + (p.per_row_positions ? row : 0);
        const T *src = (const T *)p.src + b * p.src_batch_stride + row * p.src_row_stride + hd * p.src_head_stride;
        T *dst = (T *)p.dst + b * p.dst_batch_stride + row * p.dst_row_stride + hd * p.dst_head_stride;
        const T *cr = (const T *)p.rotary_cos + (int64_t)pos * (p.rotary_dim / 2);
        const T *sr = (const T *)p.rotary_sin + (int64_t)pos * (p.rotary_dim / 2);
        rotary_slot<T>(src, dst, p.d, p.rotary_dim, p.rotary_interleaved != 0, slot, cr, sr);
    }
}

// ---- KV-cache append: k_new/v_new rows -> cache rows [cache_seqlens[b], +seqlen_new), keys optionally rotated.
// One thread = one slot of two 16-byte chunks of K and the same chunks of V.  HBM-bound elementwise pass.
template <typename T>
__global__ void kvcache_append_kernel(const fa_kvcache_append_params p) {
    const int slots = (p.d / 8 + 1) / 2, chunks = p.d >> 3;
    const int64_t total = (int64_t)p.b * p.seqlen_new * p.h_k * slots;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i % slots);
        int64_t t = i / slots;
        const int hd = (int)(t % p.h_k);
        t /= p.h_k;
        const int row = (int)(t % p.seqlen_new);
        const int b = (int)(t / p.seqlen_new);
        int dst_row = p.cache_seqlens[b] + row;
        if (dst_row < 0 || dst_row >= p.seqlen_cache) continue;
        const int pos = p.rotary_seqlens ? p.rotary_seqlens[b] + row : dst_row;   // (FA3 seqlens_rotary; default: the cache row)
        int cb = p.cache_batch_idx ? p.cache_batch_idx[b] : b;
        if (p.block_table) {
            cb = p.block_table[b * p.block_table_batch_stride + dst_row / p.page_block_size];
            dst_row %= p.page_block_size;
        }
        const T *ks = (const T *)p.k_new + b * p.knew_batch_stride + row * p.knew_row_stride + hd * p.knew_head_stride;
        const T *vs = (const T *)p.v_new + b * p.vnew_batch_stride + row * p.vnew_row_stride + hd * p.vnew_head_stride;
        T *kd = (T *)p.k_cache + cb * p.kcache_batch_stride + dst_row * p.kcache_row_stride + hd * p.kcache_head_stride;
        T *vd = (T *)p.v_cache + cb * p.vcache_batch_stride + dst_row * p.vcache_row_stride + hd * p.vcache_head_stride;
        const int rd = p.rotary_cos ? p.rotary_dim : 0;
        const T *cr = (const T *)p.rotary_cos + (int64_t)pos * (rd / 2);
        const T *sr = (const T *)p.rotary_sin + (int64_t)pos * (rd / 2);
        rotary_slot<T>(ks, kd, p.d, rd, p.rotary_interleaved != 0, slot, cr, sr);
        // V: the same enumeration without a rotary part (slot -> chunks 2 slot, 2 slot + 1)
        const int c0 = 2 * slot, c1 = 2 * slot + 1;
        if (c0 < chunks) *reinterpret_cast<uint4 *>(vd + c0 * 8) = *reinterpret_cast<const uint4 *>(vs + c0 * 8);
        if (c1 < chunks) *reinterpret_cast<uint4 *>(vd + c1 * 8) = *reinterpret_cast<const uint4 *>(vs + c1 * 8);
    }
}

// ---- sign-encoded S_dmask (FA_FLAG_SDMASK_SIGNED, include/fa_fwd.h): what the reference's CUDA forward returns for
// return_softmax under dropout (csrc/flash_attn/src/flash_fwd_kernel.h:350-360, 412-422; src/dropout.h:26-33), restated as a
// pass of its own -- a testing aid there and here, not on the hot path.  One workgroup = 8 query rows of one (batch, head);
// per row: all scores (fp32 dot products, scaled, + ALiBi, masked) into LDS, the row maximum of every key block of `block_n`
// keys, their suffix maxima (the running maximum of a sweep from the last block to the first), then
// exp(score - suffix max of its block), negated where fa_rand8 (the forward's hash) drops the element.
template <typename T>
__global__ __launch_bounds__(256) void sdmask_kernel(const fa::KParams p, T *out, int rows_r, int cols_r, int block_n) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    constexpr int ROWS = 8;
    const int nrb = (p.seqlen_q + ROWS - 1) / ROWS;
    const int rb = blockIdx.x % nrb, head = (blockIdx.x / nrb) % p.h, batch = blockIdx.x / (nrb * p.h);
    const int kv_head = head / p.h_ratio;
    int sq, sk;
    int64_t q_base, k_base;
    if (p.cu_seqlens_q) {
        const int q0 = p.cu_seqlens_q[batch], k0 = p.cu_seqlens_k[batch];
        sq = p.seqused_q ? p.seqused_q[batch] : p.cu_seqlens_q[batch + 1] - q0;
        sk = p.seqused_k ? p.seqused_k[batch] : p.cu_seqlens_k[batch + 1] - k0;
        q_base = (int64_t)q0 * p.q_row_stride;
        k_base = (int64_t)k0 * p.k_row_stride;
    } else {
        sq = p.seqused_q ? p.seqused_q[batch] : p.seqlen_q;
        sk = p.seqused_k ? p.seqused_k[batch] : p.seqlen_k;
        q_base = (int64_t)batch * p.q_batch_stride;
        k_base = (int64_t)batch * p.k_batch_stride;
    }
    if (rb * ROWS >= sq || sk <= 0) return;
    const T *qp = (const T *)p.q + q_base + (int64_t)head * p.q_head_stride;
    const T *kp = (const T *)p.k + k_base + (int64_t)kv_head * p.k_head_stride;
    T *op = out + (((int64_t)batch * p.h + head) * rows_r) * cols_r;
    const fa::Scales sc = fa::load_scales(p, batch, kv_head);
    const float alibi = p.alibi ? p.alibi[(int64_t)batch * p.alibi_bs + head] : 0.f;
    const uint32_t seed_mix = fa::fa_seed_mix(p.rng_state, batch * p.h + head);
    const int shift = sk - sq;
    float *qs = (float *)smem_;                // [ROWS][d]
    float *scs = qs + ROWS * p.d;              // [sk]
    float *bm = scs + ((sk + 3) & ~3);         // [nblk]
    const int nblk = (sk + block_n - 1) / block_n;
    for (int i = threadIdx.x; i < ROWS * p.d; i += 256) {
        const int r = rb * ROWS + i / p.d;
        qs[i] = r < sq ? (float)qp[(int64_t)r * p.q_row_stride + i % p.d] : 0.f;
    }
    __syncthreads();
    for (int rr = 0; rr < ROWS; ++rr) {
        const int row = rb * ROWS + rr;
        if (row >= sq) break;   // (uniform)
        const float *qr = qs + rr * p.d;
        for (int key = threadIdx.x; key < sk; key += 256) {
            const T *kr = kp + (int64_t)key * p.k_row_stride;
            float acc = 0.f;
            for (int c = 0; c < p.d; c += 8) {
                const uint4 w = *reinterpret_cast<const uint4 *>(kr + c);
                float x[8];
                unpack8<T>(w, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += qr[c + e] * x[e];
            }
            float sv = acc * sc.scale - alibi * fabsf((float)(row + shift - key));
            bool vis = true;
            if (p.window_right >= 0) vis = vis && key <= row + shift + p.window_right;
            if (p.window_left >= 0) vis = vis && key >= row + shift - p.window_left;
            scs[key] = vis ? sv : -INFINITY;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < nblk; j += 256) {
            float m = -INFINITY;
            for (int key = j * block_n; key < min(sk, (j + 1) * block_n); ++key) m = fmaxf(m, scs[key]);
            bm[j] = m;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int j = nblk - 2; j >= 0; --j) bm[j] = fmaxf(bm[j], bm[j + 1]);
        __syncthreads();
        for (int key = threadIdx.x; key < sk; key += 256) {
            const float m = bm[key / block_n];
            const float pv = (scs[key] == -INFINITY || m == -INFINITY) ? 0.f : __expf(scs[key] - m);
            const bool keep = fa::fa_rand8(seed_mix, (uint32_t)row, (uint32_t)key) <= (uint32_t)p.drop_thr;
            op[(int64_t)row * cols_r + key] = (T)(keep ? pv : -pv);
        }
        __syncthreads();
    }
}

int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

struct Fp8Plan {
    int64_t rows_q, rows_k, q_bytes, kv_bytes, total;
};
Fp8Plan fp8_plan(const fa_fwd_params *p) {
    Fp8Plan pl;
    pl.rows_q = p->cu_seqlens_q ? p->total_q : (int64_t)p->b * p->seqlen_q;
    pl.rows_k = p->cu_seqlens_q ? p->total_k : (int64_t)p->b * p->seqlen_k;
    pl.q_bytes = align256(pl.rows_q * p->h * p->d * 2);
    pl.kv_bytes = align256(pl.rows_k * p->h_k * p->d * 2);
    pl.total = pl.q_bytes + 2 * pl.kv_bytes;
    return pl;
}

int head_dim_tile(int d);
int block_m_of(int variant, int d);

// fp8 inputs run natively (fa_fwd_kernel_fp8.h: e4m3 operands straight into the block-scaled MFMA, no expansion pass, no
// workspace) for the shape BASELINE config 5 names -- head dim 128, dense or varlen, full / causal / right-window masks.  The
// rest of the fp8 surface (other head dims, softcap, left windows) and an explicit kernel_variant take the exact
// e4m3 -> bf16 expansion in front of the 16-bit kernels.
// ABI v12: attention_chunk and a V head dim of its own exist in fwd_kernel only (the compiler-scheduled shape)
inline bool own_dv(const fa_fwd_params *p) { return p->d_v > 0 && p->d_v != p->d; }
inline bool generic_only(const fa_fwd_params *p) { return p->attention_chunk > 0 || own_dv(p); }
inline int wide_dim(const fa_fwd_params *p) { return own_dv(p) ? std::max(p->d, std::min(p->d_v, 256)) : p->d; }  // what the LDS tile has to hold

bool fp8_native(const fa_fwd_params *p) {
    if (p->dtype != FA_DTYPE_FP8_E4M3 || p->d != 128 || generic_only(p)) return false;
    const int variant = p->kernel_variant ? p->kernel_variant : g_default_variant.load();
    if (variant != 0) return false;
    if (p->softcap > 0.f || p->alibi_slopes || p->block_table || p->kv_batch_idx || p->leftpad_k || p->p_dropout > 0.f) return false;
    if (p->window_size_left >= 0 && ((p->flags & FA_FLAG_FA3_WINDOW) || p->window_size_left < p->seqlen_k)) return false;
    const int64_t strides[] = {p->q_row_stride, p->q_head_stride, p->k_row_stride, p->k_head_stride, p->v_row_stride,
                               p->v_head_stride, p->cu_seqlens_q ? 0 : p->q_batch_stride, p->cu_seqlens_q ? 0 : p->k_batch_stride,
                               p->cu_seqlens_q ? 0 : p->v_batch_stride};
    for (int64_t st : strides)
        if (st % 16 != 0) return false;
    const void *ptrs[] = {p->q, p->k, p->v};
    for (const void *ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return false;
    // The kernel addresses one (batch, kv head)'s K and V through raw buffer descriptors: 32-bit byte offsets (num_records,
    // soffset = tile * row_stride) and int row strides.  A K or V of 2 GiB or more per batch entry has no safe path in the
    // native kernel (its generic tile uses the same descriptors): those shapes take the expansion path, whose 16-bit kernels
    // address with 64-bit tile bases.
    const int64_t rows_k = p->cu_seqlens_q ? p->total_k : p->seqlen_k;
    if (p->k_row_stride >= (int64_t(1) << 31) || p->v_row_stride >= (int64_t(1) << 31)) return false;
    if (rows_k * p->k_row_stride >= (int64_t(1) << 31) || rows_k * p->v_row_stride >= (int64_t(1) << 31)) return false;
    return true;
}

// ---- split-KV plan (role of num_splits_heuristic / set_params_splitkv, csrc/flash_attn/flash_api.cpp:257-329) ---------
// Only dense (non-varlen) 16-bit problems split.  Heuristic (num_splits == 0): split when the tiles leave most of the
// 256 CUs idle, so that tiles x splits reaches ~2 workgroups per CU, with at least 4 key blocks (256 keys) per split.
// Kernel shape for a problem.  0 = the library's choice: the 256-row software-pipelined kernel, except
//  * paged caches -> the 64-key-aligned 8-wave shape (variant 1);
//  * short dense query blocks (seqlen_q <= 128: decode steps, short prefill chunks) -> 4 waves x 32 rows (variant 2):
//    a 256-row tile would leave 2-3 of its 4 waves without rows, and this shape fits two workgroups per CU
//    (measured on decode b8 hq32/hkv8 cache 8192: 149 -> 59 us, b32: 255 -> 219 us).
bool fp8_native(const fa_fwd_params *p);
int effective_variant(const fa_fwd_params *p) {
    if (fp8_native(p)) return 0;  // one shape: 4 waves x 64 rows
    int variant = p->kernel_variant ? p->kernel_variant : g_default_variant.load();
    if (variant < 0 || variant > 3) variant = 0;
    if (p->p_dropout > 0.f) return 1;  // dropout lives in the compiler-scheduled shape only
    if (generic_only(p)) {
        // fwd_kernel, EXTRA instantiations: 8 waves x 32 rows (4 x 32 at head-dim tile 256) -- except a V head dim of its own
        // on the wide tile, which dispatch_variant<T, 256> hands to the generated-loop kernel when the features are plain
        if (p->attention_chunk == 0 && head_dim_tile(wide_dim(p)) == 256 && variant == 0) return 0;
        return 1;
    }
    if (variant == 0 || variant == 3) {
        if (p->block_table) variant = 1;
        else if (!p->cu_seqlens_q && p->seqlen_q <= 128) variant = 2;
        else if (variant == 0) {
            // Short key ranges: a 256-row workgroup sweeps only a handful of 64-key tiles, so its fixed cost (~13 us) and,
            // under a causal mask, the idle time of the waves whose rows end early (all four meet at every tile barrier)
            // outweigh the pipelined loop.  Measured on the reference's benchmark grid (16k tokens, tools/fwd_grid.py):
            //   d64  causal s512/1024/2048: 178/255/379 -> 235/332/431 TFLOP/s (4 waves x 32 rows); non-causal s512 421 -> 459
            //   d128 causal s512/1024:      230/345     -> 293/364      TFLOP/s (8 waves x 32 rows)
            const bool causal_like = p->is_causal || (p->window_size_right == 0 && p->window_size_left < 0);
            const int tile = head_dim_tile(p->d);
            // (round 2: at head-dim tile 128 the generated loop applies the causal mask itself -- the diagonal tiles no longer
            //  run the generic half-step -- and the 256-row kernel wins from seqlen 512 on: causal s512 / s1024 308 / 379
            //  (8 waves x 32 rows) -> 319 / 485 TFLOP/s.  Head-dim tile 64 has its generated loop too (FastLoop64), but with twice
            //  the VALU per MFMA the 4-wave x 32-row shape still wins on short key ranges: causal s512 / 1024 / 2048 278 / 394 / 518
            //  against 227 / 331 / 497, non-causal s512 490 against 443 (profiles/r2_fwd_grid.txt): the rule stays)
            if (tile == 64 && ((causal_like && p->seqlen_k <= 2048) || p->seqlen_k <= 512)) variant = 2;
        }
    }
    return variant;
}

struct SplitPlan {
    int splits;
    int64_t o_bytes, lse_bytes, total;  // partial O (fp32, (splits, b, sq, h, d)) and LSE (fp32, (splits, b, h, sq))
};
SplitPlan split_plan(const fa_fwd_params *p, int variant) {
    SplitPlan sp{1, 0, 0, 0};
    if (p->cu_seqlens_q || p->dtype == FA_DTYPE_FP8_E4M3 || p->seqlen_q <= 0 || p->seqlen_k <= 0) return sp;
    if (p->p_dropout > 0.f) return sp;  // (the reference does not split under dropout either: flash_api.cpp:307)
    if (own_dv(p)) return sp;           // (the partials and the merge are laid out for d columns)
    int n = p->num_splits;
    const int n_blocks = (p->seqlen_k + 63) / 64;
    if (n == 0) {
        const int bm = block_m_of(variant, wide_dim(p));
        const int64_t tiles = (int64_t)((p->seqlen_q + bm - 1) / bm) * p->h * p->b;
        // two workgroups of the 4-wave shape fit a CU: aim at ~4 per CU there, ~2 per CU for the 256-row kernel
        const int64_t cap = (variant == 2) ? 512 : 128, target = (variant == 2) ? 1024 : 512;
        n = 1;
        if (tiles > 0 && tiles <= cap && n_blocks >= 8) {
            n = (int)std::min<int64_t>((target + tiles - 1) / tiles, n_blocks / 4);
            n = std::max(1, std::min(n, 64));
        }
    }
    n = std::max(1, std::min(n, std::min(n_blocks, 128)));
    if (n <= 1) return sp;
    sp.splits = n;
    const int64_t rows = (int64_t)p->b * p->seqlen_q;
    sp.o_bytes = (n * rows * p->h * p->d * 4 + 255) & ~int64_t(255);
    sp.lse_bytes = (n * rows * p->h * 4 + 255) & ~int64_t(255);
    sp.total = sp.o_bytes + sp.lse_bytes;
    return sp;
}

// ---- split-KV merge: out = sum_s w_s O_s / sum_s w_s, w_s = exp(lse_s - max lse); lse = max + log sum w.  One thread =
// one 16-byte chunk of one (batch, row, head); splits with LSE = +inf (no key in their range) carry no weight.
template <typename T>
__global__ void combine_splits_kernel(const float *__restrict__ o_acc, const float *__restrict__ lse_acc, T *__restrict__ out,
                                      float *__restrict__ lse_out, int splits, int b, int sq, int h, int d,
                                      int64_t o_bs, int64_t o_rs, int64_t o_hs) {
    const int chunks = d >> 3;
    const int64_t total = (int64_t)b * sq * h * chunks;
    const int64_t o_split = (int64_t)b * sq * h * d, lse_split = (int64_t)b * h * sq;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        int64_t t = i / chunks;
        const int hd = (int)(t % h);
        t /= h;
        const int row = (int)(t % sq);
        const int bb = (int)(t / sq);
        const int64_t lse_idx = ((int64_t)bb * h + hd) * sq + row;
        float mx = -INFINITY;
        for (int s = 0; s < splits; ++s) {
            const float l = lse_acc[s * lse_split + lse_idx];
            if (l != INFINITY) mx = fmaxf(mx, l);
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float wsum = 0.f;
        if (mx != -INFINITY) {
            for (int s = 0; s < splits; ++s) {
                const float l = lse_acc[s * lse_split + lse_idx];
                if (l == INFINITY) continue;
                const float w = __expf(l - mx);
                wsum += w;
                const float4 *src = reinterpret_cast<const float4 *>(o_acc + s * o_split + (((int64_t)bb * sq + row) * h + hd) * d + c * 8);
                const float4 x0 = src[0], x1 = src[1];
                const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += w * x[j];
            }
            const float inv = 1.f / wsum;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] *= inv;
        }
        *reinterpret_cast<uint4 *>(out + bb * o_bs + row * o_rs + hd * o_hs + c * 8) = pack8<T>(acc);
        if (c == 0) lse_out[lse_idx] = (mx == -INFINITY) ? INFINITY : mx + __logf(wsum);
    }
}

// ---- fa_fwd_combine: the same merge over caller-provided fp32 partials with arbitrary strides.  One thread = 4
// consecutive head-dim elements of one (batch, row, head); the work is one pass over out_partial (HBM-bound), the LSE
// column of a row is re-read by the d/4 threads that share it (L1/L2 hits).
template <typename TO>
__device__ __forceinline__ void store_out(TO *dst, float x) { *dst = (TO)x; }
template <typename TO>
__global__ void combine_partials_kernel(const fa_combine_params p) {
    const int chunks = (p.d + 3) >> 2;
    const int64_t total = (int64_t)p.b * p.seqlen * p.h * chunks;
    const bool vec = (p.d % 4 == 0) && (p.op_split_stride % 4 == 0) && (p.op_batch_stride % 4 == 0) &&
                     (p.op_row_stride % 4 == 0) && (p.op_head_stride % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(p.out_partial) % 16 == 0);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        int64_t t = i / chunks;
        const int hd = (int)(t % p.h);
        t /= p.h;
        const int row = (int)(t % p.seqlen);
        const int bb = (int)(t / p.seqlen);
        const float *lp = p.lse_partial + bb * p.lp_batch_stride + row * p.lp_row_stride + hd * p.lp_head_stride;
        const float *op = p.out_partial + bb * p.op_batch_stride + row * p.op_row_stride + hd * p.op_head_stride + c * 4;
        float mx = -INFINITY;
        for (int s = 0; s < p.num_splits; ++s) mx = fmaxf(mx, lp[s * p.lp_split_stride]);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float wsum = 0.f;
        const int n = min(4, p.d - c * 4);
        if (mx != -INFINITY && mx != INFINITY) {
            for (int s = 0; s < p.num_splits; ++s) {
                const float l = lp[s * p.lp_split_stride];
                if (l == -INFINITY) continue;  // (short-circuit: the partial of an empty split is never read)
                const float w = __expf(l - mx);
                wsum += w;
                const float *src = op + s * p.op_split_stride;
                if (vec) {
                    const float4 x = *reinterpret_cast<const float4 *>(src);
                    acc[0] += w * x.x; acc[1] += w * x.y; acc[2] += w * x.z; acc[3] += w * x.w;
                } else {
                    for (int j = 0; j < n; ++j) acc[j] += w * src[j];
                }
            }
            const float inv = 1.f / wsum;
            for (int j = 0; j < 4; ++j) acc[j] *= inv;
        }
        TO *dst = reinterpret_cast<TO *>(p.out) + bb * p.o_batch_stride + row * p.o_row_stride + hd * p.o_head_stride + c * 4;
        for (int j = 0; j < n; ++j) store_out<TO>(dst + j, acc[j]);
        if (c == 0)
            p.softmax_lse[bb * p.lse_batch_stride + row * p.lse_row_stride + hd * p.lse_head_stride] =
                (mx == -INFINITY || mx == INFINITY) ? mx : mx + __logf(wsum);
    }
}

int head_dim_tile(int d) {
    if (d <= 64) return 64;
    if (d <= 128) return 128;
    return 256;
}

template <typename T, int D, int NWAVES, bool SOFTCAP, bool DROPOUT = false, int DEFF = D, bool EXTRA = false>
int launch(const fa::KParams &kp, hipStream_t stream) {
    constexpr int smem = fa::smem_bytes<D, NWAVES>();
    auto kernel = fa::fwd_kernel<T, D, NWAVES, SOFTCAP, DROPOUT, DEFF, EXTRA>;
    // the > 64 KiB dynamic-LDS opt-in is a per-device attribute of the kernel: one bit per device ordinal
    static std::atomic<uint64_t> attr_set{0};
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
    hipLaunchKernelGGL(kernel, dim3(kp.grid * (kp.num_splits > 1 ? kp.num_splits : 1)), dim3(NWAVES * 64), smem, stream, kp);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

template <typename T, int D, bool SOFTCAP, int DEFF = D, bool PERSIST = false>
int launch_w64(const fa::KParams &kp, hipStream_t stream) {
    constexpr int smem = fa::smem_bytes_w64<D>();
    auto kernel = fa::fwd_kernel_w64<T, D, SOFTCAP, DEFF, PERSIST>;
    // the > 64 KiB dynamic-LDS opt-in is a per-device attribute of the kernel: one bit per device ordinal
    static std::atomic<uint64_t> attr_set{0};
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
    // PERSIST: one workgroup per CU (a multiple of 8: ids go round-robin over the XCDs), each walks its chain of the slot list
    const int wgs = PERSIST ? std::min(kp.grid, kp.num_cus & ~7) : kp.grid * (kp.num_splits > 1 ? kp.num_splits : 1);
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), smem, stream, kp);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

// The persistent form of the 256-row kernel (fa_fwd_kernel_w64.h, PERSIST): plain dense problems at head dims 97..128 whose
// every work item sweeps at least three 64-key tiles and whose K / V tensors one 32-bit raw buffer descriptor can span.
bool persist_ok(const fa::KParams &kp) {
    const int mode = g_persist_mode.load();
    if (mode < 0) return false;
    if (kp.cu_seqlens_q || kp.cu_seqlens_k || kp.seqused_q || kp.seqused_k || kp.leftpad_k || kp.kv_batch_idx || kp.block_table) return false;
    if (kp.alibi || kp.q_descale || kp.k_descale || kp.v_descale || kp.num_splits > 1 || kp.rp_dropout != 1.f) return false;
    if (kp.d <= 96 || kp.d > 128) return false;   // (left windows: the next item's first tile is its own n_min, round 3)
    if (kp.seqlen_k % 64 != 0 || kp.seqlen_k < 192 || kp.seqlen_q > kp.seqlen_k) return false;  // (causal: bottom-right aligned, shift >= 0)
    if ((kp.num_cus & ~7) < 8) return false;
    // the kernel decodes its chain once into 32-bit entries (m_block 12 bits, head 10, batch 10), one lane per round
    if (kp.grid > 64 * (kp.num_cus & ~7) || kp.num_m_blocks > 4096 || kp.h > 1024 || kp.b > 1024) return false;
    auto extent = [&](int64_t bs, int64_t hs, int64_t rs, int lead) {
        return ((int64_t)(kp.b - 1) * bs + (int64_t)(kp.h_k - 1) * hs + (int64_t)(kp.seqlen_k - 1 + lead) * rs + 128) * 2;
    };
    if (kp.k_batch_stride < 0 || kp.k_head_stride < 0 || kp.k_row_stride <= 0 || kp.v_batch_stride < 0 || kp.v_head_stride < 0 || kp.v_row_stride <= 0) return false;
    if (extent(kp.k_batch_stride, kp.k_head_stride, kp.k_row_stride, 32) >= (1ll << 32) - 65536 ||
        extent(kp.v_batch_stride, kp.v_head_stride, kp.v_row_stride, 0) >= (1ll << 32) - 65536) return false;
    if ((int64_t)kp.seqlen_k * kp.k_row_stride >= (1ll << 30) || (int64_t)kp.seqlen_k * kp.v_row_stride >= (1ll << 30)) return false;
    if (mode > 0) return true;
    // left windows: supported (bit-identical to the hand-over kernel, tests/test_persistent_gpu.py) but measured SLOWER there --
    // b4 s4096 window (1024, 0) 538 -> 472, (512, 512) 530 -> 469, s16384 (4096, 0) 825 -> 782 TFLOP/s: a windowed item spends its
    // first tiles on the generic half-step (the left edge), where the cross-item look-ahead stream buys nothing -- not dispatched
    if (kp.window_left >= 0) return false;
    // chains of one item gain nothing; from two items per CU on the persistent form wins or ties on the whole benchmark grid
    // (profiles/r3_persist_sweep.txt: non-causal s512 .. 16k +10 / +5 / +2 / 0 %, causal +21 / +26 / +15 / +11 / +3 / +1 %)
    return kp.grid > (kp.num_cus & ~7);
}

// head dims 129 .. 256, plain features: 4 waves x 32 rows around the generated loop FastLoop256 (fa_fwd_kernel_d256.h)
template <typename T, int DEFF, bool SOFTCAP = false, bool ALIBI = false>
int launch_d256(const fa::KParams &kp, hipStream_t stream) {
    constexpr int smem = fa::smem_bytes_d256();
    auto kernel = fa::fwd_kernel_d256<T, DEFF, SOFTCAP, ALIBI>;
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            (void)hipGetLastError();
            return FA_ERR_LAUNCH;
        }
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3(kp.grid), dim3(256), smem, stream, kp);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

// the 32-row-per-wave kernel (fa_fwd_kernel_d256.h) at head-dim tile W (64 .. 256), plain / softcap / ALiBi form
template <typename T, int W>
int launch_d256_form(const fa::KParams &kp, bool softcap, hipStream_t stream) {
    if (softcap) return launch_d256<T, W, true, false>(kp, stream);
    if (kp.alibi) return launch_d256<T, W, false, true>(kp, stream);
    if constexpr (W >= 160) return launch_d256<T, W>(kp, stream);
    else return FA_ERR_UNSUPPORTED;   // (head dims <= 128 without softcap / ALiBi run the 256-row kernel: never dispatched here)
}
template <typename T>
int launch_d256_wide(const fa::KParams &kp, int w, bool softcap, hipStream_t stream) {
    if (w <= 160) return launch_d256_form<T, 160>(kp, softcap, stream);
    if (w <= 192) return launch_d256_form<T, 192>(kp, softcap, stream);
    return launch_d256_form<T, 256>(kp, softcap, stream);
}

template <typename T, int D>
int dispatch_variant(const fa::KParams &kp, bool softcap, int variant, hipStream_t stream) {
    // variant 0/3: 4 waves x 64 rows, one wave per SIMD, software-pipelined (fa_fwd_kernel_w64.h) -- the default
    // variant 1  : 8 waves x 32 rows (BLOCK_M 256), two waves per SIMD (fa_fwd_kernel.h)
    // variant 2  : 4 waves x 32 rows (BLOCK_M 128)
    // D = 256 does not fit the w64 register budget (O alone would be 256 registers): 4 waves x 32 rows.
    const bool d256_ok = (variant == 0 || variant == 3) && !(softcap && kp.alibi) && !kp.block_table && kp.num_splits <= 1 &&
                         kp.rp_dropout == 1.f && kp.chunk == 0;
    if constexpr (D == 256) {
        // a V head dim of its own on the wide tile (192 / 128, or q/k <= 64 beside v in (128, 256]) with plain features: the
        // generated-loop kernel, head-dim tile by the larger of the two (fa_fwd_kernel_d256.h reads p.dv for V and O)
        if (kp.dv != kp.d && d256_ok) return launch_d256_wide<T>(kp, std::max(kp.d, kp.dv), softcap, stream);
    }
    if (kp.chunk > 0 || kp.dv != kp.d) {
        // attention_chunk / a V head dim of its own (FA3 surface, ABI v12): the EXTRA instantiations of the compiler-scheduled
        // shape, 8 waves x 32 rows (4 at head-dim tile 256); no dropout on that surface (fa_fwd_validate)
        constexpr int NW = D == 256 ? 4 : 8;
        if (softcap) return launch<T, D, NW, true, false, D, true>(kp, stream);
        return launch<T, D, NW, false, false, D, true>(kp, stream);
    }
    if (kp.rp_dropout != 1.f) {  // dropout (p > 0, also when its 8-bit threshold keeps everything): its own instantiation of the compiler-scheduled shape (never with softcap)
        if constexpr (D == 256) return launch<T, D, 4, false, true>(kp, stream);
        else return launch<T, D, 8, false, true>(kp, stream);
    }
    if constexpr (D == 256) {
        // round 3: the plain problems (dense / varlen, causal / windows, GQA, softcap) run the generated-loop kernel; ALiBi,
        // paged caches, split-KV and the short-q / explicit shapes keep the compiler-scheduled one
        // head-dim tiles 160 / 192 / 256 (hopper/tile_size.h:20-45): the zero padding is neither multiplied nor accumulated;
        // softcap or ALiBi: the forms of the generated loop that cap / bias the fresh scores themselves (round 3)
        if (d256_ok) return launch_d256_wide<T>(kp, kp.d, softcap, stream);
        // (a DEFF = 192 instantiation -- 12 + 12 instead of 16 + 16 MFMAs per 32-key block -- was measured at exactly the
        //  per-workgroup time of the 256 one, tools/hdim_bench.py: this shape is bound by its register-staged K/V rows, which
        //  stay 512 B wide, not by the matrix pipe; head dims 129..192 therefore keep the 256 instantiation)
        if (softcap) return launch<T, D, 4, true>(kp, stream);
        return launch<T, D, 4, false>(kp, stream);
    } else {
        // variant 4 (internal, fa_fwd): softcap or ALiBi at head dims <= 128 with otherwise plain features on the head-dim-256
        // kernel's shape (one 32-row q-block per wave, FastLoop256<T, DEFF <= 128, ...>): the 256-row kernel has no generated
        // loop under either
        if constexpr (D == 128) {
            if (variant == 4) return kp.d <= 96 ? launch_d256_form<T, 96>(kp, softcap, stream) : launch_d256_form<T, 128>(kp, softcap, stream);
        } else {
            if (variant == 4) return launch_d256_form<T, 64>(kp, softcap, stream);
        }
        if (variant == 0 || variant == 3) {
            if (softcap) return launch_w64<T, D, true>(kp, stream);
            if constexpr (D == 128) {
                if (kp.d <= 96) return launch_w64<T, D, false, 96>(kp, stream);  // head-dim tile 96 (hopper/tile_size.h:10-54)
                if (persist_ok(kp)) return launch_w64<T, D, false, 128, true>(kp, stream);
            }
            return launch_w64<T, D, false>(kp, stream);
        }
        if (variant == 2) {
            if (softcap) return launch<T, D, 4, true>(kp, stream);
            return launch<T, D, 4, false>(kp, stream);
        }
        if (softcap) return launch<T, D, 8, true>(kp, stream);
        return launch<T, D, 8, false>(kp, stream);
    }
}

template <typename T>
int dispatch_hdim(const fa::KParams &kp, bool softcap, int variant, hipStream_t stream) {
    switch (head_dim_tile(std::max(kp.d, kp.dv))) {
        case 64: return dispatch_variant<T, 64>(kp, softcap, variant, stream);
        case 128: return dispatch_variant<T, 128>(kp, softcap, variant, stream);
        default: return dispatch_variant<T, 256>(kp, softcap, variant, stream);
    }
}

int block_m_of(int variant, int d) { return (variant == 2 || head_dim_tile(d) == 256) ? 128 : 256; }

}  // namespace

extern "C" {

uint32_t fa_fwd_params_size(void) { return (uint32_t)sizeof(fa_fwd_params); }
uint32_t fa_abi_version(void) { return FA_ABI_VERSION; }

#ifdef FA_TIMING
// developer-only: copy the phase timestamps of the last fwd_kernel_w64 launch (see fa_fwd_kernel_w64.h) to the host
int fa_debug_read_timing(unsigned long long *dst, int n_wg) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(fa::fa_timing_buf), sizeof(unsigned long long) * 32 * n_wg) == hipSuccess ? 0 : -1;
}
#endif
#ifdef FA_F8_DEBUG
// developer-only: read and clear the fp8 block-run counters (see fa_fwd_kernel_fp8.h)
int fa_debug_read_f8(unsigned long long *dst) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(fa::fa_f8_dbg), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(fa::fa_f8_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef FA_CYCLES
// developer-only: copy the fast-loop cycle stamps of the last fwd_kernel_w64 launch (see fa_fwd_kernel_w64.h) to the host
int fa_debug_read_cycles(unsigned long long *dst) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(fa::fa_cycle_buf), sizeof(unsigned long long) * 256 * 4 * 64) == hipSuccess ? 0 : -1;
}
#endif
void fa_set_default_variant(int32_t variant) { g_default_variant.store(variant); }
void fa_set_persist_mode(int32_t mode) { g_persist_mode.store(mode); }

const char *fa_strerror(int status) {
    switch (status) {
        case FA_OK: return "ok";
        case FA_ERR_NULL_POINTER: return "a required tensor pointer is NULL";
        case FA_ERR_BAD_DTYPE: return "FlashAttention only support fp16 and bf16 data type";
        case FA_ERR_BAD_HEAD_DIM:
            return "FlashAttention forward only supports head dimension at most 256, and head_size must be a multiple of 8";
        case FA_ERR_BAD_HEADS: return "Number of heads in key/value must divide number of heads in query";
        case FA_ERR_BAD_SHAPE: return "batch size must be positive and sequence lengths non-negative";
        case FA_ERR_BAD_STRIDE: return "tensor base pointers and row/head/batch strides must keep rows 16-byte aligned";
        case FA_ERR_UNSUPPORTED: return "feature not supported by this build of the forward";
        case FA_ERR_LAUNCH: return "kernel launch failed";
        case FA_ERR_BAD_ABI: return "fa_fwd_params abi_version/struct_size mismatch";
        case FA_ERR_NO_DEVICE: return "no gfx950 device";
        case FA_ERR_WORKSPACE: return "fp8 inputs need a 256-byte aligned workspace of fa_fwd_workspace_size() bytes";
        default: return "unknown status";
    }
}

int fa_fwd_tile_shape(int32_t d, int32_t dtype, int32_t is_causal, int32_t *block_m, int32_t *block_n) {
    (void)dtype;
    (void)is_causal;
    if (d <= 0 || d > 256 || d % 8) return FA_ERR_BAD_HEAD_DIM;
    if (block_m) *block_m = block_m_of(g_default_variant.load(), d);
    if (block_n) *block_n = fa::BLOCK_N;
    return FA_OK;
}

uint32_t fa_kvcache_append_params_size(void) { return (uint32_t)sizeof(fa_kvcache_append_params); }

int fa_kvcache_append(const fa_kvcache_append_params *p, void *stream_) {
    if (!p) return FA_ERR_NULL_POINTER;
    if (p->abi_version != FA_ABI_VERSION || p->struct_size != sizeof(fa_kvcache_append_params)) return FA_ERR_BAD_ABI;
    if (p->b <= 0 || p->h_k <= 0 || p->seqlen_new < 0 || p->seqlen_cache < 0) return FA_ERR_BAD_SHAPE;
    if (p->d <= 0 || p->d > 256 || p->d % 8 != 0) return FA_ERR_BAD_HEAD_DIM;
    if (p->seqlen_new == 0) return FA_OK;
    if (!p->k_new || !p->v_new || !p->k_cache || !p->v_cache || !p->cache_seqlens) return FA_ERR_NULL_POINTER;
    if (p->block_table && (p->page_block_size <= 0 || p->cache_batch_idx)) return FA_ERR_BAD_SHAPE;
    if (p->rotary_cos) {
        if (!p->rotary_sin) return FA_ERR_NULL_POINTER;
        if (p->dtype != FA_DTYPE_FP16 && p->dtype != FA_DTYPE_BF16) return FA_ERR_BAD_DTYPE;
        if (p->rotary_dim <= 0 || p->rotary_dim > p->d || p->rotary_dim % 16 != 0) return FA_ERR_BAD_SHAPE;  // (:1409-1410)
        if (reinterpret_cast<uintptr_t>(p->rotary_cos) % 16 != 0 || reinterpret_cast<uintptr_t>(p->rotary_sin) % 16 != 0)
            return FA_ERR_BAD_STRIDE;
    }
    const int64_t strides[] = {p->knew_batch_stride, p->knew_row_stride, p->knew_head_stride, p->vnew_batch_stride,
                               p->vnew_row_stride, p->vnew_head_stride, p->kcache_batch_stride, p->kcache_row_stride,
                               p->kcache_head_stride, p->vcache_batch_stride, p->vcache_row_stride, p->vcache_head_stride};
    for (int64_t s : strides)
        if (s % 8 != 0) return FA_ERR_BAD_STRIDE;
    const void *ptrs[] = {p->k_new, p->v_new, p->k_cache, p->v_cache};
    for (const void *ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return FA_ERR_BAD_STRIDE;
    const int64_t total = (int64_t)p->b * p->seqlen_new * p->h_k * ((p->d / 8 + 1) / 2);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 8);
    if (p->rotary_cos && p->dtype == FA_DTYPE_FP16)
        hipLaunchKernelGGL(kvcache_append_kernel<_Float16>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), *p);
    else  // (without rotary the element type does not matter: plain 16-byte copies)
        hipLaunchKernelGGL(kvcache_append_kernel<__bf16>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), *p);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

uint32_t fa_rotary_params_size(void) { return (uint32_t)sizeof(fa_rotary_params); }

int fa_rotary_apply(const fa_rotary_params *p, void *stream_) {
    if (!p) return FA_ERR_NULL_POINTER;
    if (p->abi_version != FA_ABI_VERSION || p->struct_size != sizeof(fa_rotary_params)) return FA_ERR_BAD_ABI;
    if (p->dtype != FA_DTYPE_FP16 && p->dtype != FA_DTYPE_BF16) return FA_ERR_BAD_DTYPE;
    if (p->b <= 0 || p->h <= 0 || p->s < 0) return FA_ERR_BAD_SHAPE;
    if (p->d <= 0 || p->d > 256 || p->d % 8 != 0) return FA_ERR_BAD_HEAD_DIM;
    if (p->rotary_dim <= 0 || p->rotary_dim > p->d || p->rotary_dim % 16 != 0) return FA_ERR_BAD_SHAPE;
    if (p->s == 0) return FA_OK;
    if (!p->src || !p->dst || !p->rotary_cos || !p->rotary_sin || !p->seqlen_offsets) return FA_ERR_NULL_POINTER;
    const int64_t strides[] = {p->src_batch_stride, p->src_row_stride, p->src_head_stride, p->dst_batch_stride,
                               p->dst_row_stride, p->dst_head_stride};
    for (int64_t s : strides)
        if (s % 8 != 0) return FA_ERR_BAD_STRIDE;
    const void *ptrs[] = {p->src, p->dst, p->rotary_cos, p->rotary_sin};
    for (const void *ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return FA_ERR_BAD_STRIDE;
    const int64_t total = (int64_t)p->b * p->s * p->h * ((p->d / 8 + 1) / 2);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 8);
    if (p->dtype == FA_DTYPE_FP16)
        hipLaunchKernelGGL(rotary_kernel<_Float16>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), *p);
    else
        hipLaunchKernelGGL(rotary_kernel<__bf16>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), *p);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

uint32_t fa_combine_params_size(void) { return (uint32_t)sizeof(fa_combine_params); }

int fa_fwd_combine(const fa_combine_params *p, void *stream_) {
    if (!p) return FA_ERR_NULL_POINTER;
    if (p->abi_version != FA_ABI_VERSION || p->struct_size != sizeof(fa_combine_params)) return FA_ERR_BAD_ABI;
    if (p->out_dtype != FA_DTYPE_FP32 && p->out_dtype != FA_DTYPE_FP16 && p->out_dtype != FA_DTYPE_BF16) return FA_ERR_BAD_DTYPE;
    if (p->num_splits <= 0 || p->num_splits > 256) return FA_ERR_BAD_SHAPE;  // "combine only supports num_splits at most 256"
    if (p->b < 0 || p->seqlen < 0 || p->h <= 0 || p->d <= 0) return FA_ERR_BAD_SHAPE;
    if (p->b == 0 || p->seqlen == 0) return FA_OK;
    if (!p->out_partial || !p->lse_partial || !p->out || !p->softmax_lse) return FA_ERR_NULL_POINTER;
    const int64_t total = (int64_t)p->b * p->seqlen * p->h * ((p->d + 3) / 4);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (p->out_dtype == FA_DTYPE_FP32) hipLaunchKernelGGL(combine_partials_kernel<float>, dim3(blocks), dim3(256), 0, stream, *p);
    else if (p->out_dtype == FA_DTYPE_FP16) hipLaunchKernelGGL(combine_partials_kernel<_Float16>, dim3(blocks), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(combine_partials_kernel<__bf16>, dim3(blocks), dim3(256), 0, stream, *p);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

int64_t fa_fwd_workspace_size(const fa_fwd_params *p) {
    if (!p) return FA_ERR_NULL_POINTER;
    if (p->abi_version != FA_ABI_VERSION || p->struct_size != sizeof(fa_fwd_params)) return FA_ERR_BAD_ABI;
    if (p->b <= 0 || p->h <= 0 || p->h_k <= 0 || p->d <= 0 || p->seqlen_q < 0 || p->seqlen_k < 0) return FA_ERR_BAD_SHAPE;
    if (p->cu_seqlens_q && (p->total_q < 0 || p->total_k < 0)) return FA_ERR_BAD_SHAPE;
    if (p->dtype == FA_DTYPE_FP8_E4M3) return fp8_native(p) ? 0 : fp8_plan(p).total;
    return split_plan(p, effective_variant(p)).total;
}

int fa_fwd_validate(const fa_fwd_params *p) {
    if (!p) return FA_ERR_NULL_POINTER;
    if (p->abi_version != FA_ABI_VERSION || p->struct_size != sizeof(fa_fwd_params)) return FA_ERR_BAD_ABI;
    if (p->dtype != FA_DTYPE_FP16 && p->dtype != FA_DTYPE_BF16 && p->dtype != FA_DTYPE_FP8_E4M3) return FA_ERR_BAD_DTYPE;
    if (p->b <= 0 || p->h <= 0 || p->h_k <= 0 || p->seqlen_q < 0 || p->seqlen_k < 0) return FA_ERR_BAD_SHAPE;
    if (p->d <= 0 || p->d > 256 || p->d % 8 != 0) return FA_ERR_BAD_HEAD_DIM;
    const bool fp8 = p->dtype == FA_DTYPE_FP8_E4M3;
    if (fp8 && p->d % 16 != 0) return FA_ERR_BAD_HEAD_DIM;  // hopper/flash_api.cpp:854-856
    if (p->attention_chunk < 0) return FA_ERR_BAD_SHAPE;
    if (p->d_v < 0 || p->d_v > 512 || p->d_v % 8 != 0) return FA_ERR_BAD_HEAD_DIM;  // 0 = d
    if (own_dv(p) && (fp8 || p->block_table || p->num_splits > 1)) return FA_ERR_UNSUPPORTED;
    if (generic_only(p) && p->p_dropout > 0.f) return FA_ERR_UNSUPPORTED;  // (no dropout on the FA3 surface)
    if (p->attention_chunk > 0 && p->s_dmask) return FA_ERR_UNSUPPORTED;  // (the S_dmask pass knows windows only)
    if (p->h % p->h_k != 0) return FA_ERR_BAD_HEADS;
    if ((p->cu_seqlens_q == nullptr) != (p->cu_seqlens_k == nullptr)) return FA_ERR_BAD_SHAPE;
    if (p->cu_seqlens_q && p->total_q < 0) return FA_ERR_BAD_SHAPE;
    const bool empty = (p->seqlen_q == 0) || (p->cu_seqlens_q && p->total_q == 0);
    if (!empty) {
        if (!p->q || !p->o || !p->softmax_lse) return FA_ERR_NULL_POINTER;
        if (p->seqlen_k > 0 && (!p->k || !p->v)) return FA_ERR_NULL_POINTER;
    }
    // 16-byte vector loads/stores: bases and strides must keep every row 16-byte aligned
    // (fp8 sources are read 8 bytes at a time by the expansion pass: same multiple-of-8-elements rule)
    const int64_t strides[] = {p->q_row_stride, p->q_head_stride, p->k_row_stride, p->k_head_stride,
                               p->v_row_stride, p->v_head_stride, p->o_row_stride, p->o_head_stride};
    for (int64_t s : strides)
        if (s % 8 != 0) return FA_ERR_BAD_STRIDE;
    // the kernel addresses a K/V tile as 64-bit tile base + 32-bit (row * stride) lane offset
    if (p->k_row_stride < 0 || p->v_row_stride < 0 || p->k_row_stride >= (1 << 24) || p->v_row_stride >= (1 << 24))
        return FA_ERR_BAD_STRIDE;
    if (!p->cu_seqlens_q) {
        const int64_t bs[] = {p->q_batch_stride, p->k_batch_stride, p->v_batch_stride, p->o_batch_stride};
        for (int64_t s : bs)
            if (s % 8 != 0) return FA_ERR_BAD_STRIDE;
    }
    const void *ptrs[] = {p->q, p->k, p->v, p->o};
    for (const void *ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) % (fp8 && ptr != p->o ? 8 : 16) != 0) return FA_ERR_BAD_STRIDE;
    if (fp8 && !fp8_native(p) && !empty && p->seqlen_k > 0) {
        if (!p->workspace || reinterpret_cast<uintptr_t>(p->workspace) % 256 != 0 ||
            (int64_t)p->workspace_bytes < fp8_plan(p).total)
            return FA_ERR_WORKSPACE;
    }
    if (p->num_splits < 0) return FA_ERR_BAD_SHAPE;
    if (!fp8 && !empty && p->seqlen_k > 0) {
        const SplitPlan sp = split_plan(p, effective_variant(p));
        if (sp.splits > 1 && (!p->workspace || reinterpret_cast<uintptr_t>(p->workspace) % 256 != 0 ||
                              (int64_t)p->workspace_bytes < sp.total))
            return FA_ERR_WORKSPACE;
    }
    if (p->softcap < 0.f || std::isnan(p->softcap) || std::isnan(p->softmax_scale)) return FA_ERR_BAD_SHAPE;
    if (p->alibi_slopes && (reinterpret_cast<uintptr_t>(p->alibi_slopes) % 4 != 0 || p->alibi_slopes_batch_stride < 0 ||
                            p->alibi_slopes_batch_stride > 0x7fffffff))
        return FA_ERR_BAD_STRIDE;
    if (!(p->p_dropout >= 0.f && p->p_dropout < 1.f)) return FA_ERR_BAD_SHAPE;  // "p_dropout must be in [0, 1)"
    if (p->p_dropout > 0.f) {
        if (fp8 || p->block_table || p->kv_batch_idx || p->leftpad_k) return FA_ERR_UNSUPPORTED;  // training path only
        if (p->softcap > 0.f) return FA_ERR_UNSUPPORTED;  // "Softcapping does not support dropout for now" (flash_api.cpp:377)
        if (!p->rng_state || reinterpret_cast<uintptr_t>(p->rng_state) % 8 != 0) return FA_ERR_NULL_POINTER;
    } else if (p->s_dmask) {
        return FA_ERR_UNSUPPORTED;  // the randval tensor only exists under dropout
    }
    if ((p->flags & FA_FLAG_SDMASK_SIGNED) && p->s_dmask) {
        if (p->s_dmask_rows < p->seqlen_q || p->s_dmask_cols < p->seqlen_k || p->s_dmask_block_n <= 0) return FA_ERR_BAD_SHAPE;
        if (p->seqlen_k > 32768) return FA_ERR_UNSUPPORTED;  // (one row of scores in LDS)
    }
    if (p->kv_batch_idx && (p->cu_seqlens_q || fp8)) return FA_ERR_UNSUPPORTED;  // dense 16-bit caches only
    if (p->leftpad_k && (p->block_table || fp8)) return FA_ERR_UNSUPPORTED;  // (:1396 "Paged KV and leftpad_k" not together)
    if (p->block_table) {
        if (fp8 || p->kv_batch_idx) return FA_ERR_UNSUPPORTED;  // "Paged KVcache does not support cache_batch_idx" (:1247)
        if (p->page_block_size <= 0) return FA_ERR_BAD_SHAPE;  // any size (the FA2 entry point keeps its % 256 rule in Python)
        if (p->block_table_batch_stride < 0 || p->block_table_batch_stride > 0x7fffffff) return FA_ERR_BAD_STRIDE;
    }
    return FA_OK;
}

int fa_fwd(const fa_fwd_params *p, void *stream_) {
    const int st = fa_fwd_validate(p);
    if (st != FA_OK) return st;
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    if (p->d_v > 256) {
        // V head dims above the widest tile (hopper/flash_api.cpp:783-792 allows up to 512 beside q/k <= 64): one launch per 256
        // columns of V and O -- the scores are formed again for each (d <= 64: a small part of the work), the LSE is written
        // by every launch with the same value.  Strides are untouched: the launches differ in the V / O column offset only.
        for (int c = 0; c < p->d_v; c += 256) {
            fa_fwd_params part = *p;
            part.v = static_cast<const char *>(p->v) + (size_t)c * 2;
            part.o = static_cast<char *>(p->o) + (size_t)c * 2;
            part.d_v = std::min(256, p->d_v - c);
            const int st_part = fa_fwd(&part, stream_);
            if (st_part != FA_OK) return st_part;
        }
        return FA_OK;
    }
    int variant = effective_variant(p);
    int block_m = block_m_of(variant, wide_dim(p));
    {   // softcap or ALiBi at head dims <= 128 (measured b4 s4096: softcap d128 471, d64 322, ALiBi 231 / 182 TFLOP/s through the C++
        // paths of the 256-row kernel): the generated loops that cap / bias scores exist for the 32-row-per-wave shape only ->
        // its DEFF = 64 / 96 / 128 instantiations
        const bool nothing_ = p->seqlen_q == 0 || p->seqlen_k == 0 || (p->cu_seqlens_q && p->total_q == 0);
        if (((p->softcap > 0.f) != (p->alibi_slopes != nullptr)) && variant == 0 && p->d <= 128 && !generic_only(p) &&
            !p->block_table && p->p_dropout == 0.f && !nothing_ && split_plan(p, variant).splits <= 1 &&
            !(p->dtype == FA_DTYPE_FP8_E4M3 && fp8_native(p))) {
            variant = 4;
            block_m = 128;
        }
    }

    fa::KParams kp{};
    kp.q = p->q; kp.k = p->k; kp.v = p->v; kp.o = p->o; kp.lse = p->softmax_lse;
    const bool fp8 = p->dtype == FA_DTYPE_FP8_E4M3;
    const bool nothing = p->seqlen_q == 0 || p->seqlen_k == 0 || (p->cu_seqlens_q && p->total_q == 0);
    int64_t ws_q_row = 0, ws_k_row = 0;
    const bool native8 = fp8 && fp8_native(p);
    if (fp8 && !native8 && !nothing) {
        const Fp8Plan pl = fp8_plan(p);
        char *ws = static_cast<char *>(p->workspace);
        const int rpb_q = p->cu_seqlens_q ? 0 : p->seqlen_q, rpb_k = p->cu_seqlens_q ? 0 : p->seqlen_k;
        auto expand = [&](const void *src, void *dst, int64_t rows, int rpb, int heads, int64_t bs, int64_t rs, int64_t hs) {
            const int64_t total = rows * heads * (p->d / 8);
            if (total == 0) return;
            const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 8);
            hipLaunchKernelGGL(expand_fp8_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const uint8_t *>(src),
                               static_cast<uint32_t *>(dst), rows, rpb, heads, p->d, bs, rs, hs);
        };
        expand(p->q, ws, pl.rows_q, rpb_q, p->h, p->q_batch_stride, p->q_row_stride, p->q_head_stride);
        expand(p->k, ws + pl.q_bytes, pl.rows_k, rpb_k, p->h_k, p->k_batch_stride, p->k_row_stride, p->k_head_stride);
        expand(p->v, ws + pl.q_bytes + pl.kv_bytes, pl.rows_k, rpb_k, p->h_k, p->v_batch_stride, p->v_row_stride, p->v_head_stride);
        if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
        kp.q = ws; kp.k = ws + pl.q_bytes; kp.v = ws + pl.q_bytes + pl.kv_bytes;
        ws_q_row = (int64_t)p->h * p->d;
        ws_k_row = (int64_t)p->h_k * p->d;
    }
    kp.cu_seqlens_q = p->cu_seqlens_q; kp.cu_seqlens_k = p->cu_seqlens_k;
    kp.seqused_q = p->seqused_q; kp.seqused_k = p->seqused_k;
    kp.q_batch_stride = p->q_batch_stride; kp.q_row_stride = p->q_row_stride; kp.q_head_stride = p->q_head_stride;
    kp.k_batch_stride = p->k_batch_stride; kp.k_row_stride = p->k_row_stride; kp.k_head_stride = p->k_head_stride;
    kp.v_batch_stride = p->v_batch_stride; kp.v_row_stride = p->v_row_stride; kp.v_head_stride = p->v_head_stride;
    if (fp8) {
        if (!native8) {  // the expanded copies are contiguous (rows, heads, d)
            kp.q_row_stride = ws_q_row; kp.q_head_stride = p->d; kp.q_batch_stride = ws_q_row * p->seqlen_q;
            kp.k_row_stride = kp.v_row_stride = ws_k_row; kp.k_head_stride = kp.v_head_stride = p->d;
            kp.k_batch_stride = kp.v_batch_stride = ws_k_row * p->seqlen_k;
        }
        kp.q_descale = p->q_descale; kp.k_descale = p->k_descale; kp.v_descale = p->v_descale;
        kp.qd_bs = (int32_t)p->q_descale_batch_stride; kp.qd_hs = (int32_t)p->q_descale_head_stride;
        kp.kd_bs = (int32_t)p->k_descale_batch_stride; kp.kd_hs = (int32_t)p->k_descale_head_stride;
        kp.vd_bs = (int32_t)p->v_descale_batch_stride; kp.vd_hs = (int32_t)p->v_descale_head_stride;
    }
    kp.o_batch_stride = p->o_batch_stride; kp.o_row_stride = p->o_row_stride; kp.o_head_stride = p->o_head_stride;
    kp.b = p->b; kp.seqlen_q = p->seqlen_q; kp.seqlen_k = p->seqlen_k; kp.h = p->h; kp.h_k = p->h_k; kp.d = p->d;
    kp.total_q = p->total_q;
    kp.dv = own_dv(p) ? p->d_v : p->d;
    kp.chunk = p->attention_chunk;
    kp.h_ratio = p->h / p->h_k;
    kp.num_m_blocks = (p->seqlen_q + block_m - 1) / block_m;
    const int64_t tiles = (int64_t)kp.num_m_blocks * p->h * p->b;
    if (tiles == 0) return FA_OK;  // nothing to compute (seqlen_q == 0)
    if (tiles > 0x7fffffff) return FA_ERR_BAD_SHAPE;
    kp.num_tiles = (int32_t)tiles;
    // scheduling (tile_of_wg): whole (batch, kv head) units -- everything that streams one head's K/V stays on one XCD -- for as
    // many units as deal evenly over the 8 XCDs, the remaining heads by m_block of the GQA group; problems with fewer than two
    // units per XCD entirely by m_block (fill the chip first, L2 reuse second).  (20 units used to be dealt 3+3+3+3+2+2+2+2:
    // the launch took as long as 24, tools/hdim_bench.py d192 b2 h10 2.88 ms instead of 2.41.)
    const int64_t bk_units = (int64_t)p->b * p->h_k;
    const int64_t per_kvh = (int64_t)kp.h_ratio * kp.num_m_blocks;
    const int64_t whole_units = bk_units >= 16 ? bk_units / 8 * 8 : 0;
    kp.unit_tiles = (int32_t)per_kvh;
    const int64_t whole_slots = whole_units / 8 * per_kvh;
    const int64_t rem_units = (tiles - whole_slots * 8 + kp.h_ratio - 1) / kp.h_ratio;
    const int64_t grid = 8 * (whole_slots + (rem_units + 7) / 8 * kp.h_ratio);
    if (whole_slots > 0x7fffffff) return FA_ERR_BAD_SHAPE;
    kp.whole_slots = (int32_t)whole_slots;
    if (grid > 0x7fffffff) return FA_ERR_BAD_SHAPE;
    kp.grid = (int32_t)grid;
    {   // compute units of the current device (cached per device ordinal)
        static std::atomic<int> cus[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        int n = cus[dev & 63].load(std::memory_order_relaxed);
        if (n == 0) {
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            cus[dev & 63].store(n, std::memory_order_relaxed);
        }
        kp.num_cus = n;
    }
    // split-KV: `splits` copies of the grid; partial results go to the workspace and are merged below
    const SplitPlan sp = nothing ? SplitPlan{1, 0, 0, 0} : split_plan(p, variant == 4 ? 0 : variant);
    kp.num_splits = sp.splits;
    if (sp.splits > 1) {
        if (grid * sp.splits > 0x7fffffff) return FA_ERR_BAD_SHAPE;
        char *ws = static_cast<char *>(p->workspace);
        kp.o = ws;
        kp.lse = reinterpret_cast<float *>(ws + sp.o_bytes);
        kp.o_row_stride = (int64_t)p->h * p->d; kp.o_head_stride = p->d; kp.o_batch_stride = kp.o_row_stride * p->seqlen_q;
        kp.o_split_stride = kp.o_batch_stride * p->b;
        kp.lse_split_stride = (int64_t)p->b * p->h * p->seqlen_q;
    }

    // window normalisation: csrc/flash_attn/flash_api.cpp:396-402
    int wl = p->window_size_left, wr = p->window_size_right;
    if (p->is_causal) wr = 0;
    if (p->flags & FA_FLAG_FA3_WINDOW) {
        // FA3 rule (hopper/flash_api.cpp:152-153, 589-590): a missing side becomes seqlen_k - 1 / seqlen_q - 1, which never
        // masks anything = unbounded here; sides are taken as given otherwise
    } else {
        if (wl >= p->seqlen_k) wl = -1;
        if (wr >= p->seqlen_k) wr = -1;
        if (p->is_causal) wr = 0;
        // set_params_fprop csrc/flash_attn/flash_api.cpp:141-142: a one-sided window gets seqlen_k on the other side.
        // For a left-only window that is NOT the same as unbounded when seqlen_q > seqlen_k (the bottom-right aligned
        // diagonal starts left of key 0), so it is mirrored.  The symmetric rule (right-only -> left = seqlen_k) never
        // masks anything (row + sk - sq - seqlen_k < 0 for every row) and is left as "unbounded".
        if (wl >= 0 && wr < 0) wr = p->seqlen_k;
    }
    kp.window_left = wl;
    kp.window_right = wr;

    kp.alibi = p->alibi_slopes;
    kp.alibi_bs = (int32_t)p->alibi_slopes_batch_stride;
    kp.leftpad_k = p->leftpad_k;
    // dropout: keep iff randval <= floor(255 (1 - p)); 255 = everything kept = the branch is off
    // (the 8-bit quantisation of the reference's ROCm back-end: any p > 0 gives a threshold <= 254, i.e. at least 1/256 of the
    //  elements are dropped however small p is; the kept ones are scaled by 1 / (1 - p))
    kp.drop_thr = p->p_dropout > 0.f ? (int32_t)std::floor(255.0 * (1.0 - (double)p->p_dropout)) : 255;
    kp.rp_dropout = p->p_dropout > 0.f ? 1.f / (1.f - p->p_dropout) : 1.f;
    kp.rng_state = p->rng_state;
    const bool sdmask_signed = (p->flags & FA_FLAG_SDMASK_SIGNED) && p->s_dmask;
    kp.s_dmask = sdmask_signed ? nullptr : p->s_dmask;
    kp.kv_batch_idx = p->cu_seqlens_q ? nullptr : p->kv_batch_idx;
    kp.block_table = p->block_table;
    kp.bt_bs = (int32_t)p->block_table_batch_stride;
    kp.page_size = p->page_block_size;

    const bool softcap = p->softcap > 0.f;
    constexpr float kLog2e = 1.4426950408889634f;
    if (softcap) {  // set_params_fprop csrc/flash_attn/flash_api.cpp:103-117
        kp.softcap_pre = p->softmax_scale / p->softcap;
        kp.scale = p->softcap;
        kp.scale_log2 = p->softcap * kLog2e;
    } else {
        kp.softcap_pre = 0.f;
        kp.scale = p->softmax_scale;
        kp.scale_log2 = p->softmax_scale * kLog2e;
    }

    if (native8) {  // e4m3 operands straight into the block-scaled MFMA (effective_variant() is 0 here: 256-row workgroups)
        constexpr int smem = fa::smem_bytes_fp8();
        static std::atomic<uint64_t> attr_set{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = uint64_t(1) << (dev & 63);
        if (!(attr_set.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(fa::fwd_kernel_fp8), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
                (void)hipGetLastError();
                return FA_ERR_LAUNCH;
            }
            attr_set.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL(fa::fwd_kernel_fp8, dim3(kp.grid), dim3(256), smem, stream, kp);
        return hipGetLastError() == hipSuccess ? FA_OK : FA_ERR_LAUNCH;
    }
    const bool bf16 = p->dtype == FA_DTYPE_BF16 || fp8;  // fp8: out is bf16
    const int st_main = bf16 ? dispatch_hdim<__bf16>(kp, softcap, variant, stream)
                             : dispatch_hdim<_Float16>(kp, softcap, variant, stream);
    if (st_main == FA_OK && sdmask_signed && !nothing) {
        const int nrb = (p->seqlen_q + 7) / 8;
        const int64_t blocks = (int64_t)nrb * p->h * p->b;
        const int nblk = (p->seqlen_k + p->s_dmask_block_n - 1) / p->s_dmask_block_n;
        const size_t smem = sizeof(float) * ((size_t)8 * p->d + ((p->seqlen_k + 3) & ~3) + nblk + 4);
        if (blocks > 0x7fffffff) return FA_ERR_BAD_SHAPE;
        auto launch_sd = [&](auto tag) -> int {
            using T = decltype(tag);
            auto kernel = sdmask_kernel<T>;
            if (smem > 65536 &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
                (void)hipGetLastError();
                return FA_ERR_LAUNCH;
            }
            hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), smem, stream, kp, reinterpret_cast<T *>(p->s_dmask),
                               p->s_dmask_rows, p->s_dmask_cols, p->s_dmask_block_n);
            return hipGetLastError() == hipSuccess ? FA_OK : FA_ERR_LAUNCH;
        };
        const int st_sd = bf16 ? launch_sd(__bf16{}) : launch_sd(_Float16{});
        if (st_sd != FA_OK) return st_sd;
    }
    if (st_main != FA_OK || sp.splits <= 1) return st_main;
    // merge the partial results into the caller's out / softmax_lse
    const int64_t total = (int64_t)p->b * p->seqlen_q * p->h * (p->d / 8);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 8);
    if (bf16)
        hipLaunchKernelGGL(combine_splits_kernel<__bf16>, dim3(blocks), dim3(256), 0, stream, static_cast<const float *>(kp.o),
                           kp.lse, static_cast<__bf16 *>(p->o), p->softmax_lse, sp.splits, p->b, p->seqlen_q, p->h, p->d,
                           p->o_batch_stride, p->o_row_stride, p->o_head_stride);
    else
        hipLaunchKernelGGL(combine_splits_kernel<_Float16>, dim3(blocks), dim3(256), 0, stream,
                           static_cast<const float *>(kp.o), kp.lse, static_cast<_Float16 *>(p->o), p->softmax_lse,
                           sp.splits, p->b, p->seqlen_q, p->h, p->d, p->o_batch_stride, p->o_row_stride, p->o_head_stride);
    if (hipGetLastError() != hipSuccess) return FA_ERR_LAUNCH;
    return FA_OK;
}

}  // extern "C"
