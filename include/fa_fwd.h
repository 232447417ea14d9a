/*
 * fa_fwd.h — C-ABI of the MI355X (gfx950) FlashAttention forward.
 *
 * This is the drop-in boundary of the hot path.  Plain C, raw
 * device pointers, element strides, int status codes; no torch types.  It is
 * what a binding for the reference's forward entry points would call:
 *
 *   reference interface replaced                                  entry point here
 *   ------------------------------------------------------------  -----------------
 *   mha_fwd          csrc/flash_attn/flash_api.cpp:350-512        fa_fwd (dense)
 *   mha_varlen_fwd   csrc/flash_attn/flash_api.cpp:514-755        fa_fwd (cu_seqlens_* set)
 *   mha_fwd (FA3)    hopper/flash_api.cpp:672-1198                fa_fwd (+ seqused_*, fp8 e4m3 + descale)
 *   set_params_fprop csrc/flash_attn/flash_api.cpp:26-159         fa_fwd_params (field for field)
 *   Flash_fwd_params csrc/flash_attn/src/flash.h:48-143,
 *                    hopper/flash.h:37-168                        fa_fwd_params
 *   flash_attention_forward(FlashAttentionParams&, cudaStream_t)
 *                    standalone/include/flash_api.h:222-225       fa_fwd(const fa_fwd_params*, void*)
 *   mha_fwd_kvcache  csrc/flash_attn/flash_api.cpp:1202-1476      fa_kvcache_append + fa_fwd (seqused_k, kv_batch_idx)
 *   error codes      standalone/src/flash_api.cu:403-426          FA_ERR_* / fa_strerror
 *
 * Conventions (same as the reference's params struct):
 *   - strides are in ELEMENTS, not bytes (csrc/flash_attn/flash_api.cpp:64-73);
 *     the last (head_dim) stride of q/k/v/o must be 1;
 *   - dense layout  q:(b, seqlen_q, h, d)  k,v:(b, seqlen_k, h_k, d)  o like q,
 *     softmax_lse:(b, h, seqlen_q) fp32;
 *   - varlen layout q:(total_q, h, d) k,v:(total_k, h_k, d), cu_seqlens_{q,k}
 *     int32 (b+1) device arrays, softmax_lse:(h, total_q) fp32
 *     (csrc/flash_attn/flash_api.cpp:652); *_batch_stride is ignored;
 *   - the callee never allocates, never synchronises and launches on `stream`;
 *   - outputs are written in place; inputs are borrowed.
 */
#ifndef FA_FWD_H_
#define FA_FWD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_ABI_VERSION 12

#define FA_FLAG_FA3_WINDOW 1
#define FA_FLAG_SDMASK_SIGNED 2   /* s_dmask is the reference's sign-encoded probability tensor (below), not random bytes */

/* element types of q/k/v (o has the same type; fp8 inputs produce bf16 o) */
enum fa_dtype {
    FA_DTYPE_FP16 = 0,
    FA_DTYPE_BF16 = 1,
    FA_DTYPE_FP8_E4M3 = 2, /* OCP e4m3fn; hopper/flash_api.cpp:714-722 */
    FA_DTYPE_FP32 = 3      /* output type of fa_fwd_combine only */
};

/* status codes: 0 ok, negative = rejected before launch, nothing was written */
enum fa_status {
    FA_OK = 0,
    FA_ERR_NULL_POINTER = -1,      /* a required pointer is NULL */
    FA_ERR_BAD_DTYPE = -2,         /* "FlashAttention only support fp16 and bf16 data type" */
    FA_ERR_BAD_HEAD_DIM = -3,      /* d > 256 or d % 8 != 0 */
    FA_ERR_BAD_HEADS = -4,         /* h % h_k != 0 */
    FA_ERR_BAD_SHAPE = -5,         /* b <= 0, negative lengths, ... */
    FA_ERR_BAD_STRIDE = -6,        /* misaligned rows: strides must keep 16-byte row alignment */
    FA_ERR_UNSUPPORTED = -7,       /* feature accepted by the ABI but not built for this combination */
    FA_ERR_LAUNCH = -8,            /* hipLaunchKernel failed */
    FA_ERR_BAD_ABI = -9,           /* params->abi_version / struct size mismatch */
    FA_ERR_NO_DEVICE = -10,        /* not a gfx950 device */
    FA_ERR_WORKSPACE = -11         /* fp8 inputs / split-KV need params->workspace of fa_fwd_workspace_size() bytes */
};

/*
 * Mirrors Flash_fwd_params (csrc/flash_attn/src/flash.h:48-143) restricted to
 * the forward hot path, plus the FA3 additions used by BASELINE config 5
 * (hopper/flash.h:37-168: seqused_q/k, q/k/v descale).
 */
typedef struct fa_fwd_params {
    uint32_t abi_version; /* FA_ABI_VERSION */
    uint32_t struct_size; /* sizeof(fa_fwd_params) */

    /* tensors (device pointers) */
    const void *q;
    const void *k;
    const void *v;
    void *o;
    float *softmax_lse;

    /* element strides; head_dim stride is 1 */
    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;

    /* sizes */
    int32_t b;        /* batch (varlen: number of sequences) */
    int32_t seqlen_q; /* dense: seqlen_q; varlen: max_seqlen_q */
    int32_t seqlen_k; /* dense: seqlen_k; varlen: max_seqlen_k */
    int32_t h;        /* query heads */
    int32_t h_k;      /* key/value heads; h % h_k == 0; q head i reads kv head i / (h/h_k) */
    int32_t d;        /* head dim, multiple of 8, <= 256 */
    int32_t total_q;  /* varlen: rows of q (lse row length); dense: ignored */
    int32_t dtype;    /* enum fa_dtype */

    /* varlen bookkeeping (NULL => dense); BlockInfo csrc/flash_attn/src/block_info.h:12-45 */
    const int32_t *cu_seqlens_q; /* (b+1) */
    const int32_t *cu_seqlens_k; /* (b+1) */
    const int32_t *seqused_q;    /* (b) optional: rows actually used (FA3 hopper/seqlen.h:32-93) */
    const int32_t *seqused_k;    /* (b) optional: keys actually used */

    /* softmax */
    float softmax_scale; /* scores = q.k * softmax_scale */
    float softcap;       /* >0: scores = softcap * tanh(scores / softcap) */

    /* masking: bottom-right aligned (flash_attn/flash_attn_interface.py:1164-1174) */
    int32_t is_causal;         /* != 0 => window_size_right = 0 */
    int32_t window_size_left;  /* <0: unbounded */
    int32_t window_size_right; /* <0: unbounded */

    /* fp8 only: per-(batch, kv head) fp32 descales (hopper/flash_api.cpp:1115-1146); NULL = 1.0 */
    const float *q_descale, *k_descale, *v_descale;
    int64_t q_descale_batch_stride, q_descale_head_stride;
    int64_t k_descale_batch_stride, k_descale_head_stride;
    int64_t v_descale_batch_stride, v_descale_head_stride;

    /* performance hint, never changes results: 0 = library default */
    int32_t kernel_variant;
    int32_t total_k;      /* varlen: rows of k/v (only needed for fp8 inputs: size of the expansion workspace) */

    /* caller-provided scratch (the callee never allocates), 256-byte aligned, fa_fwd_workspace_size() bytes.
     * fp8: head dim 128 (dense / varlen / causal / right window) runs natively -- e4m3 operands straight into the block-scaled
     * MFMA, no workspace (fa_fwd_workspace_size() returns 0).  The other fp8 shapes (other head dims, softcap, left windows,
     * K or V of 2 GiB or more per batch entry) expand q/k/v to bf16 here (exact: every e4m3 value is a bf16 value) and run the
     * 16-bit mainloop; either way the descales are folded into the softmax scale and the final normalisation.
     * 16-bit dense problems that split the key range (num_splits) keep their fp32 partial O / LSE here. */
    void *workspace;
    uint64_t workspace_bytes;

    /* ALiBi (csrc/flash_attn/src/alibi.h:18-71, set_params_alibi csrc/flash_attn/flash_api.cpp:331-349): fp32 slopes,
     * (h) [batch stride 0] or (b, h); NULL = off.  The bias added to the scaled (and soft-capped) score of query i,
     * key j is  -slope * |i + seqlen_k - seqlen_q - j|  (tests/test_flash_attn.py:29-56); under a causal mask this
     * differs from the reference kernel's  +slope * j  only by a per-row constant (same O; LSE includes the bias). */
    const float *alibi_slopes;
    int64_t alibi_slopes_batch_stride;

    /* KV-cache decode (mha_fwd_kvcache csrc/flash_attn/flash_api.cpp:1202-1476; FA3 kv_batch_idx
     * hopper/flash_api.cpp:686): dense layout only; batch i reads k/v rows of cache entry kv_batch_idx[i]
     * (NULL = i).  The valid length of each cache entry is given through seqused_k. */
    const int32_t *kv_batch_idx;

    /* Paged KV cache (csrc/flash_attn/flash_api.cpp:1245-1266, 538-560; src/flash_fwd_kernel.h:560-576): k and v are
     * (num_blocks, page_block_size, h_k, d) -- k/v_batch_stride is the page stride -- and key row j of batch i lives in
     * page block_table[i * block_table_batch_stride + j / page_block_size], row j % page_block_size.
     * Any page_block_size >= 1 (multiples of 64 take the tile-granular lookup, others a per-row one).  NULL = contiguous. */
    const int32_t *block_table;
    int64_t block_table_batch_stride;
    int32_t page_block_size;
    /* Split-KV (set_params_splitkv / num_splits_heuristic csrc/flash_attn/flash_api.cpp:257-329, combine kernel
     * src/flash_fwd_kernel.h:1108-1290): 1 = off, N > 1 = the key range of every tile is cut into N parts computed by
     * N workgroups and merged by a second launch, 0 = library heuristic (splits only dense problems with few tiles,
     * i.e. decode).  Needs params->workspace of fa_fwd_workspace_size() bytes when the effective value is > 1 (the partial
     * outputs and LSEs of the parts, both fp32 like the reference's out_accum / softmax_lse_accum).
     * The default-initialised struct (0) therefore may split: callers without a workspace must pass 1. */
    int32_t num_splits;

    /* Left-padded keys (leftpad_k of mha_varlen_fwd / mha_fwd_kvcache, csrc/flash_attn/src/block_info.h:22-35): the first
     * leftpad_k[i] key rows of batch i are padding -- the kernel starts reading at row leftpad_k[i] and the valid length
     * becomes (seqused_k or the sequence length) - leftpad_k[i].  (b) int32 or NULL.  Not with block_table. */
    const int32_t *leftpad_k;

    /* Attention dropout (csrc/flash_attn/flash_api.cpp:486-493, src/dropout.h).  p_dropout in [0, 1): every probability
     * is kept with probability ~(1 - p) and the kept ones scaled by 1 / (1 - p).  The decision for (batch, head, query i,
     * key j) is an 8-bit counter-based random value r = fa_rand8(seed, offset, batch * h + head, i, j) compared with
     * floor(255 (1 - p)): kept iff r <= floor(255 (1 - p)) -- the "randval" convention of the reference's ROCm tests
     * (tests/test_flash_attn_ck.py:34-38).  rng_state: device pointer to {seed, offset} (2 x uint64), read by the
     * kernel (no host sync), the same pair must be given to fa_bwd.  s_dmask (optional, testing): receives r as uint8,
     * dense (b, h, seqlen_q, seqlen_k), varlen (h, total_q, seqlen_k [= max_seqlen_k]).  Dropout runs on the
     * compiler-scheduled kernel shape (no split-KV). */
    float p_dropout;
    /* FA_FLAG_* bits.  FA_FLAG_FA3_WINDOW: window sides follow the FA3 rule (hopper/flash_api.cpp:152-153, 589-590): a
     * negative side is UNBOUNDED and stays so; without the flag a one-sided window gets seqlen_k on the other side as
     * set_params_fprop does (csrc/flash_attn/flash_api.cpp:141-142), which masks rows when seqlen_q > seqlen_k. */
    int32_t flags;
    const uint64_t *rng_state;
    uint8_t *s_dmask;
    /* ABI v11 -- FA_FLAG_SDMASK_SIGNED: s_dmask is what the reference's CUDA forward returns for return_softmax
     * (csrc/flash_attn/src/flash_fwd_kernel.h:350-360, src/dropout.h:26-33; decoded by tests/test_flash_attn.py:411-526):
     * (b, h, s_dmask_rows, s_dmask_cols) elements of the INPUT dtype, row-major, rows / cols = seqlen_q / seqlen_k rounded up
     * to 128 (varlen: the max_seqlen's; batch entry i's block at [i, :, 0:seqlen_q_i, 0:seqlen_k_i]).  Element (row, key) =
     * exp(score - m) with m the maximum of the row's scores over key blocks [key / s_dmask_block_n, last] -- the running
     * maximum of a sweep that walks the key blocks from the last to the first, as the reference's kernel does -- and a
     * NEGATIVE sign where dropout discards the element.  s_dmask_block_n = the reference's kBlockN for this head dim
     * (flash_attn/flash_attn_interface.py:23-46).  Testing aid like the reference's: written by a pass of its own behind the
     * forward (fa_sdmask in fa_fwd_api.hip), seqlen_k <= 32768. */
    int32_t s_dmask_rows, s_dmask_cols, s_dmask_block_n;
    /* ABI v12 -- attention_chunk (FA3, hopper/flash_api.cpp:148-160, hopper/mask.h:116-119, hopper/block.h:30-42; oracle
     * construct_chunk_mask hopper/test_util.py:193-223): C > 0 = query i only sees the keys of its own chunk,
     * [floor((i + seqlen_k - seqlen_q) / C) * C, ... + C), intersected with the causal / window mask.  0 = off.  Forward only
     * (the reference's backward has no such argument, hopper/flash_api.cpp:1523), 16-bit and expanded-fp8 kernels. */
    int32_t attention_chunk;
    /* ABI v12 -- head dim of V and O when it differs from d (FA3 "headdim_v", hopper/flash_api.cpp:764,782-792: q/k in
     * (128, 192] with v in (96, 128], or q/k <= 64 with v <= 512): v is (.., h_k, d_v), o (.., h, d_v).  0 = d.  Multiple of 8,
     * <= 512; above 256 the library runs one launch per 256 columns of V.  16-bit types; not with split-KV, paged or fp8. */
    int32_t d_v;
    int32_t reserved_v12;
} fa_fwd_params;

/* Validate and enqueue the forward on `stream` (a hipStream_t; NULL = default
 * stream).  Returns FA_OK or a negative fa_status; asynchronous. */
int fa_fwd(const fa_fwd_params *params, void *stream);

/* Scratch bytes fa_fwd needs in params->workspace for these params (0 for fp16/bf16 inputs; <0 = fa_status). */
int64_t fa_fwd_workspace_size(const fa_fwd_params *params);

/* Validation only (what mha_fwd's TORCH_CHECKs do); no device access. */
int fa_fwd_validate(const fa_fwd_params *params);

/* Human-readable text for a status code (static storage). */
const char *fa_strerror(int status);

/* sizeof(fa_fwd_params) as compiled, for binding self-checks. */
uint32_t fa_fwd_params_size(void);

/* FA_ABI_VERSION as compiled. */
uint32_t fa_abi_version(void);

/* Tile geometry chosen for (d, dtype, causal): writes block_m / block_n.
 * Role of tile_size_fwd_sm90 (hopper/tile_size.h:10-54). */
int fa_fwd_tile_shape(int32_t d, int32_t dtype, int32_t is_causal, int32_t *block_m, int32_t *block_n);

/*
 * In-place append of new keys/values to a KV cache -- the "Append_KV" step of mha_fwd_kvcache
 * (csrc/flash_attn/flash_api.cpp:1354-1382, src/flash_fwd_kernel.h:651-735), as its own HBM-bound launch:
 * rows k_new[i, 0:seqlen_new) go to k_cache[idx(i), cache_seqlens[i] + 0:seqlen_new) (idx = cache_batch_idx or
 * identity); rows that would fall past seqlen_cache are dropped.  16-bit elements, head_dim stride 1, d % 8 == 0.
 */
typedef struct fa_kvcache_append_params {
    uint32_t abi_version; /* FA_ABI_VERSION */
    uint32_t struct_size; /* sizeof(fa_kvcache_append_params) */
    const void *k_new; /* (b, seqlen_new, h_k, d) */
    const void *v_new;
    void *k_cache;     /* (b_cache, seqlen_cache, h_k, d) */
    void *v_cache;
    int64_t knew_batch_stride, knew_row_stride, knew_head_stride;
    int64_t vnew_batch_stride, vnew_row_stride, vnew_head_stride;
    int64_t kcache_batch_stride, kcache_row_stride, kcache_head_stride;
    int64_t vcache_batch_stride, vcache_row_stride, vcache_head_stride;
    int32_t b, seqlen_new, seqlen_cache, h_k, d;
    int32_t reserved;
    const int32_t *cache_seqlens;   /* (b) int32, rows already valid in each cache entry */
    const int32_t *cache_batch_idx; /* (b) int32 or NULL */
    const int32_t *block_table;     /* paged cache (see fa_fwd_params) or NULL; then seqlen_cache = pages per sequence
                                       x page_block_size and k/vcache_batch_stride is the page stride */
    int64_t block_table_batch_stride;
    int32_t page_block_size;
    int32_t dtype; /* enum fa_dtype (16-bit types); only read when rotary_cos is set */
    /* Rotary embedding of the appended KEYS (csrc/flash_attn/flash_api.cpp:1404-1428, src/flash_fwd_kernel.h:679-735):
     * row i of k_new is rotated by position cache_seqlens[b] + i before it is stored; values are copied unchanged.
     * rotary_cos / rotary_sin: (seqlen_ro, rotary_dim / 2), contiguous, same 16-bit dtype as k; rotary_dim % 16 == 0,
     * <= d; interleaved: pairs (2j, 2j+1) (GPT-J), otherwise (j, j + rotary_dim/2) (GPT-NeoX).  NULL = no rotary. */
    const void *rotary_cos;
    const void *rotary_sin;
    int32_t rotary_dim;
    int32_t rotary_interleaved;
    /* ABI v12 -- FA3 `seqlens_rotary` (hopper/flash_api.cpp:1074-1079, hopper/seqlen.h:89): (b) int32 rotary position of the
     * first appended row of each batch entry when it is not the cache fill level; NULL = cache_seqlens. */
    const int32_t *rotary_seqlens;
} fa_kvcache_append_params;

int fa_kvcache_append(const fa_kvcache_append_params *params, void *stream);
uint32_t fa_kvcache_append_params_size(void);

/*
 * Rotary embedding of a (b, s, h, d) tensor into `dst` (same shape; dst may alias src): row i of batch b is rotated by
 * position seqlen_offsets[b] + (per_row_positions ? i : 0) -- the query side of mha_fwd_kvcache (causal / local: one
 * position per query row; otherwise every row at cache_seqlens, flash_attn/flash_attn_interface.py:1516-1524).
 */
typedef struct fa_rotary_params {
    uint32_t abi_version;
    uint32_t struct_size;
    const void *src;
    void *dst;
    int64_t src_batch_stride, src_row_stride, src_head_stride;
    int64_t dst_batch_stride, dst_row_stride, dst_head_stride;
    int32_t b, s, h, d;
    int32_t dtype; /* FA_DTYPE_FP16 / FA_DTYPE_BF16 */
    int32_t rotary_dim;
    int32_t rotary_interleaved;
    int32_t per_row_positions;
    const void *rotary_cos; /* (seqlen_ro, rotary_dim / 2) */
    const void *rotary_sin;
    const int32_t *seqlen_offsets; /* (b) */
} fa_rotary_params;

int fa_rotary_apply(const fa_rotary_params *params, void *stream);
uint32_t fa_rotary_params_size(void);

/* Merge of split-KV partial results given by the caller: mha_combine / flash_attn_3::fwd_combine
 * (hopper/flash_api.cpp:1569-1670, hopper/flash_fwd_combine_kernel.h).
 *   lse[b, i, h] = log sum_s exp(lse_partial[s, b, i, h]);   out[b, i, h, :] = sum_s exp(lse_partial[s] - lse) out_partial[s]
 * A split with lse_partial = -inf carries no weight; rows where every split is -inf give out = 0, lse = -inf
 * (attention_combine_ref, hopper/test_flash_attn.py:1105-1114).  Partials are fp32 with arbitrary element strides
 * (head-dim stride 1); out is fp32 / fp16 / bf16.  num_splits <= 256 like the reference. */
typedef struct fa_combine_params {
    uint32_t abi_version; /* FA_ABI_VERSION */
    uint32_t struct_size; /* sizeof(fa_combine_params) */
    const float *out_partial; /* (num_splits, b, seqlen, h, d) */
    const float *lse_partial; /* (num_splits, b, seqlen, h) */
    void *out;                /* (b, seqlen, h, d) of out_dtype */
    float *softmax_lse;       /* (b, seqlen, h) through the strides below */
    int64_t op_split_stride, op_batch_stride, op_row_stride, op_head_stride;
    int64_t lp_split_stride, lp_batch_stride, lp_row_stride, lp_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    int64_t lse_batch_stride, lse_row_stride, lse_head_stride;
    int32_t num_splits, b, seqlen, h, d;
    int32_t out_dtype; /* FA_DTYPE_FP32 / FA_DTYPE_FP16 / FA_DTYPE_BF16 */
} fa_combine_params;

int fa_fwd_combine(const fa_combine_params *params, void *stream);
uint32_t fa_combine_params_size(void);

/* Test hook: overrides the default kernel variant process-wide (0 = default). */
void fa_set_default_variant(int32_t variant);
/* Test hook: the persistent form of the 256-row kernel (one workgroup per CU walking a chain of work items):
 * 0 = the library's choice, -1 = never, 1 = for every problem it can run (also chains of a single item). */
void fa_set_persist_mode(int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* FA_FWD_H_ */
