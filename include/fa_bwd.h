/*
 * fa_bwd.h — C-ABI of the MI355X (gfx950) FlashAttention backward.
 *
 *   reference interface replaced                                  entry point here
 *   ------------------------------------------------------------  -----------------
 *   mha_bwd          csrc/flash_attn/flash_api.cpp:767-971        fa_bwd (dense)
 *   mha_varlen_bwd   csrc/flash_attn/flash_api.cpp:973-1200       fa_bwd (cu_seqlens_* set)
 *   set_params_dgrad csrc/flash_attn/flash_api.cpp:161-221        fa_bwd_params
 *   Flash_bwd_params csrc/flash_attn/src/flash.h:147-191          fa_bwd_params
 *
 * Same conventions as fa_fwd.h: raw device pointers, ELEMENT strides, head-dim stride 1, the callee never
 * allocates or synchronises and launches on `stream`.  Three launches: D = rowsum(dO * O) (role of
 * compute_dot_do_o, csrc/flash_attn/src/flash_bwd_preprocess_kernel.h:60-127), then dK/dV (one workgroup per
 * 128-key block of a kv head, looping over the query tiles of every query head of its GQA group) and dQ (one
 * workgroup per 128-row query block, looping over the key tiles).  Each output element is produced by exactly
 * one workgroup in a fixed order: no atomics, bit-reproducible ("deterministic" is always on); the price is
 * that S and dP are recomputed by the dQ pass (7 matrix products instead of the reference's 5).
 */
#ifndef FA_BWD_H_
#define FA_BWD_H_

#include <stdint.h>

#include "fa_fwd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fa_bwd_params {
    uint32_t abi_version; /* FA_ABI_VERSION */
    uint32_t struct_size; /* sizeof(fa_bwd_params) */

    /* inputs (device pointers): the forward's operands and results, and the incoming gradient */
    const void *q;
    const void *k;
    const void *v;
    const void *o;
    const void *dout;
    const float *softmax_lse; /* as written by fa_fwd: (b, h, seqlen_q) or varlen (h, total_q) */
    /* outputs */
    void *dq; /* like q */
    void *dk; /* like k: (b, seqlen_k, h_k, d) -- already summed over the query heads of each GQA group */
    void *dv; /* like v */
    float *softmax_d; /* D = rowsum(dO * O): dense (b, h, softmax_d_row_len), varlen (h, softmax_d_row_len) */

    /* element strides; head_dim stride is 1 */
    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    int64_t do_batch_stride, do_row_stride, do_head_stride;
    int64_t dq_batch_stride, dq_row_stride, dq_head_stride;
    int64_t dk_batch_stride, dk_row_stride, dk_head_stride;
    int64_t dv_batch_stride, dv_row_stride, dv_head_stride;
    int64_t softmax_d_row_len; /* >= seqlen_q (dense; the reference rounds to 128) / >= total_q (varlen) */

    /* sizes: as fa_fwd_params */
    int32_t b, seqlen_q, seqlen_k, h, h_k, d, total_q, total_k;
    int32_t dtype; /* FA_DTYPE_FP16 / FA_DTYPE_BF16 */

    const int32_t *cu_seqlens_q; /* (b+1) or NULL = dense */
    const int32_t *cu_seqlens_k;

    float softmax_scale;
    float softcap;
    int32_t is_causal;
    int32_t window_size_left;
    int32_t window_size_right;

    const float *alibi_slopes; /* as fa_fwd_params */
    int64_t alibi_slopes_batch_stride;

    int32_t deterministic; /* accepted; results are always bit-reproducible */
    float p_dropout;       /* as fa_fwd_params; rng_state must be the pair the forward used */
    const uint64_t *rng_state;
    int32_t flags;         /* FA_FLAG_* as fa_fwd_params (FA_FLAG_FA3_WINDOW) */
    /* ABI v12 -- head dim of v / o / dout / dv when it differs from d (FA3 headdim_v, hopper/flash_api.cpp:1345-1369:
     * the reference's backward rounds both to the larger): 0 = d.  Built for the wide head-dim tile only (max(d, d_v) in
     * (128, 256], e.g. 192 / 128); other pairs return FA_ERR_UNSUPPORTED. */
    int32_t d_v;
} fa_bwd_params;

/* Validate and enqueue the backward on `stream`.  Returns FA_OK or a negative fa_status; asynchronous. */
int fa_bwd(const fa_bwd_params *params, void *stream);

/* Validation only (what mha_bwd's TORCH_CHECKs do); no device access. */
int fa_bwd_validate(const fa_bwd_params *params);

/* sizeof(fa_bwd_params) as compiled, for binding self-checks. */
uint32_t fa_bwd_params_size(void);

#ifdef __cplusplus
}
#endif
#endif /* FA_BWD_H_ */
