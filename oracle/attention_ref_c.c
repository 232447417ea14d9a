/*
 * ORACLE — test infrastructure only; never linked into or called by the product path.
 *
 * Plain-C, double-precision restatement of exact softmax attention, independent of torch:
 *   scores = softmax_scale * q.k^T  (optionally softcap*tanh(./softcap))      tests/test_util.py:219-228
 *   mask   : key j visible from query i iff                                    tests/test_util.py:150-182
 *            max(0, i + sk - sq - left) <= j < min(sk, i + sk - sq + right + 1)   (bottom-right aligned;
 *            left/right < 0 = unbounded; causal => right = 0)                  csrc/flash_attn/src/mask.h:156-186
 *   out    = softmax(scores) . v ; rows with no visible key -> 0               tests/test_util.py:249-252
 *   lse    = log sum exp(scores) ; +inf for such rows                          csrc/flash_attn/src/softmax.h:178-180
 *   GQA    : kv head = q head / (h / h_k)                                      csrc/flash_attn/src/flash_fwd_kernel.h:148
 * Layout: q (b,sq,h,d), k/v (b,sk,h_k,d), out (b,sq,h,d), lse (b,h,sq); all contiguous fp32 in, fp64 math.
 * Used by tests/ to cross-check the torch restatement (oracle/attention_ref.py) and the HIP path.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

int fa_oracle_attention_f32(const float *q, const float *k, const float *v, float *out, float *lse,
                            int b, int sq, int sk, int h, int h_k, int d, float softmax_scale,
                            int is_causal, int window_left, int window_right, float softcap) {
    if (b <= 0 || h <= 0 || h_k <= 0 || h % h_k != 0 || d <= 0) return -1;
    if (is_causal) window_right = 0;
    const int g = h / h_k;
    double *p = (double *)malloc(sizeof(double) * (size_t)(sk > 0 ? sk : 1));
    double *acc = (double *)malloc(sizeof(double) * (size_t)d);
    if (!p || !acc) { free(p); free(acc); return -2; }
    for (int bi = 0; bi < b; ++bi)
        for (int hi = 0; hi < h; ++hi) {
            const int hk = hi / g;
            for (int i = 0; i < sq; ++i) {
                const float *qr = q + (((size_t)bi * sq + i) * h + hi) * d;
                float *orow = out + (((size_t)bi * sq + i) * h + hi) * d;
                long lo = 0, hi_excl = sk;
                const long diag = (long)i + sk - sq;
                if (window_right >= 0 && diag + window_right + 1 < hi_excl) hi_excl = diag + window_right + 1;
                if (window_left >= 0 && diag - window_left > lo) lo = diag - window_left;
                double m = -INFINITY;
                for (long j = lo; j < hi_excl; ++j) {
                    const float *kr = k + (((size_t)bi * sk + j) * h_k + hk) * d;
                    double s = 0.0;
                    for (int t = 0; t < d; ++t) s += (double)qr[t] * (double)kr[t];
                    s *= (double)softmax_scale;
                    if (softcap > 0.f) s = (double)softcap * tanh(s / (double)softcap);
                    p[j] = s;
                    if (s > m) m = s;
                }
                for (int t = 0; t < d; ++t) acc[t] = 0.0;
                double l = 0.0;
                for (long j = lo; j < hi_excl; ++j) {
                    const double e = exp(p[j] - m);
                    l += e;
                    const float *vr = v + (((size_t)bi * sk + j) * h_k + hk) * d;
                    for (int t = 0; t < d; ++t) acc[t] += e * (double)vr[t];
                }
                if (hi_excl <= lo || l == 0.0) {
                    for (int t = 0; t < d; ++t) orow[t] = 0.f;
                    if (lse) lse[((size_t)bi * h + hi) * sq + i] = INFINITY;
                } else {
                    for (int t = 0; t < d; ++t) orow[t] = (float)(acc[t] / l);
                    if (lse) lse[((size_t)bi * h + hi) * sq + i] = (float)(m + log(l));
                }
            }
        }
    free(p);
    free(acc);
    return 0;
}
