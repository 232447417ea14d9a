// Probe: do the packed fp32 VALU ops (v_pk_fma/add/mul_f32) need wait states before / after their neighbours in the sequences
// the generated backward loops use?  Each case runs the tight sequence and the same sequence padded with s_nop 7 between all
// instructions, on per-lane data, many times, and counts bitwise mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N7 "s_nop 7\n\t"
__global__ void k(unsigned *bad, int iters) {
    const int lane = threadIdx.x;
    unsigned nb[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        float s0 = 0.37f * lane + 0.011f * it, s1 = -0.21f * lane + 0.5f + 0.007f * it;
        float l0 = 1.25f + 0.01f * lane, d0 = 0.3f - 0.02f * lane, p0 = 0.11f * lane - 1.f, p1 = 0.05f * lane + 0.3f;
        const uint64_t c2 = (uint64_t)__float_as_uint(0.1275f);
        uint32_t r[6], q[6];
        // tight
        asm volatile(
            "v_mov_b32 v0, %6\n\tv_mov_b32 v1, %7\n\tv_mov_b32 v220, %8\n\tv_mov_b32 v221, %8\n\tv_mov_b32 v222, %9\n\tv_mov_b32 v223, %9\n\t"
            "v_mov_b32 v32, %10\n\tv_mov_b32 v33, %11\n\t" N7
            "v_pk_fma_f32 v[0:1], v[0:1], %12, v[220:221] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
            "v_exp_f32 v0, v0\n\t"
            "v_exp_f32 v1, v1\n\t"
            "v_pk_add_f32 v[32:33], v[32:33], v[222:223] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_mul_f32 v[32:33], v[0:1], v[32:33]\n\t"
            "v_cvt_pk_bf16_f32 v128, v32, v33\n\t"
            "v_cvt_f16_f32 v129, v32\n\t"
            "v_cvt_f16_f32_sdwa v129, v33 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
            N7
            "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v32\n\tv_mov_b32 %3, v33\n\tv_mov_b32 %4, v128\n\tv_mov_b32 %5, v129\n\t"
            : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5])
            : "v"(s0), "v"(s1), "v"(l0), "v"(d0), "v"(p0), "v"(p1), "s"(c2)
            : "v0", "v1", "v32", "v33", "v128", "v129", "v220", "v221", "v222", "v223");
        // padded
        asm volatile(
            "v_mov_b32 v0, %6\n\tv_mov_b32 v1, %7\n\tv_mov_b32 v220, %8\n\tv_mov_b32 v221, %8\n\tv_mov_b32 v222, %9\n\tv_mov_b32 v223, %9\n\t"
            "v_mov_b32 v32, %10\n\tv_mov_b32 v33, %11\n\t" N7
            "v_pk_fma_f32 v[0:1], v[0:1], %12, v[220:221] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t" N7
            "v_exp_f32 v0, v0\n\t" N7
            "v_exp_f32 v1, v1\n\t" N7
            "v_pk_add_f32 v[32:33], v[32:33], v[222:223] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" N7
            "v_pk_mul_f32 v[32:33], v[0:1], v[32:33]\n\t" N7
            "v_cvt_pk_bf16_f32 v128, v32, v33\n\t" N7
            "v_cvt_f16_f32 v129, v32\n\t" N7
            "v_cvt_f16_f32_sdwa v129, v33 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
            N7
            "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v32\n\tv_mov_b32 %3, v33\n\tv_mov_b32 %4, v128\n\tv_mov_b32 %5, v129\n\t"
            : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]), "=v"(q[4]), "=v"(q[5])
            : "v"(s0), "v"(s1), "v"(l0), "v"(d0), "v"(p0), "v"(p1), "s"(c2)
            : "v0", "v1", "v32", "v33", "v128", "v129", "v220", "v221", "v222", "v223");
        for (int i = 0; i < 6; ++i) nb[i] += r[i] != q[i];
    }
    for (int i = 0; i < 6; ++i) atomicAdd(&bad[i], nb[i]);
}
int main() {
    unsigned *d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    k<<<256, 64>>>(d, 2000);
    unsigned h[6]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("mismatches tight vs padded (256 waves x 2000 iterations x 64 lanes):\n");
    printf("  exp(pk_fma lo) %u  exp(pk_fma hi) %u  pk_mul lo %u  pk_mul hi %u  cvt_pk_bf16 %u  cvt_f16+sdwa %u\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
