// Probe: semantics of the packed fp32 VALU forms the generated backward loops use (op_sel / neg on v_pk_fma_f32, v_pk_add_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(float *out, float c) {
    float s0 = 1.5f + threadIdx.x, s1 = 2.5f + threadIdx.x;
    float l0 = 10.f, l1 = 20.f, d0 = 100.f, d1 = 200.f, p0 = 3.f, p1 = 4.f;
    const uint64_t c2 = (uint64_t)__float_as_uint(c);
    float a0, a1, b0, b1, e0, e1, f0, f1, g0, g1;
    asm volatile(
        "v_mov_b32 v0, %10\n\tv_mov_b32 v1, %11\n\tv_mov_b32 v220, %12\n\tv_mov_b32 v221, %13\n\t"
        "v_mov_b32 v222, %14\n\tv_mov_b32 v223, %15\n\tv_mov_b32 v32, %16\n\tv_mov_b32 v33, %17\n\t"
        "v_mov_b32 v2, v0\n\tv_mov_b32 v3, v1\n\t"
        "v_pk_fma_f32 v[0:1], v[0:1], %18, v[220:221] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_pk_fma_f32 v[2:3], v[2:3], %18, v[220:221] op_sel:[0,0,1] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_mov_b32 v34, v32\n\tv_mov_b32 v35, v33\n\t"
        "v_pk_add_f32 v[32:33], v[32:33], v[222:223] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 v[34:35], v[34:35], v[222:223] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 v[36:37], v[0:1], v[32:33]\n\t"
        "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3\n\t"
        "v_mov_b32 %4, v32\n\tv_mov_b32 %5, v33\n\tv_mov_b32 %6, v34\n\tv_mov_b32 %7, v35\n\tv_mov_b32 %8, v36\n\tv_mov_b32 %9, v37\n\t"
        : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1), "=v"(e0), "=v"(e1), "=v"(f0), "=v"(f1), "=v"(g0), "=v"(g1)
        : "v"(s0), "v"(s1), "v"(l0), "v"(l1), "v"(d0), "v"(d1), "v"(p0), "v"(p1), "s"(c2)
        : "v0", "v1", "v2", "v3", "v32", "v33", "v34", "v35", "v36", "v37", "v220", "v221", "v222", "v223");
    if (threadIdx.x == 1) {
        out[0] = a0; out[1] = a1; out[2] = b0; out[3] = b1; out[4] = e0; out[5] = e1; out[6] = f0; out[7] = f1; out[8] = g0; out[9] = g1;
    }
}
int main() {
    float *d; hipMalloc(&d, 64); k<<<1, 64>>>(d, 2.f); float h[10]; hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    // lane 1: s0 = 2.5, s1 = 3.5, c = 2
    printf("fma lo-sel : %g %g   (expect 2.5*2-10 = -5, 3.5*2-10 = -3)\n", h[0], h[1]);
    printf("fma hi-sel : %g %g   (expect 2.5*2-20 = -15, 3.5*2-20 = -13)\n", h[2], h[3]);
    printf("add lo-sel : %g %g   (expect 3-100 = -97, 4-100 = -96)\n", h[4], h[5]);
    printf("add hi-sel : %g %g   (expect 3-200 = -197, 4-200 = -196)\n", h[6], h[7]);
    printf("mul        : %g %g   (expect -5*-97 = 485, -3*-96 = 288)\n", h[8], h[9]);
    return 0;
}
