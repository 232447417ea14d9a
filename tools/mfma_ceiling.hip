// tools/mfma_ceiling.hip — what the gfx950 matrix pipe delivers in isolation (developer aid, not product code).
//
// Bare MFMA loops, one wave per SIMD (256 workgroups x 256 threads), operands in registers, random data, for both bf16
// shapes (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16), the non-scaled fp8 forms and the block-scaled
// f8f6f4 forms with e4m3 operands, each with and without the softmax's VALU mix between the MFMAs.  Reports TFLOP/s,
// cycles per MFMA and the in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz, median over workgroups) after
// >= 2 s of back-to-back launches (MI355X_MICROARCH.md "DVFS give-back" item 6).  This is the denominator DESIGN.md
// quotes next to the 2.5 PF spec peak: the chip lowers its clock under MFMA load, so spec peak is not reachable on
// random data by any instruction stream.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two bf16 values, roughly N(0,1): sign random, exponent 2^-2..2^1 weighted, random mantissa
__device__ __forceinline__ uint32_t rand_bf16x2(uint32_t seed) {
    const uint32_t h = hash32(seed);
    auto one = [](uint32_t b) -> uint32_t {  // 16 random bits -> bf16 bits
        const uint32_t sign = (b >> 15) & 1, e = 125 + ((b >> 12) & 3), man = b & 0x7f;
        return (sign << 15) | (e << 7) | man;
    };
    return one(h & 0xffff) | (one(h >> 16) << 16);
}
// four e4m3 values, magnitudes 2^-2..2^1
__device__ __forceinline__ uint32_t rand_fp8x4(uint32_t seed) {
    const uint32_t h = hash32(seed);
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = (h >> (8 * i)) & 0xff;
        const uint32_t sign = b >> 7, e = 5 + ((b >> 3) & 3), man = b & 7;
        r |= ((sign << 7) | (e << 3) | man) << (8 * i);
    }
    return r;
}

// VALU mix of the attention loop at head dim 128: one pair of scores per 64 MFMA-cycles of the bf16 forms (per 32 of the
// block-scaled fp8 forms, which do the same products in half the cycles).  FILL = 1: the softmax without the scale/max fma
// (exp exp | add add cvt_pk); FILL = 2: the round-1 mix (fma fma exp exp | add add cvt_pk).  Front and back halves go
// behind different MFMAs.
template <int FILL>
__device__ __forceinline__ void fill_front(float &t0, float &t1, float s0, float s1, float c, float mc) {
    if constexpr (FILL == 2) asm volatile("v_fma_f32 %0, %2, %4, -%5\n\tv_fma_f32 %1, %3, %4, -%5\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1"
                                          : "=&v"(t0), "=&v"(t1) : "v"(s0), "v"(s1), "s"(c), "v"(mc));
    if constexpr (FILL == 1) asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(t0), "=&v"(t1) : "v"(s0), "v"(s1));
}
template <int FILL>
__device__ __forceinline__ void fill_back(uint32_t &pk, float &ps0, float &ps1, float t0, float t1) {
    if constexpr (FILL != 0) asm volatile("v_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %4\n\tv_cvt_pk_bf16_f32 %0, %3, %4"
                                          : "=&v"(pk), "+v"(ps0), "+v"(ps1) : "v"(t0), "v"(t1));
}

enum Shape { BF16_32 = 0, BF16_16 = 1, FP8_32 = 2, FP8_16 = 3, MX8_32 = 4, MX8_16 = 5 };

template <int SHAPE> struct Traits;
template <> struct Traits<BF16_32> { static constexpr int NACC = 4, PER_ITER = 16; static constexpr double FLOP = 2.0 * 32 * 32 * 16; static constexpr int CYC = 32; static constexpr const char *name = "bf16 32x32x16"; };
template <> struct Traits<BF16_16> { static constexpr int NACC = 16, PER_ITER = 32; static constexpr double FLOP = 2.0 * 16 * 16 * 32; static constexpr int CYC = 16; static constexpr const char *name = "bf16 16x16x32"; };
template <> struct Traits<FP8_32> { static constexpr int NACC = 4, PER_ITER = 16; static constexpr double FLOP = 2.0 * 32 * 32 * 16; static constexpr int CYC = 32; static constexpr const char *name = "fp8  32x32x16 (non-scaled)"; };
template <> struct Traits<FP8_16> { static constexpr int NACC = 16, PER_ITER = 32; static constexpr double FLOP = 2.0 * 16 * 16 * 32; static constexpr int CYC = 16; static constexpr const char *name = "fp8  16x16x32 (non-scaled)"; };
template <> struct Traits<MX8_32> { static constexpr int NACC = 4, PER_ITER = 16; static constexpr double FLOP = 2.0 * 32 * 32 * 64; static constexpr int CYC = 64; static constexpr const char *name = "e4m3 32x32x64 f8f6f4 (scaled)"; };
template <> struct Traits<MX8_16> { static constexpr int NACC = 16, PER_ITER = 32; static constexpr double FLOP = 2.0 * 16 * 16 * 128; static constexpr int CYC = 32; static constexpr const char *name = "e4m3 16x16x128 f8f6f4 (scaled)"; };

// One kernel per (shape, fill).  Output tile per wave is the same for every shape: 64 x 64 (2x2 tiles of 32x32 or 4x4 of
// 16x16), two k-steps per iteration with distinct operand fragments.
template <int SHAPE> struct Frag { typedef u32x4 type; };
template <> struct Frag<FP8_32> { typedef u32x2 type; };
template <> struct Frag<FP8_16> { typedef u32x2 type; };
template <> struct Frag<MX8_32> { typedef u32x8 type; };
template <> struct Frag<MX8_16> { typedef u32x8 type; };

template <int SHAPE, typename ACC, typename FRAG>
__device__ __forceinline__ void mfma(ACC &acc, const FRAG &a, const FRAG &b) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass would try to match "v" against x86 vector registers)
    const uint32_t one = 0x7f7f7f7fu;  // E8M0 scale 2^0 for every block
    if constexpr (SHAPE == BF16_32) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if constexpr (SHAPE == FP8_32) asm volatile("v_mfma_f32_32x32x16_fp8_fp8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if constexpr (SHAPE == MX8_32) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(one));
    if constexpr (SHAPE == BF16_16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if constexpr (SHAPE == FP8_16) asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if constexpr (SHAPE == MX8_16) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(one));
#endif
}

template <int SHAPE, int FILL>
__global__ __launch_bounds__(256, 1) void bare(unsigned long long *stamps, float *sink, int iters) {
    using Tr = Traits<SHAPE>;
    using frag = typename Frag<SHAPE>::type;
    constexpr bool BIG = (SHAPE == BF16_32 || SHAPE == FP8_32 || SHAPE == MX8_32);
    constexpr int NT = BIG ? 2 : 4;               // tiles per side of the 64 x 64 output
    constexpr int W = sizeof(frag) / 4;
    using acc_t = typename std::conditional<BIG, f32x16, f32x4>::type;
    const uint32_t lane_seed = (blockIdx.x * 256 + threadIdx.x) * 128u;
    float ps0 = 0.f, ps1 = 0.f, mc = 3.0f;
    float sv[8];
    for (int i = 0; i < 8; ++i) sv[i] = (float)((hash32(lane_seed + 1000 + i) & 0xffff)) * (1.f / 16384.f) - 2.f;
    uint32_t pk = 0;
    float t0 = 0.f, t1 = 0.f;
    unsigned long long c0 = 0, r0 = 0, c1, r1;
    frag a[2][NT], b[2][NT];  // [k-step][tile]
    for (int ks = 0; ks < 2; ++ks)
        for (int t = 0; t < NT; ++t)
            for (int w = 0; w < W; ++w) {
                const uint32_t sd = lane_seed + ks * 64 + t * 8 + w;
                a[ks][t][w] = (SHAPE == BF16_32 || SHAPE == BF16_16) ? rand_bf16x2(sd) : rand_fp8x4(sd);
                b[ks][t][w] = (SHAPE == BF16_32 || SHAPE == BF16_16) ? rand_bf16x2(sd + 32) : rand_fp8x4(sd + 32);
            }
    acc_t acc[NT][NT] = {};
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
    for (int it = 0; it < iters; ++it) {
        int n = 0;  // MFMA index inside the iteration (compile-time after unrolling)
#pragma unroll
        for (int rep = 0; rep < (BIG ? 2 : 1); ++rep)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j, ++n) {
                        mfma<SHAPE>(acc[i][j], a[ks][i], b[ks][j]);
                        const int e = n & 7;
                        // one pair of scores per PERIOD MFMA-cycles: 64 for the bf16-rate forms, 32 for the block-scaled ones
                        constexpr int PERIOD = (SHAPE == MX8_32 || SHAPE == MX8_16) ? 32 : 64;
                        constexpr int PER = PERIOD / Tr::CYC;  // MFMAs per pair: 4, 2, 1, or 0 (= two pairs per MFMA)
                        if constexpr (PER == 0) {
                            fill_front<FILL>(t0, t1, sv[e], sv[(e + 1) & 7], 0.125f, mc);
                            fill_back<FILL>(pk, ps0, ps1, t0, t1);
                            fill_front<FILL>(t0, t1, sv[(e + 2) & 7], sv[(e + 3) & 7], 0.125f, mc);
                            fill_back<FILL>(pk, ps0, ps1, t0, t1);
                        } else if constexpr (PER == 1) {
                            fill_front<FILL>(t0, t1, sv[e], sv[(e + 1) & 7], 0.125f, mc);
                            fill_back<FILL>(pk, ps0, ps1, t0, t1);
                        } else {
                            if (n % PER == 0) fill_front<FILL>(t0, t1, sv[e], sv[(e + 1) & 7], 0.125f, mc);
                            if (n % PER == PER / 2) fill_back<FILL>(pk, ps0, ps1, t0, t1);
                        }
                    }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
    float s = ps0 + ps1 + __builtin_bit_cast(float, pk);
    for (int i = 0; i < NT; ++i) for (int j = 0; j < NT; ++j) for (int e = 0; e < (BIG ? 16 : 4); ++e) s += acc[i][j][e];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int FILL>
void run(unsigned long long *d_stamps, float *d_sink, int num_cu) {
    using Tr = Traits<SHAPE>;
    const int iters = 8192 * 32 / Tr::CYC * 16 / Tr::PER_ITER;  // ~ 4.2 M MFMA-cycles per launch
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // >= 2 s of back-to-back launches, then the timed batch
    float warm_ms = 0.f;
    int launches = 0;
    while (warm_ms < 2000.f) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) bare<SHAPE, FILL><<<num_cu, 256>>>(d_stamps, d_sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        warm_ms += ms;
        launches += 20;
    }
    const int N = 40;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) bare<SHAPE, FILL><<<num_cu, 256>>>(d_stamps, d_sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * num_cu);
    CHECK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int i = 0; i < num_cu; ++i) {
        clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);  // GHz
        cyc.push_back((double)st[2 * i]);
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double mfmas = (double)iters * Tr::PER_ITER;
    const double flops = (double)num_cu * 4 * mfmas * Tr::FLOP;
    const double tf = flops / (ms / N * 1e-3) * 1e-12;
    printf("%-32s fill=%d  %8.1f TFLOP/s  %6.2f cyc/MFMA (floor %d)  clock %.3f GHz  (%.3f ms/launch)\n", Tr::name, FILL, tf,
           cyc[num_cu / 2] / mfmas, Tr::CYC, clk[num_cu / 2], ms / N);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int num_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs; fill: 0 = bare MFMA, 1 = + (exp exp add add cvt_pk) per score pair, 2 = + fma fma too (the attention loop VALU mix at d128)\n", prop.name, num_cu);
    unsigned long long *d_stamps;
    float *d_sink;
    CHECK(hipMalloc(&d_stamps, 2 * num_cu * 8));
    CHECK(hipMalloc(&d_sink, 1024));
    run<BF16_32, 0>(d_stamps, d_sink, num_cu);
    run<BF16_16, 0>(d_stamps, d_sink, num_cu);
    run<BF16_32, 1>(d_stamps, d_sink, num_cu);
    run<BF16_16, 1>(d_stamps, d_sink, num_cu);
    run<BF16_32, 2>(d_stamps, d_sink, num_cu);
    run<BF16_16, 2>(d_stamps, d_sink, num_cu);
    run<FP8_32, 0>(d_stamps, d_sink, num_cu);
    run<FP8_16, 0>(d_stamps, d_sink, num_cu);
    run<MX8_32, 0>(d_stamps, d_sink, num_cu);
    run<MX8_16, 0>(d_stamps, d_sink, num_cu);
    run<MX8_16, 1>(d_stamps, d_sink, num_cu);
    run<BF16_32, 0>(d_stamps, d_sink, num_cu);  // again: drift check
    return 0;
}
