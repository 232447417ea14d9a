// tools/probe_layouts.hip — operand lane maps of the gfx950 instructions the forward kernels rely on, measured with
// one-hot operands (developer aid).  The guide documents the bf16 maps; the fp8 / block-scaled ones and the 8-bit
// transposing LDS read are "check with exact integer data before relying on it" — this is that check, and its output is
// kept in profiles/ next to the kernels that depend on it.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe_layouts.hip -o /tmp/probe_layouts && /tmp/probe_layouts
//
// For every MFMA form: A one-hot at (lane L, element j) x B all-ones gives the ROW of that A element; B one-hot x A
// all-ones gives the COLUMN of a B element; A one-hot x B one-hot over all (L', j') gives the B elements that share its k.
// For the transposing reads: every lane points at its own 64-byte LDS region, so each destination byte names its source
// (lane, byte).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum Kind { BF16_16 = 0, FP8_16 = 1, MX8_16 = 2 };
template <int KIND> struct K;
template <> struct K<BF16_16> { static constexpr int NE = 8, EB = 2; static constexpr uint32_t ONE = 0x3f80; typedef u32x4 frag; static constexpr const char *name = "v_mfma_f32_16x16x32_bf16"; };
template <> struct K<FP8_16> { static constexpr int NE = 8, EB = 1; static constexpr uint32_t ONE = 0x38; typedef u32x2 frag; static constexpr const char *name = "v_mfma_f32_16x16x32_fp8_fp8"; };
template <> struct K<MX8_16> { static constexpr int NE = 32, EB = 1; static constexpr uint32_t ONE = 0x38; typedef u32x8 frag; static constexpr const char *name = "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, unit scales)"; };

template <int KIND>
__device__ __forceinline__ f32x4 mfma(typename K<KIND>::frag a, typename K<KIND>::frag b) {
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t one = 0x7f7f7f7fu;
    if constexpr (KIND == BF16_16) asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "=&v"(d) : "v"(a), "v"(b));
    if constexpr (KIND == FP8_16) asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "=&v"(d) : "v"(a), "v"(b));
    if constexpr (KIND == MX8_16) asm volatile("s_nop 7\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "=&v"(d) : "v"(a), "v"(b), "v"(one));
#endif
    return d;
}

template <int KIND>
__device__ __forceinline__ typename K<KIND>::frag fill(bool all_ones, bool hot, int j) {
    typename K<KIND>::frag f;
    constexpr int W = sizeof(f) / 4, PER = 4 / K<KIND>::EB;
    for (int w = 0; w < W; ++w) {
        uint32_t x = 0;
        for (int e = 0; e < PER; ++e) {
            const bool on = all_ones || (hot && (w * PER + e) == j);
            if (on) x |= K<KIND>::ONE << (8 * K<KIND>::EB * e);
        }
        f[w] = x;
    }
    return f;
}

// out[(L * NE + j) * 4 + {0: row of A(L,j), 1: col of B(L,j), 2: B lane' with lane'&15 == 0 sharing k with A(L,j) ... + 64 * j', 3: #matches}]
template <int KIND>
__global__ void probe_mfma(int *out) {
    constexpr int NE = K<KIND>::NE;
    const int L = blockIdx.x, lane = threadIdx.x;
    for (int j = 0; j < NE; ++j) {
        int *o = out + (L * NE + j) * 4;
        {   // row of A(L, j)
            const f32x4 d = mfma<KIND>(fill<KIND>(false, lane == L, j), fill<KIND>(true, false, 0));
            for (int r = 0; r < 4; ++r)
                if (d[r] != 0.f && (lane & 15) == 0) o[0] = 4 * (lane >> 4) + r;
        }
        {   // column of B(L, j)
            const f32x4 d = mfma<KIND>(fill<KIND>(true, false, 0), fill<KIND>(false, lane == L, j));
            if (d[0] != 0.f && lane < 16) o[1] = lane;
        }
        int matches = 0, first = -1;
        for (int L2 = 0; L2 < 64; ++L2)
            for (int j2 = 0; j2 < NE; ++j2) {
                const f32x4 d = mfma<KIND>(fill<KIND>(false, lane == L, j), fill<KIND>(false, lane == L2, j2));
                const bool nz = d[0] != 0.f || d[1] != 0.f || d[2] != 0.f || d[3] != 0.f;
                if (__any(nz)) {
                    ++matches;
                    if ((L2 & 15) == 0) first = L2 + 64 * j2;
                }
            }
        if (lane == 0) { o[2] = first; o[3] = matches; }
    }
}

// each lane points at its own 64-byte region; pass 0: byte = lane id of the region, pass 1: byte = offset inside the region
template <int BITS>
__global__ void probe_tr(uint32_t *out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 64];
    const int lane = threadIdx.x;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = lane; i < 64 * 64; i += 64) lds[i] = pass == 0 ? (i >> 6) : (i & 63);
        __syncthreads();
        u32x2 r;
        if constexpr (BITS == 8) {
            const i32x2 t = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2 *)(lds + 64 * lane));
            r = __builtin_bit_cast(u32x2, t);
        } else {
            const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + 64 * lane));
            r = __builtin_bit_cast(u32x2, t);
        }
        out[(pass * 64 + lane) * 2] = r[0];
        out[(pass * 64 + lane) * 2 + 1] = r[1];
        __syncthreads();
    }
}

template <int KIND>
void run_mfma() {
    constexpr int NE = K<KIND>::NE;
    int *d;
    CHECK(hipMalloc(&d, 64 * NE * 4 * sizeof(int)));
    CHECK(hipMemset(d, 0xff, 64 * NE * 4 * sizeof(int)));
    probe_mfma<KIND><<<64, 64>>>(d);
    CHECK(hipDeviceSynchronize());
    std::vector<int> h(64 * NE * 4);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost));
    printf("== %s: lane L element j ->  A row | B col | k (named after the B element of lane 16g', element j': k = %d g' + j') | #B elements sharing k\n", K<KIND>::name, NE);
    int bad_row = 0, bad_col = 0, bad_k = 0, bad_n = 0;
    for (int L = 0; L < 64; ++L)
        for (int j = 0; j < NE; ++j) {
            const int *o = &h[(L * NE + j) * 4];
            const int k = o[2] < 0 ? -1 : NE * ((o[2] & 63) >> 4) + (o[2] >> 6);
            bad_row += o[0] != (L & 15);
            bad_col += o[1] != (L & 15);
            bad_k += k != NE * (L >> 4) + j;
            bad_n += o[3] != 16;
        }
    printf("   hypothesis  A[row = L&15][k = %d (L>>4) + j], B[k = %d (L>>4) + j][col = L&15]:  row mismatches %d, col mismatches %d, k mismatches %d, match-count != 16: %d\n",
           NE, NE, bad_row, bad_col, bad_k, bad_n);
    if (bad_row || bad_col || bad_k || bad_n) {
        for (int L = 0; L < 64; ++L) {
            printf("   L=%2d:", L);
            for (int j = 0; j < NE; ++j) {
                const int *o = &h[(L * NE + j) * 4];
                const int k = o[2] < 0 ? -1 : NE * ((o[2] & 63) >> 4) + (o[2] >> 6);
                printf(" (r%d c%d k%d n%d)", o[0], o[1], k, o[3]);
            }
            printf("\n");
        }
    }
    CHECK(hipFree(d));
}


// ---- v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3): A row / B col = lane & 31 and k = 32 (lane >> 5) + byte? ----
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ f32x16 mfma32(u32x8 a, u32x8 b) {
    f32x16 d;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t one = 0x7f7f7f7fu;
    asm volatile("s_nop 7\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "=&v"(d) : "v"(a), "v"(b), "v"(one));
#endif
    return d;
}
__device__ __forceinline__ u32x8 fill32(bool all_ones, bool hot, int j) {
    u32x8 f;
    for (int w = 0; w < 8; ++w) {
        uint32_t x = 0;
        for (int e = 0; e < 4; ++e)
            if (all_ones || (hot && (w * 4 + e) == j)) x |= 0x38u << (8 * e);
        f[w] = x;
    }
    return f;
}
// out[(L * 32 + j) * 4 + {0: row of A(L,j) from the 32x32 C layout, 1: col of B(L,j), 2: (lane' & 31 == 0) partner lane + 64 j', 3: #matches}]
__global__ void probe_mfma32(int *out) {
    const int L = blockIdx.x, lane = threadIdx.x;
    for (int j = 0; j < 32; ++j) {
        int *o = out + (L * 32 + j) * 4;
        {
            const f32x16 d = mfma32(fill32(false, lane == L, j), fill32(true, false, 0));
            for (int r = 0; r < 16; ++r)
                if (d[r] != 0.f && (lane & 31) == 0) o[0] = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        }
        {
            const f32x16 d = mfma32(fill32(true, false, 0), fill32(false, lane == L, j));
            if (d[0] != 0.f && lane < 32) o[1] = lane;
        }
        int matches = 0, first = -1;
        for (int L2 = 0; L2 < 64; ++L2)
            for (int j2 = 0; j2 < 32; ++j2) {
                const f32x16 d = mfma32(fill32(false, lane == L, j), fill32(false, lane == L2, j2));
                bool nz = false;
                for (int r = 0; r < 16; ++r) nz = nz || d[r] != 0.f;
                if (__any(nz)) {
                    ++matches;
                    if ((L2 & 31) == 0) first = L2 + 64 * j2;
                }
            }
        if (lane == 0) { o[2] = first; o[3] = matches; }
    }
}
void run_mfma32() {
    int *d;
    CHECK(hipMalloc(&d, 64 * 32 * 4 * sizeof(int)));
    CHECK(hipMemset(d, 0xff, 64 * 32 * 4 * sizeof(int)));
    probe_mfma32<<<64, 64>>>(d);
    CHECK(hipDeviceSynchronize());
    std::vector<int> h(64 * 32 * 4);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost));
    int bad_row = 0, bad_col = 0, bad_k = 0, bad_n = 0;
    for (int L = 0; L < 64; ++L)
        for (int j = 0; j < 32; ++j) {
            const int *o = &h[(L * 32 + j) * 4];
            const int k = o[2] < 0 ? -1 : 32 * ((o[2] & 63) >> 5) + (o[2] >> 6);
            bad_row += o[0] != (L & 31);
            bad_col += o[1] != (L & 31);
            bad_k += k != 32 * (L >> 5) + j;
            bad_n += o[3] != 32;
        }
    printf("== v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3, unit scales): hypothesis A[row = L&31][k = 32 (L>>5) + j], B[k = 32 (L>>5) + j][col = L&31], C row = (r&3) + 8 (r>>2) + 4 (lane>>5):\n"
           "   row mismatches %d, col mismatches %d, k mismatches %d, match-count != 32: %d\n", bad_row, bad_col, bad_k, bad_n);
    if (bad_row || bad_col || bad_k || bad_n)
        for (int L = 0; L < 64; ++L) {
            printf("   L=%2d:", L);
            for (int j = 0; j < 32; ++j) {
                const int *o = &h[(L * 32 + j) * 4];
                printf(" (r%d c%d k%d n%d)", o[0], o[1], o[2] < 0 ? -1 : 32 * ((o[2] & 63) >> 5) + (o[2] >> 6), o[3]);
            }
            printf("\n");
        }
    CHECK(hipFree(d));
}

template <int BITS>
void run_tr() {
    uint32_t *d;
    CHECK(hipMalloc(&d, 2 * 64 * 2 * 4));
    probe_tr<BITS><<<1, 64>>>(d);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> h(2 * 64 * 2);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    printf("== ds_read_b64_tr_b%d, lane l pointing at LDS byte 64 l: destination lane -> (source lane . source byte) of its 8 bytes\n", BITS);
    for (int l = 0; l < 64; ++l) {
        printf("   lane %2d:", l);
        for (int b = 0; b < 8; ++b) {
            const int sl = (h[(0 * 64 + l) * 2 + b / 4] >> (8 * (b & 3))) & 0xff;
            const int sb = (h[(1 * 64 + l) * 2 + b / 4] >> (8 * (b & 3))) & 0xff;
            printf(" %2d.%d", sl, sb);
        }
        printf("\n");
    }
    CHECK(hipFree(d));
}

int main() {
    run_mfma<BF16_16>();
    run_mfma<FP8_16>();
    run_mfma<MX8_16>();
    run_mfma32();
    run_tr<16>();
    run_tr<8>();
    return 0;
}
