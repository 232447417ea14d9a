// tools/probe_bufload.hip — semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 (developer aid): where the 16
// bytes of lane l land in LDS (M0, instruction offset), what out-of-range lanes write (raw buffer, num_records), and how
// soffset / the instruction offset enter the global address.   hipcc --offload-arch=gfx950 -O3 tools/probe_bufload.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// src: n_dw dwords, src[i] = i.  Descriptor: base = src, num_records = valid_bytes.  Each test fills 4 KiB of LDS with
// 0xAAAAAAAA, issues ONE load, and dumps LDS.
__global__ void probe(const uint32_t *src, uint32_t valid_bytes, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024 * 4];
    const int lane = threadIdx.x;
    const uint64_t base = (uint64_t)src;
    u32x4 desc = {(uint32_t)base, (uint32_t)(base >> 32) & 0xffffu, valid_bytes, 0x00020000u};
    desc[0] = __builtin_amdgcn_readfirstlane(desc[0]); desc[1] = __builtin_amdgcn_readfirstlane(desc[1]);
    desc[2] = __builtin_amdgcn_readfirstlane(desc[2]); desc[3] = __builtin_amdgcn_readfirstlane(desc[3]);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds;
    for (int test = 0; test < 4; ++test) {
        for (int i = lane; i < 1024; i += 64) lds[i] = 0xAAAAAAAAu;
        __syncthreads();
        const uint32_t voff = lane * 16;                      // lane l reads dwords 4l .. 4l+3 (+ offsets)
        const uint32_t m0v = lds0 + 256;                      // LDS destination base
        if (test == 0)  // plain: soffset 0, no instruction offset
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" :: "v"(voff), "s"(desc), "s"(m0v) : "memory");
        if (test == 1)  // soffset 512 bytes
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)" :: "v"(voff), "s"(desc), "s"(m0v), "s"(512u) : "memory");
        if (test == 2)  // instruction offset 1024
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:1024 lds\n\ts_waitcnt vmcnt(0)" :: "v"(voff), "s"(desc), "s"(m0v) : "memory");
        if (test == 3) {  // lanes >= 32 out of range through voffset (valid_bytes = 512 + ...)
            const uint32_t voff2 = lane * 16 + (lane >= 48 ? 1u << 20 : 0);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" :: "v"(voff2), "s"(desc), "s"(m0v) : "memory");
        }
        __syncthreads();
        for (int i = lane; i < 1024; i += 64) out[test * 1024 + i] = lds[i];
        __syncthreads();
    }
}

int main() {
    const int n_dw = 4096;
    std::vector<uint32_t> h(n_dw);
    for (int i = 0; i < n_dw; ++i) h[i] = i;
    uint32_t *d_src, *d_out;
    CHECK(hipMalloc(&d_src, n_dw * 4));
    CHECK(hipMalloc(&d_out, 4 * 1024 * 4));
    CHECK(hipMemcpy(d_src, h.data(), n_dw * 4, hipMemcpyHostToDevice));
    const uint32_t valid = 32 * 16 + 8 * 16;  // 40 lanes' worth of bytes in range for the plain test
    probe<<<1, 64>>>(d_src, valid, d_out);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> o(4 * 1024);
    CHECK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost));
    const char *names[4] = {"plain (M0 = lds+256)", "soffset 512", "inst offset 1024", "lanes >= 48 far out of range"};
    for (int t = 0; t < 4; ++t) {
        printf("== test %d: %s; num_records = %u bytes.  LDS dword index : value (0xAAAAAAAA = untouched)\n", t, names[t], valid);
        int first = -1, last = -1;
        for (int i = 0; i < 1024; ++i) if (o[t * 1024 + i] != 0xAAAAAAAAu) { if (first < 0) first = i; last = i; }
        printf("   touched dwords [%d, %d]\n", first, last);
        for (int i = first; i >= 0 && i <= last; i += 4)
            if ((i - first) / 4 % 8 == 0 || o[t * 1024 + i] == 0 || i + 4 > last)
                printf("   lds[%4d..] = %u %u %u %u\n", i, o[t * 1024 + i], o[t * 1024 + i + 1], o[t * 1024 + i + 2], o[t * 1024 + i + 3]);
    }
    return 0;
}
